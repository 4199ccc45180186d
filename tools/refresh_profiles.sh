#!/bin/bash
# Regenerates the round's measurement evidence on a GPU box (run through gpurun from the repo root):
#   tools/refresh_profiles.sh r04
# Outputs land in gpurun_out/refresh/; copy what should be judged into profiles/ (tools/README.md).
# Every rocprofv3 run is csv-only and wrapped in `timeout` (a run that builds the rocpd database can hang for minutes).
set -u
TAG=${1:-r06}
export GIMS_HEAD=${2:-unknown}        # commit the profiles are taken on (no .git on the GPU box: pass $(git rev-parse --short HEAD))
R=$PWD
O=$R/gpurun_out/refresh
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
# the driver's command: headline 2x4096x8 + the 2x1024x32 block, CPU baseline included
python bench.py --steps 20 --warmup 3 > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
cp $R/bench_extra.json $O/${TAG}_bench_extra.json 2>/dev/null      # the full record of THAT run (later runs of this script overwrite bench_extra.json)
cd /tmp && export TMPDIR=/tmp
# kernel stats of the SAME command (no CPU baseline: it only adds host time)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_default -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/ks_default.log 2>&1
cp $(find $O/ks_default -name '*kernel_stats.csv' | head -1) $O/${TAG}_rocprofv3_kernel_stats_bench_default.csv
rm -rf $O/ks_default
for cfg in "1024 32" "4096 8"; do
  set -- $cfg
  tag=${1}x${2}
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -- python3 $R/bench.py --kpts $1 --pairs $2 --steps 20 --warmup 3 --no-cpu-baseline > $O/ks_$tag.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf_$tag -- python3 $R/bench.py --kpts $1 --pairs $2 --steps 1 --warmup 1 --no-cpu-baseline > $O/pf_$tag.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw_$tag -- python3 $R/bench.py --kpts $1 --pairs $2 --steps 1 --warmup 1 --no-cpu-baseline > $O/pw_$tag.log 2>&1
  f=$(find $O/pf_$tag -name '*counter_collection.csv' | head -1)
  w=$(find $O/pw_$tag -name '*counter_collection.csv' | head -1)
  python3 $R/tools/pmc_summary.py $f $w $tag $O/pmc_traffic.json > $O/${TAG}_pmc_traffic_$tag.txt
  cp $(find $O/ks_$tag -name '*kernel_stats.csv' | head -1) $O/${TAG}_rocprofv3_kernel_stats_bench_$tag.csv
  rm -rf $O/pf_$tag $O/pw_$tag $O/ks_$tag
done
# matrix-pipe busy share of the final build (its own pass: SQ counters only)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -- python3 $R/bench.py --kpts 4096 --pairs 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/sq.log 2>&1
python3 - <<PY > $O/${TAG}_pmc_sq_counters_4096x8.txt
import csv, glob, re, collections
f = glob.glob("$O/sq/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])) if f else []:
    k = re.sub(r"<.*|\(.*", "", r["Kernel_Name"].replace("void ", "").replace("gims::", ""))
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
print("kernel, launches, SQ_BUSY_CYCLES (sum), SQ_VALU_MFMA_BUSY_CYCLES (sum), MFMA busy / SQ busy   [bench.py --kpts 4096 --pairs 8, 2 steps]")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:14]:
    b, m = v.get("SQ_BUSY_CYCLES", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    print(f"{k:34s} {cnt[(k, 'SQ_BUSY_CYCLES')]:5d} {b:16.0f} {m:16.0f} {m / b if b else 0:8.3f}")
PY
rm -rf $O/sq
cat $O/${TAG}_bench_default.json | head -c 600
