#!/bin/bash
# Regenerates the round's measurement evidence on a GPU box (run through gpurun from the repo root).
# Outputs land in gpurun_out/refresh/; copy what should be judged into profiles/.
set -u
R=$PWD
O=$R/gpurun_out/refresh
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
python bench.py --steps 10 --warmup 3 > $O/bench_1024x32.json 2> $O/bench_1024x32.err
python bench.py --kpts 4096 --pairs 8 --steps 6 --warmup 2 > $O/bench_4096x8.json 2> $O/bench_4096x8.err
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 32" "4096 8"; do
  set -- $cfg
  tag=${1}x${2}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -- python3 $R/bench.py --kpts $1 --pairs $2 --steps 3 --warmup 1 --no-cpu-baseline > $O/ks_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf_$tag -- python3 $R/bench.py --kpts $1 --pairs $2 --steps 1 --warmup 1 --no-cpu-baseline > $O/pf_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw_$tag -- python3 $R/bench.py --kpts $1 --pairs $2 --steps 1 --warmup 1 --no-cpu-baseline > $O/pw_$tag.log 2>&1
  f=$(find $O/pf_$tag -name '*counter_collection.csv' | head -1)
  w=$(find $O/pw_$tag -name '*counter_collection.csv' | head -1)
  python3 $R/tools/pmc_summary.py $f $w $tag $O/pmc_traffic.json
  cp $(find $O/ks_$tag -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$tag.csv
  rm -rf $O/pf_$tag $O/pw_$tag $O/ks_$tag
done
cat $O/bench_1024x32.json $O/bench_4096x8.json
