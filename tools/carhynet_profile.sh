# per-kernel stats of the CAR-HyNet descriptor stage (run through gpurun from the repo root); csv only, wrapped in timeout
R=$PWD; O=$R/gpurun_out/ch; mkdir -p $O
python tools/carhynet_bench.py ${CH_BENCH_ARGS:---no-cpu} > $O/bench_carhynet.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/tools/carhynet_bench.py --patches 16384 --reps 2 --no-cpu > $O/ks.log 2>&1
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats_carhynet.csv
rm -rf $O/ks
python3 - <<PY
import csv, json
print(open("$O/bench_carhynet.json").read()[:260])
for r in list(csv.DictReader(open("$O/kernel_stats_carhynet.csv")))[:12]:
    print(f"{r['Name'][:78]:78s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:7.2f} ms {r['Percentage']}%")
PY
