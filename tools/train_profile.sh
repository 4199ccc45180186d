# per-kernel stats of the training step (run through gpurun from the repo root); csv only, wrapped in timeout
R=$PWD; O=$R/gpurun_out/train; mkdir -p $O
timeout 200 python tools/train_bench.py ${TRAIN_BENCH_ARGS:---no-cpu} > $O/bench_train.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/tools/train_bench.py --steps 3 --warmup 1 --no-cpu > $O/ks.log 2>&1
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats_train.csv
rm -rf $O/ks
python3 - <<PY
import csv
print(open("$O/bench_train.json").read()[:900])
for r in list(csv.DictReader(open("$O/kernel_stats_train.csv")))[:24]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
PY
