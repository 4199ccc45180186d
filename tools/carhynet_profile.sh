R=$PWD; O=$R/gpurun_out/ch; mkdir -p $O
python tools/carhynet_bench.py > $O/bench_carhynet.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/tools/carhynet_bench.py --patches 16384 --reps 2 --no-cpu > $O/ks.log 2>&1
cp $(find $O/ks -name '*kernel_stats.csv' | head -1) $O/kernel_stats_carhynet.csv
rm -rf $O/ks
cat $O/bench_carhynet.json
