R=$PWD; O=$R/gpurun_out/ks; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "1024 32" "4096 8"; do set -- $cfg; tag=${1}x${2}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -- python3 $R/bench.py --kpts $1 --pairs $2 --steps 3 --warmup 1 --no-cpu-baseline > $O/ks_$tag.log 2>&1
cp $(find $O/ks_$tag -name '*kernel_stats.csv' | head -1) $O/kernel_stats_$tag.csv
python3 $R/tools/gap_report.py $(find $O/ks_$tag -name '*kernel_trace.csv' | head -1) > $O/gap_$tag.txt
rm -rf $O/ks_$tag; done
