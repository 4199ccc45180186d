R=$PWD; O=$R/gpurun_out/seq; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --kpts 1024 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline > $O/log 2>&1
python3 $R/tools/seq_report.py $(find $O/t -name '*kernel_trace.csv' | head -1) > $O/seq_1024x32.txt
rm -rf $O/t
