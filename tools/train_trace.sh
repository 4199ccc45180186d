# kernel timeline of the training step (through gpurun from the repo root): rocprofv3 --kernel-trace CSV -> tools/train_timeline.py
R=$PWD; O=$R/gpurun_out/train; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/train_bench.py --steps 3 --warmup 1 --no-cpu > $O/kt.log 2>&1
f=$(find $O/kt -name '*kernel_trace.csv' | head -1)
python3 $R/tools/train_timeline.py $f ${1:-adam_kernel} | tee $O/train_timeline.txt
rm -rf $O/kt
