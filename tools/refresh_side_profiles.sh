#!/bin/bash
# The round's evidence for the rows either side of the matcher: CAR-HyNet, descriptors + matcher, patch extraction, training step.
#   tools/refresh_side_profiles.sh r03     (through gpurun from the repo root; outputs in gpurun_out/side/, copy into profiles/)
TAG=${1:-r04}
R=$PWD; O=$R/gpurun_out/side; mkdir -p $O
python tools/carhynet_bench.py > $O/${TAG}_bench_carhynet_16384.json 2> $O/carhynet.err
python tools/pipeline_bench.py --kpts 8192 --pairs 2 > $O/${TAG}_pipeline_8192x2.json 2> $O/pipe8192.err
python tools/pipeline_bench.py --kpts 4096 --pairs 8 > $O/${TAG}_pipeline_4096x8.json 2> $O/pipe4096.err
python tools/patches_bench.py > $O/${TAG}_bench_patches_8192.json 2> $O/patches.err
TRAIN_BENCH_ARGS="--steps 10" bash tools/train_profile.sh > $O/train_profile.log 2>&1
cp gpurun_out/train/bench_train.json $O/${TAG}_bench_train_2048.json
cp gpurun_out/train/kernel_stats_train.csv $O/${TAG}_rocprofv3_kernel_stats_train_2048.csv
bash tools/carhynet_profile.sh > $O/carhynet_profile.log 2>&1
cp gpurun_out/ch/kernel_stats_carhynet.csv $O/${TAG}_rocprofv3_kernel_stats_carhynet_16384.csv
for f in $O/${TAG}_*.json; do echo "== $f"; tail -1 $f | cut -c1-300; done
