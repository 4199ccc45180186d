#!/bin/bash
# Power, clocks and temperature of the GPU while the headline workload runs (through gpurun from the repo root):  tools/power_watch.sh [kpts pairs]
# -> gpurun_out/power_watch.txt.  rocm-smi is sampled from a second process every ~0.5 s; the first samples are the idle chip.
K=${1:-4096}; P=${2:-8}
O=gpurun_out/power_watch.txt
mkdir -p gpurun_out
{
echo "== devices"; rocm-smi --showid 2>&1 | grep -i "GPU\[" | head -10
echo "== cap"; rocm-smi --showmaxpower 2>&1 | grep -i "GPU\[" | head -4
echo "== idle"; rocm-smi --showpower --showclocks --showtemp --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
GIMS_BENCH_NO_STAGE_TIMERS=1 python bench.py --kpts $K --pairs $P --steps 2000 --warmup 3 --no-cpu-baseline > gpurun_out/power_watch_bench.json 2> /dev/null &
BP=$!
sleep 14      # import + warm-up + calibration
for i in $(seq 16); do
  echo "== sample $i"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|Temperature (Sensor junction)\|Temperature (Sensor memory)" | head -12
  sleep 1
done
wait $BP
echo "== bench line"; tail -1 gpurun_out/power_watch_bench.json | cut -c1-200
} > $O 2>&1
cat $O
