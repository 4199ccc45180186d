#!/bin/bash
# Run-to-run spread of the headline workload on ONE box (through gpurun):  tools/bench_20runs.sh r03
# 20 consecutive processes; prints pairs/s, ms per step, host wall per step {median, max} and the number of on-chip Sinkhorn launches
# that fell to the rescue (status 2) -- the bounded waits of ot_res2_kernel are the thing this file watches.
TAG=${1:-r04}
O=gpurun_out/refresh
mkdir -p $O
F=$O/${TAG}_bench_20runs.txt
echo '20 consecutive `python bench.py --kpts 4096 --pairs 8 --steps 20 --warmup 3 --no-cpu-baseline` on one box (final build):' > $F
echo 'pairs/s, ms per step, host wall per step {median, max}, Sinkhorn stage ms, on-chip solves that fell to the rescue' >> $F
for i in $(seq 20); do
  python bench.py --kpts 4096 --pairs 8 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
h = d.get('host_step_ms', {})
print('%.1f %.2f %.2f %.2f %.3f %d' % (d['value'], d['ms_per_step'], h.get('median', float('nan')), h.get('max', float('nan')), d['stage_ms_per_step']['sinkhorn'], d.get('sinkhorn_rescues', -1)))" >> $F
done
cat $F
