#!/bin/bash
# AGC kernels under rocprofv3 at 4096x8 + the AGC parity tests (through gpurun):  tools/agc_quick.sh
O=gpurun_out/agc; mkdir -p $O; rm -rf $O/prof
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_gmatcher_gpu.py -x -q -m gpu -k "agc or window or capacity or e2e_vs_reference" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --kpts 4096 --pairs 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 tools/kstats.py $(ls $O/prof/*/*kernel_stats.csv | head -1) agc_ | tee $O/agc_kstats.txt
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/agc/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['stage_ms_per_step'])
PY
