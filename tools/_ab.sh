export GIMS_BENCH_NO_STAGE_TIMERS=1
for rep in 1 2 3; do for v in "auto 1" "auto 0" "bf16 1"; do set -- $v; GIMS_GUARD_WALK=$2 python bench.py --kpts 4096 --pairs 8 --no-cpu-baseline --attention-precision $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 walk=$2', d['value'], d['ms_per_step'])"; done; done
