for cfg in "" "GIMS_CH_HALF_LDS=90000" "GIMS_CH_STAGGER=0" "GIMS_CH_HALF=0"; do echo "== $cfg"; env $cfg python tools/carhynet_bench.py --no-cpu 2>/dev/null | tail -1 | cut -c70-130; done
GIMS_CH_PROF=1 python tools/carhynet_bench.py --no-cpu --reps 1 2>&1 | grep "ch_conv_block" | sort | uniq -c | sort -rn | head -12
