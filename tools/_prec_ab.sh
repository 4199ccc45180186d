for v in "x6 x6 x6" "x3 x6 x6" "x3 x3 x6" "x3 x3 x3"; do
set -- $v
echo "== attn_bwd $1 wgrad $2 agrad $3"
GIMS_TRAIN_PREC_ATTN_BWD=$1 GIMS_TRAIN_PREC_WGRAD=$2 GIMS_TRAIN_PREC_AGRAD=$3 python tools/train_prec_probe.py 2>&1 | grep trainstep
GIMS_TRAIN_PREC_ATTN_BWD=$1 GIMS_TRAIN_PREC_WGRAD=$2 GIMS_TRAIN_PREC_AGRAD=$3 python tools/train_bench.py --steps 10 --no-cpu 2>/dev/null | tail -1 | cut -c1-200
done
