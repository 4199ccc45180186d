# LDS bank-conflict share per kernel (rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; own pass, csv, timeout) of a command
# usage (from the repo root, through gpurun): bash tools/lds_conflicts.sh <tag> python3 <script> [args]
R=$PWD; TAG=$1; shift; O=$R/gpurun_out/lds; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/$TAG -- "$@" > $O/$TAG.log 2>&1
python3 - <<PY > $O/lds_$TAG.txt
import csv, glob, re, collections
f = glob.glob("$O/$TAG/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])) if f else []:
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("gims::", ""))[:80]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k] += 1
print("kernel, launches, SQ_LDS_BANK_CONFLICT (cycles, summed), SQ_LDS_IDX_ACTIVE (cycles), conflict share")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_LDS_BANK_CONFLICT", 0))[:18]:
    b, a = v.get("SQ_LDS_BANK_CONFLICT", 0.0), v.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"{k:82s} {cnt[k] // 2:5d} {b:14.0f} {a:14.0f} {b / a if a else 0:7.3f}")
PY
rm -rf $O/$TAG
cat $O/lds_$TAG.txt
