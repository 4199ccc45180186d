# PMC counters of the CAR-HyNet per-patch kernels (separate passes; csv; from the repo root through gpurun): bash tools/ch_pmc.sh
R=$PWD; O=$R/gpurun_out/chpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/carhynet_bench.py --patches 16384 --reps 1 --no-cpu > $O/p$i.log 2>&1
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_table.py $f ch_conv_block ch_sandglass > $O/pmc_$i.txt 2>&1
  rm -rf $O/p$i
done
cat $O/pmc_*.txt
