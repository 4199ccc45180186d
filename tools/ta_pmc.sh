# PMC counters of the training-attention kernels (separate passes; csv; from the repo root through gpurun): bash tools/ta_pmc.sh
R=$PWD; O=$R/gpurun_out/tapmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES"; do
  i=$((i+1))
  PROBE_SPLITS=4 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/train_attn_probe.py 2048 > $O/p$i.log 2>&1
  f=$(find $O/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $R/tools/pmc_table.py $f ta_fwd_kernel ta_bwd_q_kernel ta_bwd_kv_kernel > $O/pmc_$i.txt 2>&1
  rm -rf $O/p$i
done
cat $O/pmc_*.txt
