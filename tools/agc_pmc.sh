#!/bin/bash
# PMC passes over the graph-build kernels (through gpurun):  tools/agc_pmc.sh   -> gpurun_out/agc/pmc_*.txt
O=$PWD/gpurun_out/agc; mkdir -p $O
python3 tools/agc_loop.py 4096 16 5
i=0
for set in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 tools/agc_loop.py 4096 16 3 > $O/p$i.log 2>&1
  f=$(ls $O/p$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_table.py $f "agc_simw_kernel<2>" "agc_simw_kernel<1>" > $O/pmc_$i.txt; cat $O/pmc_$i.txt; else echo "pass $i ($set): no counters"; tail -3 $O/p$i.log; fi
  rm -rf $O/p$i
done
