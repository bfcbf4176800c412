#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dgpmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0; files=""
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_CYCLES" \
           "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); rm -rf $O/sq$i
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq$i -o t -- python3 $R/tools/matvec_dg_cheby.py 4 18 5 --bases 0 --outer 1 > $O/sq$i.log 2>&1
  files="$files $(find $O/sq$i -name '*counter_collection.csv' | head -1)"
done
python3 $R/tools/pmc_summary.py $files --min-blocks 256 --match dg_cell > $O/summary.txt
rm -rf $O/sq1 $O/sq2 $O/sq3
cat $O/summary.txt
