#!/bin/bash
# SQ / TA / TCP / TCC counters of the macro-element kernels, one rocprofv3 --pmc pass per counter set (no trace domains
# next to --pmc).  usage: tools/r04_sq.sh <tag> <match> <matvec_loop.py arguments...>     -> gpurun_out/r04/sq_<tag>.txt
# MGX_LIB_PATH / MATVEC_OPTS select the library variant / context options.
# (a pass with TA_* counters aborted rocprofv3 on this pool and is left out)
tag=$1; match=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04/sq_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
sets=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum"
 "GRBM_GUI_ACTIVE GRBM_TA_BUSY"
)
files=""
i=0
for set in "${sets[@]}"; do
  i=$((i + 1)); rm -rf $O/p$i
  echo "pass $i: $set"
  if timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o t -- python3 $R/tools/matvec_loop.py "$@" > $O/p$i.log 2>&1; then
    f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && files="$files $f"
  else
    echo "pass $i failed: $set" >> $O/failed.txt
  fi
done
python3 $R/tools/pmc_summary.py $files --min-blocks 200 --match "$match" > $R/gpurun_out/r04/sq_$tag.txt
[ -f $O/failed.txt ] && cat $O/failed.txt >> $R/gpurun_out/r04/sq_$tag.txt
rm -rf $O
cat $R/gpurun_out/r04/sq_$tag.txt
