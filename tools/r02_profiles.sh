#!/bin/bash
# Round-2 evidence, collected on the MI355X box into gpurun_out/r02/ (copied to profiles/ afterwards):
#   bench line, rocprofv3 kernel trace + stats of the same command, HBM traffic (FETCH_SIZE / WRITE_SIZE in
#   separate passes) and SQ counters of the macro-element brick kernel, the p = 8 bench line, the DG harness.
# usage: tools/r02_profiles.sh [stage ...]   stages: bench trace forms pmc sq p8 dg shell   (default: all)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stages=${@:-bench trace forms pmc sq p8 dg shell}
for s in $stages; do
case $s in
bench)
  python3 $R/bench.py > $O/bench_128cube_p4.json 2> $O/bench_128cube_p4.err
  tail -1 $O/bench_128cube_p4.json | cut -c1-400 ;;
trace)
  rm -rf $O/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python3 $R/bench.py --no-cpu-baseline > $O/bench_profiled.log 2>&1
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_128cube_p4.csv
  python3 $R/tools/summarize_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 2.0 > $O/kernel_trace_by_grid_128cube_p4.txt
  rm -rf $O/kt
  head -12 $O/kernel_trace_by_grid_128cube_p4.txt | cut -c1-160 ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 $R/tools/matvec_loop.py 128 3 cheb > $O/pmc_$c.log 2>&1
  done
  python3 $R/tools/make_traffic_json.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
      $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 128 $O/pmc_traffic_128cube_p4.json
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
sq)
  i=0; files=""
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_CYCLES" \
             "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1)); rm -rf $O/sq$i
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/sq$i -o t -- python3 $R/tools/matvec_loop.py 128 3 cheb > $O/sq$i.log 2>&1
    files="$files $(find $O/sq$i -name '*counter_collection.csv' | head -1)"
  done
  python3 $R/tools/pmc_summary.py $files --min-blocks 256 --match brick_macro > $O/pmc_sq_macro_kernel_128cube_p4.txt
  rm -rf $O/sq1 $O/sq2 $O/sq3
  head -30 $O/pmc_sq_macro_kernel_128cube_p4.txt ;;
p8)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc8_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc8_$c -o t -- python3 $R/tools/matvec_loop.py 64 3 cheb 8 > $O/pmc8_$c.log 2>&1
  done
  python3 $R/tools/make_traffic_json.py $(find $O/pmc8_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
      $(find $O/pmc8_WRITE_SIZE -name "*counter_collection.csv" | head -1) 64 $O/pmc_traffic_64cube_p8.json 8
  rm -rf $O/pmc8_FETCH_SIZE $O/pmc8_WRITE_SIZE
  python3 $R/bench.py --degree 8 --cells 64 --no-cpu-baseline > $O/bench_64cube_p8.json 2> $O/bench_64cube_p8.err
  tail -1 $O/bench_64cube_p8.json | cut -c1-400
  rm -rf $O/kt8
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt8 -o t -- python3 $R/bench.py --degree 8 --cells 64 --no-cpu-baseline --steps 5 > $O/bench_p8_profiled.log 2>&1
  python3 $R/tools/summarize_trace.py $(find $O/kt8 -name "*kernel_trace.csv" | head -1) 2.0 > $O/kernel_trace_by_grid_64cube_p8.txt
  rm -rf $O/kt8 ;;
dg)
  python3 $R/tools/matvec_dg_cheby.py 4 21 10 --outer 3 --json --cpu-baseline > $O/matvec_dg_cheby_p4_262M.txt 2>&1
  python3 $R/tools/matvec_dg_cheby.py 3 21 10 --outer 3 --json > $O/matvec_dg_cheby_p3_134M.txt 2>&1
  rm -rf $O/ktdg
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktdg -o t -- python3 $R/tools/matvec_dg_cheby.py 4 18 20 > $O/dg_profiled.log 2>&1
  cp $(find $O/ktdg -name "*kernel_stats.csv" | head -1) $O/kernel_stats_matvec_dg_cheby_p4_33M.csv
  rm -rf $O/ktdg
  grep Best $O/matvec_dg_cheby_p4_262M.txt | cut -c1-200 ;;
forms)
  # only the finest level runs here: the per-symbol averages of --stats are per colour launch of the 135 M DoF level
  # (in the bench trace a symbol also covers the launches of the coarser brick levels with the same persistent grid)
  rm -rf $O/ktf
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktf -o t -- python3 $R/tools/matvec_loop.py 128 10 cheb > $O/forms_profiled.log 2>&1
  cp $(find $O/ktf -name "*kernel_stats.csv" | head -1) $O/kernel_stats_finest_level_forms_128cube_p4.csv
  rm -rf $O/ktf
  head -8 $O/kernel_stats_finest_level_forms_128cube_p4.csv | cut -c1-60,200-330 ;;
shell)
  python3 $R/tools/shell_bench.py 4 6 > $O/shell_sector_matvec_p4.txt 2>&1
  tail -1 $O/shell_sector_matvec_p4.txt ;;
esac
done
