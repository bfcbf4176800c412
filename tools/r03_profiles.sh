#!/bin/bash
# Round-3 evidence, collected on the MI355X box into gpurun_out/r03p/ (copied to profiles/r03_* afterwards).
# usage: tools/r03_profiles.sh [stage ...]   stages: bench trace pmc levels p8 f32 shell dg rank  (default: all)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stages=${@:-bench trace pmc levels p8 f32 shell dg rank}
for s in $stages; do
case $s in
bench)
  python3 $R/bench.py > $O/bench_128cube_p4.json 2> $O/bench_128cube_p4.err
  tail -1 $O/bench_128cube_p4.json | cut -c1-300 ;;
trace)
  rm -rf $O/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python3 $R/bench.py --no-cpu-baseline --no-verify > $O/bench_profiled.log 2>&1
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_128cube_p4.csv
  python3 $R/tools/summarize_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 1.0 > $O/kernel_trace_by_grid_128cube_p4.txt
  rm -rf $O/kt
  head -14 $O/kernel_trace_by_grid_128cube_p4.txt | cut -c30-160 ;;
pmc)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 $R/tools/matvec_loop.py 128 2 all > $O/pmc_$c.log 2>&1
  done
  python3 $R/tools/make_traffic_json.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
      $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) 128 $O/pmc_traffic_128cube_p4.json
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE ;;
levels)
  python3 $R/tools/vcycle_levels.py 128 4 > $O/vcycle_levels_128cube_p4.txt 2>&1
  cat $O/vcycle_levels_128cube_p4.txt ;;
p8)
  python3 $R/bench.py --degree 8 --cells 64 --no-cpu-baseline > $O/bench_64cube_p8.json 2> $O/bench_64cube_p8.err
  tail -1 $O/bench_64cube_p8.json | cut -c1-300 ;;
f32)
  python3 $R/bench.py --vcycle-number f32 --no-cpu-baseline > $O/bench_128cube_p4_f32vcycle.json 2> $O/bench_f32.err
  tail -1 $O/bench_128cube_p4_f32vcycle.json | cut -c1-300 ;;
shell)
  python3 $R/tools/shell_bench.py 4 5 6 > $O/hyper_shell6_matvec_p4.txt 2>&1
  python3 $R/tools/poisson_shell.py 4 40000000 --cycles 4:11 > $O/poisson_shell_p4.txt 2>&1
  tail -3 $O/hyper_shell6_matvec_p4.txt; tail -8 $O/poisson_shell_p4.txt ;;
dg)
  python3 $R/tools/matvec_dg_cheby.py 4 21 10 --outer 3 --json > $O/matvec_dg_cheby_p4_262M.txt 2>&1
  python3 $R/tools/matvec_dg_cheby.py 8 18 10 --outer 3 --json --bases 0 > $O/matvec_dg_cheby_p8_191M.txt 2>&1
  python3 $R/tools/poisson_dg.py 4 > $O/poisson_dg_p4.txt 2>&1 || true
  grep Best $O/matvec_dg_cheby_p4_262M.txt | cut -c1-120 ;;
rank)
  for n in 2 4 8; do python3 $R/tools/rank_emulation.py $n 128 10 levels 2>&1 | grep -v "version\|Hostname\|Librccl" ; done > $O/rank_emulation_strong_scaling_128cube_p4.txt
  cat $O/rank_emulation_strong_scaling_128cube_p4.txt ;;
esac
done
