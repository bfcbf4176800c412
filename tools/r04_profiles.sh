#!/bin/bash
# Round-4 evidence, collected on the MI355X box into gpurun_out/r04p/ (copied to profiles/r04_* afterwards).
# usage: tools/r04_profiles.sh [stage ...]   stages: bench trace pmc pmc8 sq levels f32 p8 rank markers  (default: all)
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
stages=${@:-bench trace pmc pmc8 levels f32 p8 rank markers}
pmc_traffic() { # <cells> <degree> <out.json>
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$c $O/cal_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o t -- python3 $R/tools/matvec_loop.py $1 2 all $2 > $O/pmc_$c.log 2>&1
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/cal_$c -o t -- python3 $R/tools/matvec_loop.py $1 6 calib $2 > $O/cal_$c.log 2>&1
    echo "pmc $c done"
  done
  python3 $R/tools/make_traffic_json.py $(find $O/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
      $(find $O/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $1 $3 $2 \
      $(find $O/cal_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/cal_WRITE_SIZE -name "*counter_collection.csv" | head -1)
  rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/cal_FETCH_SIZE $O/cal_WRITE_SIZE
}
for s in $stages; do
case $s in
bench)
  python3 $R/bench.py > $O/bench_128cube_p4.json 2> $O/bench_128cube_p4.err
  tail -1 $O/bench_128cube_p4.json | cut -c1-300 ;;
trace)
  rm -rf $O/kt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o t -- python3 $R/bench.py --no-cpu-baseline --no-verify --no-secondary > $O/bench_profiled.log 2>&1
  cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bench_128cube_p4.csv
  python3 $R/tools/summarize_trace.py $(find $O/kt -name "*kernel_trace.csv" | head -1) 1.0 > $O/kernel_trace_by_grid_128cube_p4.txt
  rm -rf $O/kt
  head -14 $O/kernel_trace_by_grid_128cube_p4.txt | cut -c30-160 ;;
pmc) pmc_traffic 128 4 $O/pmc_traffic_128cube_p4.json ;;
pmc8) pmc_traffic 64 8 $O/pmc_traffic_64cube_p8.json ;;
levels)
  python3 $R/tools/vcycle_levels.py 128 4 > $O/vcycle_levels_128cube_p4.txt 2>&1
  cat $O/vcycle_levels_128cube_p4.txt ;;
p8)
  python3 $R/bench.py --degree 8 --cells 64 --no-cpu-baseline > $O/bench_64cube_p8.json 2> $O/bench_64cube_p8.err
  tail -1 $O/bench_64cube_p8.json | cut -c1-300 ;;
f32)
  python3 $R/bench.py --vcycle-number f32 --no-cpu-baseline > $O/bench_128cube_p4_f32vcycle.json 2> $O/bench_f32.err
  tail -1 $O/bench_128cube_p4_f32vcycle.json | cut -c1-300 ;;
rank)
  for n in 2 4 8; do python3 $R/tools/rank_emulation.py $n 128 10 levels 2>&1 | grep -v "version\|Hostname\|Librccl" ; done > $O/rank_emulation_strong_scaling_128cube_p4.txt
  cat $O/rank_emulation_strong_scaling_128cube_p4.txt ;;
markers)
  rm -rf $O/markers
  MGX_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/markers -o t -- $R/multigrid_amd/poisson_cube 4 2000000 3000000 1 3 3 square > $O/markers.log 2>&1
  cp $O/markers/t_marker_api_stats.csv $O/marker_trace_poisson_cube_driver_p4.csv; rm -rf $O/markers
  head -12 $O/marker_trace_poisson_cube_driver_p4.csv ;;
esac
done
