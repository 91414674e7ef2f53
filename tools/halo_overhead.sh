#!/bin/bash
# profiles/<tag>_halo_overhead.txt: tools/halo_overlap_bench.py at the C4 block size, every mode in its own process
# (the overlap knob is read at pcl_comm_init), bubble and dense state, twice.   Usage: tools/halo_overhead.sh [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
STEPS=${1:-300}
for st in bubble dense; do
  echo "== state $st, 4096 x 2048 block (BASELINE configs[3]'s 2 x 4 layout), 8 self-neighbours, $STEPS steps"
  for rep in 1 2; do
    for mode in none seq ovl ahead split; do
      PCL_HALO_BENCH_STATE=$st python3 $R/tools/halo_overlap_bench.py 4096 2048 $STEPS $mode || exit 1
    done
  done
done
