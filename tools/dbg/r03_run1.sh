#!/bin/bash
# default bench line (incl. app_run), unsplit PMC passes, halo overhead table
set -x
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof_r03a/pmc_unsplit_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-states --unsplit > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof_r03a/pmc_3d_$C -- python3 $R/bench.py --steps 2 --warmup 1 --ndim 3 --unsplit --nx 256 > /dev/null 2>&1
done
cd $R
for S in bubble dense; do
  echo "== state $S, 4096 x 2048 block (BASELINE configs[3]'s 2 x 4 layout), 8 self-neighbours, 300 steps"
  for i in 1 2; do PCL_HALO_BENCH_STATE=$S python tools/halo_overlap_bench.py 4096 2048 300 2>&1 | grep ms/step; done
done > gpurun_out/r03_halo_overhead.txt
