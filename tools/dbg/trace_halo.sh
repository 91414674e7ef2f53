cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in ahead; do
PCL_HALO_BENCH_STATE=bubble rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$mode -- python3 $R/tools/halo_overlap_bench.py 4096 2048 20 $mode > $R/gpurun_out/tr_$mode.log 2>&1
f=$(find $R/gpurun_out/tr_$mode -name "*kernel_trace.csv" | head -1)
echo "== $mode"; python3 $R/tools/dbg/trace_step.py $f
done
