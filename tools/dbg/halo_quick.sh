R=${GRAFT_REPO_ROOT:-/root/repo}
for sz in "4096 4096" "4096 2048"; do for st in bubble dense; do echo "== $sz $st"; for mode in none seq ovl ahead; do PCL_HALO_BENCH_STATE=$st python3 $R/tools/halo_overlap_bench.py $sz 300 $mode 2>&1 | grep ms/step; done; done; done
