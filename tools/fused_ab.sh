#!/bin/bash
# tools/fused_ab.sh -- same-box A/B of builds of the one-kernel dim-split step (tile shapes of classic_fused.hpp):
# each variant is a library under build/ (PCL_LIB_OVERRIDE) plus optional environment, the step form is pinned to the
# one kernel, the runs are interleaved.
#   MODES="bubble ablate dense" tools/fused_ab.sh <out-file> <variant>...     variant = lib.so|-[@K=V[,K=V]]  ("-" = in-tree library)
out=$1; shift
: > "$out"
MODES=${MODES:-bubble ablate dense}
for round in 1 2; do
  for var in "$@"; do
    lib=${var%%@*}; venv=""
    [ "$var" != "$lib" ] && venv=$(echo "${var#*@}" | tr ',' ' ')
    for mode in $MODES; do
      extra=""; envs="PCL_TUNE_FUSED_STEP=1 $venv"
      [ "$mode" = ablate ] && envs="$envs PCL_TUNE_ABLATE=1"
      [ "$mode" = dense ] && extra="--state dense"
      [ "$mode" = developed ] && extra="--state developed"
      [ "$lib" != "-" ] && envs="$envs PCL_LIB_OVERRIDE=$PWD/$lib"
      line=$(env $envs python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-states --no-app-run $extra 2>/dev/null | tail -1)
      echo "$round $var $mode $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms_per_step %.4f  kernel_ms %s" % (d["ms_per_step"], ["%.4f" % v for v in d["roofline"]["avg_ms"].values() if v]))')" | tee -a "$out"
    done
  done
done
