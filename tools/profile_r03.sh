#!/bin/bash
# Round-3 profile artefacts (run through gpurun; copy gpurun_out/prof_<tag>/summary/* into profiles/ afterwards).
#   1. in SEPARATE passes --pmc FETCH_SIZE / --pmc WRITE_SIZE (MI355X_MICROARCH.md: KB units, FETCH_SIZE x2 on gfx950) of
#      one bench command per kernel family -> <tag>_pmc_hbm.json via tools/pmc_summary.py (bench.py reads
#      roofline.traffic from it; written into the box's profiles/ first, so the lines of step 2 carry THIS build's bytes);
#   2. rocprofv3 --kernel-trace --stats of the same commands, the bench JSON of the SAME run next to it;
#   3. SQ issue counters of the dense state (one-kernel step; PCL_TUNE_FUSED_STEP=0: the two passes) and of the SharpClaw
#      right-hand side, and SQ_LDS_BANK_CONFLICT of the y pass with the swizzled tile (the build)
#      and with the padded round-1 tile (build/libs/libpyclaw_amd_ypad.so, -DPCL_YTILE_PAD=1).
# Usage: tools/profile_r03.sh <tag> <commit>
set -e
TAG=${1:-r03}
COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-states"
declare -A CMD
CMD[exact]="$R/bench.py $B --math exact"
CMD[exact_twopass]="$R/bench.py $B --math exact"          # PCL_TUNE_FUSED_STEP=0: x pass + y pass (classic.hpp)
CMD[exact_dense]="$R/bench.py $B --math exact --state dense"
CMD[fast_dense]="$R/bench.py $B --math fast --state dense"
CMD[unsplit]="$R/bench.py $B --unsplit"
CMD[sharpclaw]="$R/bench.py $B --solver sharpclaw"
CMD[3d_dimsplit]="$R/bench.py --ndim 3 --nx 512"
CMD[3d_unsplit]="$R/bench.py --ndim 3 --unsplit --nx 256"
CMD[sphere_classic]="$R/bench.py --app sphere"
CMD[sphere_sharpclaw]="$R/bench.py --app sphere --solver sharpclaw"
ORDER="exact exact_twopass exact_dense fast_dense unsplit sharpclaw 3d_dimsplit 3d_unsplit sphere_classic sphere_sharpclaw"
for name in $ORDER; do
  export PCL_TUNE_FUSED_STEP=1; [ "$name" = exact_twopass ] && export PCL_TUNE_FUSED_STEP=0
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${name}_$C -- python3 ${CMD[$name]} --steps 2 --warmup 1 > $OUT/pmc_${name}_$C.log 2>&1
  done
  echo "pmc $name done"
done
export PCL_TUNE_FUSED_STEP=1
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_modes.json
# the HBM summary goes into the box's copy of profiles/ BEFORE the kernel-stats passes, so that the bench lines of those
# passes carry roofline.traffic of THIS build (bench.py reads profiles/<tag>_pmc_hbm.json)
python3 - $OUT $TAG $COMMIT $R <<'PY'
import json, sys
out, tag, commit, root = sys.argv[1:5]
modes = json.load(open(out + "/pmc_modes.json"))
res = {"commit": commit,
       "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py <the command of the mode> --steps 2 --warmup 1",
       "correction": "MI355X_MICROARCH.md HBM section: counters are in KB; on gfx950 FETCH_SIZE reads 1/2 of streamed bytes -> x2; WRITE_SIZE exact",
       "modes": modes}
for dst in ("%s/summary/%s_pmc_hbm.json" % (out, tag), "%s/profiles/%s_pmc_hbm.json" % (root, tag)):
    json.dump(res, open(dst, "w"), indent=1)
PY
for name in $ORDER; do
  export PCL_TUNE_FUSED_STEP=1; [ "$name" = exact_twopass ] && export PCL_TUNE_FUSED_STEP=0
  steps=20; [ "$name" = sharpclaw ] && steps=4; [ "$name" = 3d_dimsplit ] && steps=6; [ "$name" = 3d_unsplit ] && steps=6
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 ${CMD[$name]} --steps $steps --warmup 3 \
      > $OUT/summary/${TAG}_bench_$name.json 2> $OUT/stats_$name.err
  cp $(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1) $OUT/summary/${TAG}_kernel_stats_$name.csv
  echo "stats $name done"
done
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
SQB="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/sq_a -- python3 ${CMD[exact_dense]} --steps 3 --warmup 1 > $OUT/sq_a.log 2>&1
rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/sq_b -- python3 ${CMD[exact_dense]} --steps 3 --warmup 1 > $OUT/sq_b.log 2>&1
rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/sq_fast_a -- python3 ${CMD[fast_dense]} --steps 3 --warmup 1 > $OUT/sq_fast_a.log 2>&1
rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/sq_sharp_a -- python3 ${CMD[sharpclaw]} --steps 1 --warmup 1 > $OUT/sq_sharp_a.log 2>&1
export PCL_TUNE_FUSED_STEP=0
rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/sq_2p_a -- python3 ${CMD[exact_dense]} --steps 3 --warmup 1 > $OUT/sq_2p_a.log 2>&1
rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/sq_2p_b -- python3 ${CMD[exact_dense]} --steps 3 --warmup 1 > $OUT/sq_2p_b.log 2>&1
export PCL_LIB_OVERRIDE=$R/build/libs/libpyclaw_amd_ypad.so
rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/sq_2p_b_ypad -- python3 ${CMD[exact_dense]} --steps 3 --warmup 1 > $OUT/sq_2p_b_ypad.log 2>&1
unset PCL_LIB_OVERRIDE
export PCL_TUNE_FUSED_STEP=1
echo "sq done"
python3 - $OUT $TAG $COMMIT <<'PY'
import csv, glob, json, sys, collections
out, tag, commit = sys.argv[1], sys.argv[2], sys.argv[3]
with open("%s/summary/%s_pmc_sq.txt" % (out, tag), "w") as fo:
    fo.write("rocprofv3 --pmc <SQ counters> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-states "
             "--state dense   (commit %s)\n  sq_a / sq_b: --math exact (the one-kernel step), two passes of 8 counters;  sq_fast_a: --math fast;\n"
             "  sq_sharp_a: bench.py --solver sharpclaw (the SharpClaw right-hand side);\n"
             "  sq_2p_a / sq_2p_b: PCL_TUNE_FUSED_STEP=0, the x pass + y pass form of the same step;\n"
             "  sq_2p_b_ypad: the sq_2p_b counters with build/libs/libpyclaw_amd_ypad.so (-DPCL_YTILE_PAD=1: the y-pass tile padded to 17 "
             "doubles per row instead of XOR-swizzled) -- the before / after pair for SQ_LDS_BANK_CONFLICT\n" % commit)
    for sub in ("sq_a", "sq_b", "sq_fast_a", "sq_sharp_a", "sq_2p_a", "sq_2p_b", "sq_2p_b_ypad"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "sweep_kernel" not in k and "step2ds_kernel" not in k and "sharp_kernel" not in k: continue
                k = k[:80]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(k, r["Counter_Name"])] += 1
        fo.write("== %s\n" % sub)
        for k in acc:
            fo.write(k + "\n")
            for c, v in sorted(acc[k].items()):
                fo.write("   %-24s %.5g per launch (%d launches)\n" % (c, v / n[(k, c)], n[(k, c)]))
print(open("%s/summary/%s_pmc_sq.txt" % (out, tag)).read()[:6000])
PY
ls $OUT/summary
