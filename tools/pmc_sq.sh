#!/bin/bash
# SQ issue/wait breakdown of the sweep kernels (own PMC pass, kernel-trace only).  Usage: tools/pmc_sq.sh <math> <outdir>
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${2:-pmc_sq}
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES \
  --kernel-trace --output-format csv -d $OUT/a -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --math $1 > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $OUT/b -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --math $1 > $OUT/b.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sweep_kernel" not in k: continue
            k = k[:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(k)
        for c, v in sorted(acc[k].items()):
            print("   %-24s %.4g per launch (%d launches)" % (c, v / n[(k, c)], n[(k, c)]))
PY
