#!/bin/bash
# Regenerate the judged profile artefacts on the GPU box (then copy gpurun_out/prof_<tag>/summary/* into profiles/).
#   * rocprofv3 --kernel-trace --stats of the bench command, per arithmetic mode and per state (the bench JSON of the SAME
#     command is saved next to the kernel stats, so the two average durations can be compared);
#   * in SEPARATE passes the HBM PMC counters FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md: KB units; FETCH_SIZE x2 on
#     gfx950) of the same commands, of the capacity-function unsplit step (tools/kbench.py --child --unsplit --capa) and
#     of the sphere app (bench.py --app sphere);
#   * the SQ issue/wait counters of the dense-state kernels (own passes).
# Usage: tools/profile_round.sh <tag> <commit>      (run through gpurun; rocprofv3 launches python3 directly)
set -e
TAG=${1:-r02}
COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-states"
run_stats() {   # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -- python3 $R/bench.py --steps 20 --warmup 5 $B "$@" \
      > $OUT/summary/${TAG}_bench_$name.json 2> $OUT/stats_$name.err
  cp $(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1) $OUT/summary/${TAG}_bench_kernel_stats_$name.csv
  echo "stats $name done"
}
run_pmc() {     # name, command after python3 ...
  local name=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${name}_$C -- python3 "$@" > $OUT/pmc_${name}_$C.log 2>&1
  done
  echo "pmc $name done"
}
run_stats exact --math exact
run_stats fast --math fast
run_stats exact_dense --math exact --state dense
run_stats fast_dense --math fast --state dense
run_stats exact_developed --math exact --state developed
run_stats exact_unsplit --math exact --unsplit
run_pmc exact $R/bench.py --steps 3 --warmup 1 $B --math exact
run_pmc fast $R/bench.py --steps 3 --warmup 1 $B --math fast
run_pmc exact_dense $R/bench.py --steps 3 --warmup 1 $B --math exact --state dense
run_pmc exact_unsplit $R/bench.py --steps 3 --warmup 1 $B --math exact --unsplit
run_pmc exact_unsplit_capa $R/tools/kbench.py --child --unsplit --capa --state random --reps 3 --warm 1 --math exact
run_pmc exact_sphere $R/bench.py --app sphere --steps 3 --warmup 1 --math exact
# SQ counters, dense state (VALU-bound regime)
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
SQB="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
rocprofv3 --pmc $SQA --kernel-trace --output-format csv -d $OUT/sq_a -- python3 $R/bench.py --steps 3 --warmup 1 $B --math exact --state dense > $OUT/sq_a.log 2>&1
rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/sq_b -- python3 $R/bench.py --steps 3 --warmup 1 $B --math exact --state dense > $OUT/sq_b.log 2>&1
echo "sq done"
python3 - $OUT $TAG $COMMIT <<'PY'
import csv, glob, json, sys, collections
out, tag, commit = sys.argv[1], sys.argv[2], sys.argv[3]
res = {"commit": commit,
       "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python3 <bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-states ... | tools/kbench.py --child ...>",
       "correction": "MI355X_MICROARCH.md HBM section: counters are in KB; on gfx950 FETCH_SIZE reads 1/2 of streamed bytes -> x2; WRITE_SIZE exact",
       "modes": {}}
ALG = {"exact": 80 * 4096 * 4096, "fast": 80 * 4096 * 4096, "exact_dense": 80 * 4096 * 4096,
       "exact_unsplit": None, "exact_unsplit_capa": None, "exact_sphere": None}
for name in ALG:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, name, c), recursive=True):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if ("sweep_kernel" in k or "unsplit_" in k) and r["Counter_Name"] == c:
                    per[(k, r["Dispatch_Id"])] += float(r["Counter_Value"])
            for (k, _), v in per.items():
                acc[k][c].append(v)
    res["modes"][name] = {}
    for k, d in acc.items():
        f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
        w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
        e = {"FETCH_SIZE_KB_avg_per_launch": f, "WRITE_SIZE_KB_avg_per_launch": w,
             "launches_sampled": [len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"])],
             "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024}
        if ALG[name]:
            e["algorithmic_bytes_per_launch"] = ALG[name]
        res["modes"][name][k] = e
json.dump(res, open("%s/summary/%s_pmc_hbm.json" % (out, tag), "w"), indent=1)
with open("%s/summary/%s_pmc_sq_dense.txt" % (out, tag), "w") as fo:
    fo.write("rocprofv3 --pmc <SQ counters> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
             "--no-states --math exact --state dense   (commit %s; two passes of 8 counters)\n" % commit)
    for sub in ("sq_a", "sq_b"):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
        for f in glob.glob(out + "/%s/**/*counter_collection.csv" % sub, recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "sweep_kernel" not in k: continue
                k = k[:80]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(k, r["Counter_Name"])] += 1
        for k in acc:
            fo.write(k + "\n")
            for c, v in sorted(acc[k].items()):
                fo.write("   %-24s %.5g per launch (%d launches)\n" % (c, v / n[(k, c)], n[(k, c)]))
print(json.dumps(res["modes"], indent=1)[:3000])
print(open("%s/summary/%s_pmc_sq_dense.txt" % (out, tag)).read())
PY
