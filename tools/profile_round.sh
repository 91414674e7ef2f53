#!/bin/bash
# Regenerate the judged profile artefacts on the GPU box (then copy gpurun_out/prof_<tag>/summary/* into profiles/):
#   kernel-trace/stats of the default bench line (both arithmetic modes) and, in SEPARATE passes, the HBM
#   PMC counters FETCH_SIZE and WRITE_SIZE (MI355X_MICROARCH.md: KB units; FETCH_SIZE x2 on gfx950).
# Usage: tools/profile_round.sh <tag>            (run through gpurun; rocprofv3 launches python3 directly)
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
for MODE in exact fast; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$MODE -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --math $MODE > $OUT/summary/${TAG}_bench_$MODE.json 2> $OUT/stats_$MODE.err
  cp $(find $OUT/stats_$MODE -name "*kernel_stats.csv" | head -1) $OUT/summary/${TAG}_bench_kernel_stats_$MODE.csv
  echo "stats $MODE done"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_${MODE}_$C -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --math $MODE > $OUT/pmc_${MODE}_$C.log 2>&1
    echo "pmc $MODE $C done"
  done
done
python3 $R/bench.py --extras > $OUT/summary/${TAG}_bench_extras.json 2> $OUT/extras.err
echo "extras done"
python3 - $OUT $TAG <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
res = {"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --math <mode>",
       "correction": "MI355X_MICROARCH.md HBM section: counters are in KB; on gfx950 FETCH_SIZE reads 1/2 of streamed bytes -> x2; WRITE_SIZE exact",
       "modes": {}}
for mode in ("exact", "fast"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, mode, c), recursive=True):
            per = collections.defaultdict(float)
            for r in csv.DictReader(open(f)):
                if "sweep_kernel" in r["Kernel_Name"] and "pcl::%s::" % mode in r["Kernel_Name"] and r["Counter_Name"] == c:
                    per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
            for (k, _), v in per.items():
                acc[k][c].append(v)
    res["modes"][mode] = {}
    for k, d in acc.items():
        f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
        w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
        res["modes"][mode][k] = {"FETCH_SIZE_KB_avg_per_launch": f, "WRITE_SIZE_KB_avg_per_launch": w,
                                 "launches_sampled": [len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"])],
                                 "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024,
                                 "algorithmic_bytes_per_launch": 80 * 4096 * 4096}
json.dump(res, open("%s/summary/%s_pmc_hbm.json" % (out, tag), "w"), indent=1)
print(json.dumps(res["modes"], indent=1)[:1500])
PY
