#!/usr/bin/env python
"""
tools/pmc_summary.py <dir> -- HBM bytes per launch of every kernel found in the rocprofv3 PMC passes under <dir>
(sub-directories pmc_<name>_FETCH_SIZE and pmc_<name>_WRITE_SIZE, one counter per pass as MI355X_MICROARCH.md
prescribes).  Counters are in KB; on gfx950 FETCH_SIZE reads half of the streamed bytes (x2), WRITE_SIZE is exact.
Prints JSON: {name: {kernel: {FETCH_KB, WRITE_KB, launches, hbm_bytes_per_launch}}}.
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    names = sorted(set(os.path.basename(d)[4:].rsplit("_", 2)[0] for d in glob.glob(out + "/pmc_*_SIZE")))
    res = {}
    for name in names:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, name, c), recursive=True):
                per = collections.defaultdict(float)
                for r in csv.DictReader(open(f)):
                    if r["Counter_Name"] == c and "pcl::" in r["Kernel_Name"]:
                        per[(r["Kernel_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
                for (k, _), v in per.items():
                    acc[k][c].append(v)
        res[name] = {}
        for k, d in acc.items():
            f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"]))
            w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
            if 2 * f + w < 1024:          # small helper kernels (< 1 MB per launch)
                continue
            res[name][k.split("(")[0][-110:]] = {"FETCH_SIZE_KB_avg_per_launch": f, "WRITE_SIZE_KB_avg_per_launch": w,
                                                 "launches_sampled": [len(d["FETCH_SIZE"]), len(d["WRITE_SIZE"])],
                                                 "fetch_bytes_corrected": 2 * f * 1024, "write_bytes": w * 1024,
                                                 "hbm_bytes_per_launch_corrected": (2 * f + w) * 1024}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
