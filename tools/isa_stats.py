#!/usr/bin/env python
"""
isa_stats.py -- static instruction histogram of one kernel in a hipcc -save-temps .s file.

  python tools/isa_stats.py kernels-hip-amdgcn-amd-amdhsa-gfx950.s 'sweep_kernelINS0_6Euler5ELi2ELb0ELb0ELb0ELb0'

Prints the opcode histogram grouped by class (f64 VALU, DPP moves, other VALU, LDS, VMEM, SALU), the
register/occupancy lines of the kernel descriptor and -- with a weight table from
tools/ubench/valu_rates.hip -- an estimated issue-cycle total.  Static counts: every unrolled copy of
the per-strip body is counted once, so divide by the number of strips a wavefront walks.
"""
import collections
import re
import sys

QUARTER = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64",
           "v_trig_preop_f64", "v_frexp")


def classify(op, line):
    if op.startswith("v_") and "dpp" in line:
        return "dpp"
    if op.startswith("v_") and ("_f64" in op):
        return "f64q" if op.startswith(QUARTER) else "f64"
    if op.startswith("v_cndmask") or op.startswith("v_cmp") or op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "vmisc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    S = open(path).read().split("\n")
    start = None
    for i, l in enumerate(S):
        if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = start
    while not S[end].startswith(".Lfunc_end"):
        end += 1
    ops = collections.Counter()
    cls = collections.Counter()
    for l in S[start + 1:end]:
        t = l.strip()
        if not t or t[0] in ".;/" or t.endswith(":"):
            continue
        op = t.split()[0]
        ops[op] += 1
        cls[classify(op, t)] += 1
    print(S[start].split(":")[0])
    for k, v in cls.most_common():
        print("  %-6s %6d" % (k, v))
    print("  total  %6d" % sum(cls.values()))
    for l in S[end:end + 60]:
        if re.search(r"NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|NumSgprs|TotalNumVgprs", l):
            print(" ", l.strip())
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for op, v in ops.most_common(n):
        print("    %-28s %6d" % (op, v))


if __name__ == "__main__":
    main()
