#!/usr/bin/env python3
"""Static instruction counts of a kernel's loops from the compiler's assembly (no GPU needed): a first look at what a source change does to a
hot loop before it goes to a box. Compiles one translation unit of mercury_amd/csrc for gfx950 with the product's flags, then counts per
depth-2 loop (the decoders' bin loops) or per basic block (--blocks N: blocks with >= N vector instructions) vector / fp64 / compare-select-move /
integer / scalar / LDS / vector-memory instructions, plus the kernel's VGPR count, scratch bytes and spill instructions.
    tools/isa_count.py ldpc.hip mgpu_ldpc_spa_kernel_ne6
    tools/isa_count.py frontend.hip mgpu_frontend_kernel --blocks 25
All branch paths of a loop are counted (a wavefront whose lanes spread over the cases issues all of them); the dynamic figure is the PMC's."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mercury_amd.build import HIPCC_FLAGS, _hipcc  # noqa: E402


def assembly(unit):
    out = os.path.join(tempfile.gettempdir(), "isa_" + unit + ".s")
    subprocess.run([_hipcc()] + HIPCC_FLAGS + ["-I", os.path.join(ROOT, "mercury_amd", "csrc"), "--cuda-device-only", "-S", "-o", out,
                    os.path.join(ROOT, "mercury_amd", "csrc", unit)], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def classify(op, c):
    if op.startswith("v_"):
        c["valu"] += 1
        if "f64" in op:
            c["fp64"] += 1
        elif op.startswith(("v_cmp", "v_cndmask", "v_mov", "v_readfirstlane", "v_readlane", "v_bfrev")):
            c["cmp_sel_mov"] += 1
        else:
            c["int"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        c["vmem"] += 1
        if op.startswith("scratch_"):
            c["spill"] += 1


def main():
    unit, kern = sys.argv[1], sys.argv[2]
    blocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 0
    lines = assembly(unit)
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    foot = "\n".join(lines[end:end + 60])
    print(kern, {k: re.search(r"; %s: (\d+)" % k, foot).group(1) for k in ("NumVgprs", "ScratchSize", "Occupancy") if re.search(r"; %s: (\d+)" % k, foot)})
    groups = collections.OrderedDict()
    key = "entry"
    total = collections.Counter()
    for l in lines[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
            m = re.search(r"Header=(BB\d+_\d+) Depth=2", l)
            key = l.split(":")[0].strip() if blocks else (m.group(1) if m else None)
            continue
        if "Inner Loop Header: Depth=2" in l and not blocks:
            continue
        t = l.strip()
        if not t or t[0] in ";.":
            continue
        classify(t.split()[0], total)
        if key:
            classify(t.split()[0], groups.setdefault(key, collections.Counter()))
    for k, c in groups.items():
        if not blocks or c["valu"] >= blocks:
            print("%-12s" % k, dict(c))
    print("whole kernel", dict(total))


if __name__ == "__main__":
    main()
