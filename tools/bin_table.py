#!/usr/bin/env python3
"""Per-bin table of the fp64 sum-product kernel's layout (first-fit decreasing of whole checks into 64-slot bins, tables.cpp: load_graph)
for one LDPC rate: the checks' degrees, lanes in use, product-walk steps (= the largest degree).   tools/bin_table.py [K=1400]"""
import struct
import sys
import os

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
b = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mercury_amd", "data", "mercury_ldpc_tables.bin"), "rb").read()
assert b[:4] == b"MLDP"
nr = struct.unpack_from("<I", b, 8)[0]
off = 12
for _ in range(nr):
    k, P, N, E, cw, vw = struct.unpack_from("<6I", b, off)
    off += 24
    cdeg = list(b[off:off + P])
    off += P + 2 * E + N + 2 * E
    if k != K:
        continue
    fill, mem = [], []
    for d in sorted(cdeg, reverse=True):
        for i in range(len(fill)):
            if fill[i] + d <= 64:
                fill[i] += d
                mem[i].append(d)
                break
        else:
            fill.append(d)
            mem.append([d])
    print("rate %d/1600: %d checks, %d edges, %d bins (%.1f bins' worth of edges); lanes in use %.1f %%; sum of walk steps %d (mean %.1f per bin)"
          % (K, P, E, len(fill), E / 64.0, 100.0 * E / (64 * len(fill)), sum(max(m) for m in mem), sum(max(m) for m in mem) / len(mem)))
    big = sum(1 for d in cdeg if d > 32)
    print("checks of more than 32 edges (no two share a bin): %d; of exactly 32: %d; smaller: %d" % (big, sum(1 for d in cdeg if d == 32), sum(1 for d in cdeg if d < 32)))
    print("useful multiplications / multiplications issued (64 lanes x walk steps): %.3f" % (sum(d * (d - 1) for d in cdeg) / float(sum(64 * max(m) for m in mem))))
    print("bin  checks (degrees)            lanes  walk steps")
    for i, m in enumerate(mem):
        print("%3d  %-28s %5d  %5d" % (i, " ".join(map(str, m)), sum(m), max(m)))
