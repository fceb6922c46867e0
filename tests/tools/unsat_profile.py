#!/usr/bin/env python3
"""How many parity checks are odd after each sum-product iteration (checker-side study behind the fp64 decoder's adaptive look policy,
mercury_amd/csrc/ldpc.hip "adaptive", profiles/NOTES.md R6.8).

Frames of one mode at one Es/N0 go through the ORACLE's front-end (tests/oraclelib.py; this script lives under tests/ because it uses the
checker); a plain numpy flooding sum-product decoder (ldpc_decoder_SPA.cc:127-210's rule, not its rounding: statistics only) then records
the syndrome's weight after every iteration, next to what the kernel's sample (the 16 first bins of the first-fit-decreasing layout,
tables.cpp) would have counted.

    python tests/tools/unsat_profile.py <cfg> <EsN0 dB> [frames=40]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oraclelib as O  # noqa: E402


def first_bins(cdeg, nfirst=16):
    """Checks of bins 0..nfirst-1 of the kernel's layout: whole checks, by descending degree, first fit into 64-slot bins."""
    fill, mem = [], []
    for c in np.argsort(-cdeg, kind="stable"):
        d = cdeg[c]
        for i in range(len(fill)):
            if fill[i] + d <= 64:
                fill[i] += d
                mem[i].append(c)
                break
        else:
            fill.append(d)
            mem.append([c])
    return len(fill), np.array(sum(mem[:nfirst], []))


def main():
    cfg, es = int(sys.argv[1]), float(sys.argv[2])
    nf = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    o = O.Oracle(cfg, 50)
    checks, _ = O.ldpc_graph(o.K)
    P, N = len(checks), 1600
    cdeg = np.array([len(c) for c in checks])
    nbins, first = first_bins(cdeg)
    ce = np.concatenate([np.full(len(c), i) for i, c in enumerate(checks)])
    ve = np.concatenate(checks)
    E = len(ce)
    print("cfg %d: K %d, %d checks, %d edges, %d bins; %d checks in bins 0..15" % (cfg, o.K, P, E, nbins, len(first)))
    na = O.noise_amp_for(es)
    its, last = [], []

    def weight(post):
        syn = np.bincount(ce, weights=(post < 0)[ve].astype(float), minlength=P).astype(int) & 1
        return int(syn.sum()), int(syn[first].sum())

    for f in range(nf):
        bb, _ = o.gen_frame(0x4D455243, f, na)
        llr = np.array(o.rx(bb)["llr_ldpc"], np.float64)
        R = np.zeros(E)
        post = llr.copy()
        seq = [weight(post)]
        done = 0
        if seq[0][0]:
            done = 51
            for it in range(1, 51):
                T = np.tanh(0.5 * (post[ve] - R))
                neg = T < 0
                lm = np.log(np.maximum(np.abs(T), 1e-300))
                tot = np.bincount(ce, weights=lm, minlength=P)
                sg = np.bincount(ce, weights=neg.astype(float), minlength=P).astype(int) & 1
                prod = np.exp(tot[ce] - lm) * np.where(sg[ce] ^ neg, -1.0, 1.0)
                R = 2 * np.arctanh(np.clip(prod, -0.9999999, 0.9999999))
                post = llr + np.bincount(ve, weights=R, minlength=N)
                seq.append(weight(post))
                if seq[-1][0] == 0:
                    done = it
                    break
        its.append(done)
        if done and done <= 50:
            last.append(seq[-2][0])
        print("frame %3d: %2d iterations; odd checks (sampled) %s" % (f, done, " ".join("%d(%d)" % s for s in seq[:12])))
    print("iterations: mean %.2f, histogram %s" % (np.mean(its), np.bincount(its).tolist()))
    if last:
        print("largest weight a frame converged from in ONE iteration: %d (five largest: %s)" % (max(last), sorted(last)[-5:]))


if __name__ == "__main__":
    main()
