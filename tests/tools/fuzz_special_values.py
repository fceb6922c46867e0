#!/usr/bin/env python3
"""Special-value fuzz of the GPU path against the CPU oracle (test infrastructure: the oracle is the checker).
 (1) cl_ldpc::decode (mgpu_ldpc_batch, the reference's fp64 sum-product) on LLR words salted with +-Inf, NaN, +-0, float denormals,
     FLT_MAX-scale values and exact ties: hard bits and iteration counts must equal the oracle's (whatever libm's tanh / atanh make of them).
 (2) the zero-forcing modes (15, 16) in the receive_byte variant (agc = 1, variance from the equalised pilots: a ~1e-33 variance turns the
     LLRs into +-Inf / rounding noise - what RX_SHM really feeds the decoder in those modes): every frame's LLRs (bit pattern, NaNs in the
     same places), iteration count, CRC and payload against the oracle, from 40 dB down into the noise.
  python tests/tools/fuzz_special_values.py [words_per_rate=64] [seed=1]          (GPU box; prints one line per case, exit code 1 on a difference)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oraclelib  # noqa: E402
from mercury_amd import DEC_SPA, RxPhy  # noqa: E402
from oraclelib import FLAGS_RECEIVE_BYTE, Oracle, noise_amp_for  # noqa: E402

RATE_CFGS = [0, 1, 2, 4, 5, 8, 11, 16]      # one mode per code rate


def salted_words(rng, n, K, N):
    """LLR words: noisy BPSK-like LLRs of magnitude ~1..8 with special values scattered in."""
    specials = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 1e-45, -1e-45, 1e-38, 3.0e38, -3.0e38, 1e30, -1e30, 44.0, -44.0, 88.0, 1e-30,
                         0.69314718, -0.69314718, 2.0, -2.0, 22.0, -22.0], np.float32)
    out = []
    for w in range(n):
        scale = [0.3, 1.0, 3.0, 8.0][w % 4]
        l = (rng.standard_normal(N) * scale + scale * 0.8).astype(np.float32)       # mostly-positive word (all-zero codeword) with errors
        kind = w % 8
        if kind == 0:       # a handful of specials
            idx = rng.choice(N, 12, replace=False)
            l[idx] = rng.choice(specials, 12)
        elif kind == 1:     # many infinities of both signs
            idx = rng.choice(N, N // 4, replace=False)
            l[idx] = np.where(rng.random(idx.size) < 0.9, np.inf, -np.inf).astype(np.float32)
        elif kind == 2:     # one NaN
            l[rng.integers(N)] = np.nan
        elif kind == 3:     # exact ties and zeros
            l = np.round(l).astype(np.float32)
        elif kind == 4:     # everything huge
            l = (l * np.float32(1e37)).astype(np.float32)
        elif kind == 6:     # hard words (what the zero-forcing modes hand over): every |LLR| at or above the decoder's whole-frame threshold of 200, sign errors left in
            mag = [np.float32(200.0), np.float32(np.inf), np.float32(1e32), np.float32(250.0)][(w // 8) % 4]
            l = np.where(l < 0, -mag, mag).astype(np.float32)
        elif kind == 7:     # just not hard: one value below the threshold (still saturating tanh), or every value just below it
            if (w // 8) % 2 == 0:
                l = np.where(l < 0, np.float32(-1e30), np.float32(1e30)).astype(np.float32)
                l[rng.integers(N)] = np.float32([199.99, -60.0, 44.0, 150.0][(w // 16) % 4])
            else:
                l = np.where(l < 0, np.float32(-199.99), np.float32(199.99)).astype(np.float32)
        # kind 5: plain
        out.append(l)
    return np.stack(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bad = 0
    with np.errstate(all="ignore"):
        for cfg in RATE_CFGS:
            orc = Oracle(cfg, 50)
            rx = RxPhy(cfg, max_iters=50, decoder=DEC_SPA, max_batch=n)
            words = salted_words(rng, n, orc.K, orc.N)
            bits, iters = rx.ldpc_decode(words)
            diff = 0
            for w in range(n):
                rb, ri = orc.ldpc_decode(words[w])
                if ri != int(iters[w]) or not np.array_equal(rb.astype(np.uint8), bits[w]):
                    diff += 1
                    if diff <= 3:
                        print("   cfg %d word %d (kind %d): iterations gpu %d cpu %d, differing bits %d" % (cfg, w, w % 6, iters[w], ri, int((rb.astype(np.uint8) != bits[w]).sum())))
            print("decoder, rate of cfg %3d: %d salted words, %d differ" % (cfg, n, diff))
            bad += diff
            rx.close()
        for cfg in (15, 16):
            orc = Oracle(cfg, 50)
            snrs = [60.0, 40.0, 30.0, 24.0, 20.0, 18.0, 16.0, 14.0, 12.0, 10.0, 6.0, 0.0, -15.0] * 2
            frames = np.stack([orc.gen_frame(seed + 77, i, noise_amp_for(s))[0] for i, s in enumerate(snrs)])
            rx = RxPhy(cfg, max_iters=50, decoder=DEC_SPA, agc=1, variance_source=1, max_batch=len(snrs))
            out = rx.receive(frames, want_llr=True)
            diff = 0
            for i in range(len(snrs)):
                ref = orc.rx(frames[i], FLAGS_RECEIVE_BYTE)
                nan = np.isnan(ref["llr_ldpc"])
                got = out["llr_ldpc"][i]
                st = out["stats"][i]
                ok = (np.array_equal(np.isnan(got), nan) and np.array_equal(got[~nan].view(np.uint32), ref["llr_ldpc"][~nan].view(np.uint32))
                      and (st["iterations_done"], st["crc"], st["all_zeros"]) == (ref["iterations"], ref["crc"], ref["all_zeros"])
                      and np.array_equal(out["payload"][i], ref["bytes"].astype(np.uint8)))
                if not ok:
                    diff += 1
                    if diff <= 4:
                        print("   cfg %d frame %d (%.0f dB): iterations gpu %d cpu %d, llr equal %s, inf LLRs %d, nan %d" % (
                            cfg, i, snrs[i], st["iterations_done"], ref["iterations"],
                            np.array_equal(got[~nan].view(np.uint32), ref["llr_ldpc"][~nan].view(np.uint32)), int(np.isinf(ref["llr_ldpc"]).sum()), int(nan.sum())))
            print("receive_byte variant, ZF mode %d: %d frames (60 dB .. -15 dB), %d differ; decoded %d" % (
                cfg, len(snrs), diff, int(sum(out["stats"][i]["message_decoded"] for i in range(len(snrs))))))
            bad += diff
            rx.close()
    print("TOTAL differing: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
