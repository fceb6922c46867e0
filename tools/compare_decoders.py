#!/usr/bin/env python3
"""Decode rate and agreement of the fast decoders against the bit-exact sum-product decoder on the SAME frames, all 20
modes, at each mode's operating point (tests/conftest.py OPERATING_ESN0) and 1.5 dB below it (inside the waterfall).
Frames are generated on the device (mgpu_txgen); every decoder sees the same samples.
  python tools/compare_decoders.py [frames_per_mode=4096] > gpurun_out/compare_decoders.json     (GPU box)
Reported per mode and point: decoded fraction and average iterations per decoder, and for each fast decoder the number of
frames it decodes to a DIFFERENT payload than the reference decoder although both report success (must be 0)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import OPERATING_ESN0  # noqa: E402
from mercury_amd import DEC_MINSUM, DEC_SPA, DEC_SPA_FAST, RxPhy  # noqa: E402

SEED = 0x4D455243


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cfgs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(17)) + [100, 101, 102]
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    res = {"frames": F, "modes": {}}
    for cfg in cfgs:
        agc, vs = (0, 0) if cfg in (15, 16) else (1, 1)
        phys = {n: RxPhy(cfg, max_iters=50, decoder=d, agc=agc, variance_source=vs, device=0, max_batch=F)
                for n, d in (("spa", DEC_SPA), ("spa_fast", DEC_SPA_FAST), ("minsum", DEC_MINSUM))}
        rx = phys["spa"]
        bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
        sent = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
        m = {}
        for label, off in (("operating", 0.0), ("waterfall", -1.5)):
            esn0 = OPERATING_ESN0[cfg] + off
            amp = float(10.0 ** (-esn0 / 20.0) / np.sqrt(2.0))
            rx.txgen_dev(SEED, (cfg + 1) << 32, F, amp, bb.data_ptr(), sent.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            outs = {}
            for n, phy in phys.items():
                payload = torch.zeros((F, rx.payload_stride), dtype=torch.uint8, device=dev)
                stats = torch.zeros((F, 6), dtype=torch.int32, device=dev)
                phy.enable_timing(True)
                phy.receive_dev(bb.data_ptr(), F, payload.data_ptr(), stats.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                _, dec_ms, _ = phy.kernel_ms_avg()
                phy.enable_timing(False)
                outs[n] = (payload.cpu().numpy(), stats.cpu().numpy(), dec_ms)
            r = {"esn0_db": esn0}
            ref_ok = outs["spa"][1][:, 3] != 0
            nb = rx.payload_bytes
            truth = sent.cpu().numpy()[:, :nb]
            for n, (pl, st, ms) in outs.items():
                ok = st[:, 3] != 0
                r[n] = {"decoded_fraction": float(ok.mean()), "avg_iters": float(np.minimum(st[:, 0], 50).mean()), "ldpc_ms": ms,
                        "decoded_but_wrong_payload": int((ok & (pl[:, :nb] != truth).any(axis=1)).sum())}
                if n != "spa":
                    both = ok & ref_ok
                    r[n]["both_decode"] = int(both.sum())
                    r[n]["payload_differs_where_both_decode"] = int((pl[both] != outs["spa"][0][both]).any(axis=1).sum())
                    r[n]["decoded_fraction_minus_spa"] = float(ok.mean() - ref_ok.mean())
            m[label] = r
            print("cfg %3d %-9s Es/N0 %6.1f  spa %.4f (%.1f it)  spa_fast %.4f (%.1f it, diff %d)  minsum %.4f (%.1f it, diff %d)" % (
                cfg, label, esn0, r["spa"]["decoded_fraction"], r["spa"]["avg_iters"], r["spa_fast"]["decoded_fraction"], r["spa_fast"]["avg_iters"],
                r["spa_fast"]["payload_differs_where_both_decode"], r["minsum"]["decoded_fraction"], r["minsum"]["avg_iters"],
                r["minsum"]["payload_differs_where_both_decode"]), file=sys.stderr)
        res["modes"][str(cfg)] = m
        for phy in phys.values():
            phy.close()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
