#!/usr/bin/env python3
"""Kernel-only timings of the synchroniser building blocks on W capture windows."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mercury_amd import RxPhy  # noqa: E402


def main():
    cfg, W = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 256
    rx = RxPhy(cfg, max_batch=1)
    n = rx.Nofdm * 85 * 4
    rng = np.random.default_rng(0)
    wins = rng.standard_normal((W, n)) * 0.05
    fc = 48000.0 * 50.0 / 256 / 4 / 2 + 300
    res = {"windows": W, "window_samples": n}
    bbi = rx.passband_to_baseband(wins, fc, which=0)
    bbi = rx.passband_to_baseband(wins, fc, which=0)
    ms = rx.last_sync_kernel_ms()
    res["p2b_full_ms"] = ms
    res["p2b_full_GBps"] = W * n * (8 + 16) / ms / 1e6
    nfr = (rx.preamble_nsymb + rx.Nsymb) * rx.Nofdm
    rx.passband_to_baseband(wins, fc, which=1, start=np.full(W, 5000, np.int32), count=nfr, decimation=4)
    res["p2b_extract_ms"] = rx.last_sync_kernel_ms()
    rx.time_sync_preamble(bbi, 100)
    res["tsync_coarse_ms"] = rx.last_sync_kernel_ms()
    sym = rx.Nofdm * 4
    rx.time_sync_preamble(bbi[:, : (rx.preamble_nsymb + 4) * sym], 1, 0, 2)
    res["tsync_fine_ms"] = rx.last_sync_kernel_ms()
    rx.freq_sync(bbi[:, ::4][:, : 4 * rx.Nofdm])
    res["moose_ms"] = rx.last_sync_kernel_ms()
    tot = res["p2b_full_ms"] + res["p2b_extract_ms"] + res["tsync_coarse_ms"] + res["tsync_fine_ms"] + res["moose_ms"]
    res["windows_per_s_kernels_only"] = W / tot * 1e3
    # universal ACK / BREAK tone-pattern detector (every mode) on the same windows
    rx.detect_ack_pattern(bbi, 1)
    rx.detect_ack_pattern(bbi, 1)
    res["ack_detect_slot_energy_ms"] = rx.last_sync_kernel_ms()
    rx.close()
    # MFSK time sync: ROBUST_0 frames are 324 symbols, the reference's capture buffer holds two of them
    rx = RxPhy(100, max_batch=1)
    Wm = max(1, W // 8)
    nm = rx.Nofdm * 2 * (rx.Nsymb + rx.preamble_nsymb) * 4
    bbm = (rng.standard_normal((Wm, nm)) + 1j * rng.standard_normal((Wm, nm))) * 0.05
    rx.time_sync_mfsk(bbm)
    rx.time_sync_mfsk(bbm)
    res["mfsk_windows"] = Wm
    res["mfsk_window_samples"] = nm
    res["mfsk_tsync_slot_energy_ms"] = rx.last_sync_kernel_ms()
    res["mfsk_tsync_GBps"] = Wm * nm * 16 / res["mfsk_tsync_slot_energy_ms"] / 1e6
    print(json.dumps(res))


if __name__ == "__main__":
    main()
