"""What cl_telecom_system::receive_byte leaves UNWRITTEN (Missing 4).

The receive_stats member survives from call to call and receive_byte assigns several of its fields only on the paths that reach them
(telecom_system.cc:646-1503): a noise-only window after a decoded one still reports the decoded window's iterations_done / SNR / crc, the
caller's `out` array keeps the previous payload, freq_offset keeps the last decoded OFDM frame's. The earlier pins reset every field before
each call (oracle/ref_ts_harness.cc:mrefts_receive_byte); here the reference's real object runs >= 12 consecutive windows - decoded,
noise only, cut off by the window's end, silence, decoded again ... - with NO reset in between (mrefts_receive_byte_raw), and after every call
every field of the struct it returned and of the member it holds is compared with

  * (CPU) the oracle's stateless call (morc_receive_byte) folded into held state by the path table below - the same table as
    mgpu::detail::apply_receive_byte (include/mercury_gpu.hpp) and INTEGRATION.md 1.3b;
  * (GPU) the reference's object with receive_byte replaced by mgpu_receive_byte_batch (oracle/ref_ts_gpu_harness.cc, mode WHOLE), and by
    the product's C++ mirror mgpu::cl_rx_phy::receive_byte running on its own receive_stats (mode MIRROR).

Fields the reference's constructor never initialises (crc, all_zeros, coarse_metric: telecom_system.cc:38-53) hold heap garbage until their
first write and are compared from then on."""
import numpy as np
import pytest

import oraclelib
from oraclelib import CARRIER, LinkState, Oracle, RefTelecomSystem, RefTelecomSystemGpu

pytestmark = pytest.mark.skipif(not RefTelecomSystem.available(), reason="oracle/_ref/libmercury_ref_ts.so not built (needs /root/reference)")

INTS = RefTelecomSystem.RAW_INTS
DOUBLES = RefTelecomSystem.RAW_DOUBLES
MODE_WHOLE, MODE_MIRROR = 4, 8


def sequence(orc, seed, noise=0.02):
    """>= 12 consecutive capture windows of one receiver: (kind, samples)"""
    n = orc.buffer_samples()
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    rng = np.random.default_rng(seed)
    kinds = ["frame", "noise", "frame", "edge", "silence", "frame", "noise", "noise", "weak", "frame", "silence", "edge", "frame", "noise"]
    for kind in kinds:
        x = rng.standard_normal(n) * (1e-10 if kind == "silence" else noise)
        pl = rng.integers(0, 256, orc.payload_bytes)
        pb = orc.transmit_byte(pl.astype(np.int32), message_location=3) * float(rng.uniform(0.8, 2.0))
        if kind == "frame":
            d = int(rng.integers(0, n - used))
            x[d: d + used] += pb[:used]
        elif kind == "weak":                                   # found by the synchroniser, reaches the decoder, fails the CRC
            d = int(rng.integers(0, n - used))
            x[d: d + used] += pb[:used]
            x += rng.standard_normal(n) * (2.0 if orc.mfsk_M else 0.6)
        elif kind == "edge":
            d = n - int(rng.integers(used // 4, used // 2))
            x[d:] += pb[: n - d]
        yield kind, x


def fold(held, out, r, mfsk, payload_bytes):
    """one stateless call's results -> the held st_receive_stats, path by path (= mgpu::detail::apply_receive_byte)"""
    attempted = r["iterations_done"] != -1 or r["message_decoded"]
    for k in ("message_decoded", "frame_overflow_symbols", "sync_trials", "delay"):
        held[k] = r[k]
    held["signal_stregth_dbm"] = r["signal_strength_dbm"]
    if not mfsk:
        held["coarse_metric"] = r["coarse_metric"]
    if attempted:
        for k in ("iterations_done", "crc", "all_zeros"):
            held[k] = r[k]
        held["SNR"] = r["snr_db"]
        out[:payload_bytes] = r["payload"]
    if r["message_decoded"] and not mfsk:
        held["freq_offset"] = r["freq_offset"]
    held["delay_of_last_decoded_message"] = r["state"].delay_of_last_decoded_message
    held["freq_offset_of_last_decoded_message"] = r["state"].freq_offset_of_last_decoded_message
    return attempted


def comparable(k, written):
    return k not in ("crc", "all_zeros", "coarse_metric") or k in written


def same(a, b):
    return a == b or (a != a and b != b)


@pytest.mark.parametrize("cfg", [0, 8, 11, 13, 16, 100, 101, 102])
def test_consecutive_windows_without_resets_oracle_plus_path_table_equals_the_real_object(cfg):
    ref, orc = RefTelecomSystem(cfg), Oracle(cfg)
    mfsk = bool(orc.mfsk_M)
    held = dict(iterations_done=-1, delay=0, delay_of_last_decoded_message=-1, sync_trials=0, message_decoded=0, crc=None, all_zeros=None,
                mfsk_search_raw=0, frame_overflow_symbols=0, freq_offset=0.0, freq_offset_of_last_decoded_message=0.0, SNR=-99.9,
                signal_stregth_dbm=-999.0, coarse_metric=None)          # telecom_system.cc:38-53
    out = np.full(1600, -1, np.int32)
    written = set()
    stale_seen = {"iterations_done": 0, "SNR": 0, "out": 0}
    kinds = []
    for w, (kind, x) in enumerate(sequence(orc, 600 + cfg, 0.003 if cfg >= 14 else 0.02)):
        st = LinkState(held["delay_of_last_decoded_message"], held["freq_offset_of_last_decoded_message"], 0, 0)
        r = orc.receive_byte(x, carrier=CARRIER, state=st)
        before = dict(held)
        attempted = fold(held, out, r, mfsk, orc.payload_bytes)
        if attempted:
            written |= {"crc", "all_zeros"}
        if not mfsk:
            written.add("coarse_metric")
        ret, real, real_out = ref.receive_byte_raw(x)
        for k in INTS + DOUBLES:
            if comparable(k, written):
                assert same(real[k], held[k]), (cfg, w, kind, "held", k, real[k], held[k])
                assert same(ret[k], held[k]), (cfg, w, kind, "returned", k, ret[k], held[k])
        assert np.array_equal(real_out[: orc.payload_bytes], out[: orc.payload_bytes]), (cfg, w, kind)
        if not attempted and before["iterations_done"] != -1:           # the case the earlier pins could not see
            stale_seen["iterations_done"] += real["iterations_done"] == before["iterations_done"] != -1
            stale_seen["SNR"] += real["SNR"] == before["SNR"]
            stale_seen["out"] += int(real_out[0] != -1)
        kinds.append((kind, int(attempted), r["message_decoded"]))
    assert sum(d for _, _, d in kinds) >= 2 and sum(1 - a for _, a, _ in kinds) >= 2, kinds
    assert min(stale_seen.values()) >= 1, (stale_seen, kinds)            # un-attempted windows really did keep the previous call's values
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [MODE_WHOLE, MODE_MIRROR])
@pytest.mark.parametrize("cfg", [0, 8, 11, 13, 16, 100, 101, 102])
def test_consecutive_windows_without_resets_gpu_backed_receive_byte_equals_the_real_object(cfg, mode):
    if not RefTelecomSystemGpu.available():
        pytest.skip("oracle/_ref/libmercury_ref_ts_gpu.so not built")
    a, b = RefTelecomSystem(cfg), RefTelecomSystemGpu(cfg, mode)
    orc = Oracle(cfg)
    written = set()
    decoded = 0
    for w, (kind, x) in enumerate(sequence(orc, 600 + cfg, 0.003 if cfg >= 14 else 0.02)):
        ra, ha, oa = a.receive_byte_raw(x)
        rb, hb, ob = b.receive_byte_raw(x)
        if ha["iterations_done"] != -1:
            written |= {"crc", "all_zeros"}
        if not orc.mfsk_M:
            written.add("coarse_metric")
        for k in INTS + DOUBLES:
            if not comparable(k, written):
                continue
            exact = k in INTS or k in ("SNR",)
            for name, p, q in (("held", ha, hb), ("returned", ra, rb)):
                assert same(p[k], q[k]) or (not exact and abs(p[k] - q[k]) <= 1e-9 * max(1.0, abs(p[k]))), (cfg, mode, w, kind, name, k, p[k], q[k])
        assert np.array_equal(oa, ob), (cfg, mode, w, kind)
        decoded += ha["message_decoded"]
    assert decoded >= 2 and b.counters()["receive_byte"][1] == 14 and b.error() == ""
    a.close(); b.close()
