"""Synchroniser building blocks in front of the RX hot path:
passband_to_baseband (mixer + FIR + decimation), Schmidl-Cox time sync, Moose frequency sync.

CPU part: the C oracle against the compiled reference (bit-identical). GPU part: the HIP kernels against
the oracle — indices and integer outputs exact; the mixed-down baseband bit-identical when the windows share a carrier (host-libm
mixer table) and within 1e-11 relative when every window has its own (device cos/sin, last ulp), plus an end-to-end capture-window chain.
"""
import numpy as np
import pytest

import oraclelib
from conftest import SEED
from oraclelib import CARRIER, Oracle, RefLib


def _window(orc, delay, noise=0.01, seed=1, carrier=CARRIER, frame_idx=1):
    payload = orc.gen_payload(SEED, frame_idx)
    bits = orc.payload_to_bits(payload)
    pb = orc.tx_passband(bits, carrier=carrier)
    W = orc.Nofdm * 85 * 4                      # buffer_Nsymb = 85 (SURVEY.md §8c anchor for cfg 8)
    rng = np.random.default_rng(seed)
    win = rng.standard_normal(W) * noise
    win[delay: delay + pb.size] += pb
    return win, payload


@pytest.mark.skipif(not RefLib.available(), reason="oracle/_ref not built")
@pytest.mark.parametrize("cfg", [8, 10, 16])
def test_oracle_sync_blocks_identical_to_reference(cfg):
    o, r = Oracle(cfg), RefLib(cfg)
    assert o.preamble().tobytes() == r.preamble().tobytes()
    for w in (0, 1):
        assert o.fir_taps(w).tobytes() == r.fir_taps(w).tobytes() and o.fir_taps(w).size == 33
    win, _ = _window(o, 9321)
    bits = o.payload_to_bits(o.gen_payload(SEED, 1))
    assert o.tx_passband(bits).tobytes() == r.tx_passband(bits).tobytes()
    for w in (0, 1):
        assert o.passband_to_baseband(win, which=w).tobytes() == r.passband_to_baseband(win, which=w).tobytes()
    assert o.passband_to_baseband(win[777:], which=1, decimation=4).tobytes() == r.passband_to_baseband(win[777:], which=1, decimation=4).tobytes()
    bbi = o.passband_to_baseband(win, which=0)
    assert o.time_sync_preamble(bbi, 100) == r.time_sync_preamble(bbi, 100)
    d, _ = o.time_sync_preamble(bbi, 100)
    ps = max(1, d // (o.Nofdm * 4))
    seg = bbi[(ps - 1) * o.Nofdm * 4: (ps - 1) * o.Nofdm * 4 + (o.preamble_nsymb + 4) * o.Nofdm * 4]
    for loc in (0, 1):
        assert o.time_sync_preamble(seg, 1, loc, 2) == r.time_sync_preamble(seg, 1, loc, 2)
    fine = (ps - 1) * o.Nofdm * 4 + o.time_sync_preamble(seg, 1, 0, 2)[0]
    bb = o.passband_to_baseband(win, which=1)[fine::4]
    assert o.freq_sync(bb[16:]) == r.freq_sync(bb[16:])


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 16])
def test_gpu_passband_to_baseband(cfg):
    from mercury_amd import RxPhy
    o = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=1)
    wins = np.stack([_window(o, 5000 + 1111 * i, seed=i)[0] for i in range(3)])
    carriers = np.array([CARRIER, CARRIER + 3.7, CARRIER - 11.0])
    for which in (0, 1):
        got = rx.passband_to_baseband(wins, carriers, which=which)
        for w in range(3):
            ref = o.passband_to_baseband(wins[w], carrier=carriers[w], which=which)
            assert _rel(got[w], ref) < 1e-11, (cfg, which, w)
    # fused extraction: only the kept samples are filtered (rational_resampler of telecom_system.cc:1105 folded in)
    starts = np.array([4001, 17, 60000], np.int32)
    count = (o.preamble_nsymb + o.Nsymb) * o.Nofdm
    got = rx.passband_to_baseband(wins, carriers, which=1, start=starts, count=count, decimation=4)
    for w in range(3):
        full = o.passband_to_baseband(wins[w], carrier=carriers[w], which=1)
        ref = full[starts[w]::4][:count]
        assert _rel(got[w, : ref.size], ref) < 1e-11
    # edges of the window: taps that fall outside the input are skipped exactly like cl_FIR::apply
    short = wins[0, :300]
    assert _rel(rx.passband_to_baseband(short, CARRIER, which=0)[0], o.passband_to_baseband(short, which=0)) < 1e-11
    # one carrier for all windows (every call receive_byte makes before a frequency offset is known): the mixer's cos / sin come from
    # the host's libm the way the reference evaluates them, and the baseband is bit-identical
    for which in (0, 1):
        got = rx.passband_to_baseband(wins, np.full(3, CARRIER + 1.25), which=which)
        for w in range(3):
            assert np.array_equal(got[w], o.passband_to_baseband(wins[w], carrier=CARRIER + 1.25, which=which)), (cfg, which, w)
    got = rx.passband_to_baseband(wins, np.full(3, CARRIER), which=1, start=starts, count=count, decimation=4)
    for w in range(3):
        ref = o.passband_to_baseband(wins[w], which=1)[starts[w]::4][:count]
        assert np.array_equal(got[w, : ref.size], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 13, 16])
def test_gpu_time_and_frequency_sync(cfg):
    from mercury_amd import RxPhy
    o = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=1)
    delays = [9321, 20500, 3000]
    wins = np.stack([_window(o, d, seed=10 + i, carrier=CARRIER + (0.0, 4.0, -6.0)[i])[0] for i, d in enumerate(delays)])
    bbi = np.stack([o.passband_to_baseband(w, which=0) for w in wins])     # identical inputs for both sides
    d_gpu, c_gpu = rx.time_sync_preamble(bbi, 100)
    for w in range(3):
        d_ref, c_ref = o.time_sync_preamble(bbi[w], 100)
        assert d_gpu[w] == d_ref and c_gpu[w] == c_ref
    sym = o.Nofdm * 4
    segs, bases = [], []
    for w in range(3):
        ps = max(1, int(d_gpu[w]) // sym)
        bases.append((ps - 1) * sym)
        segs.append(bbi[w, bases[-1]: bases[-1] + (o.preamble_nsymb + 4) * sym])
    segs = np.stack(segs)
    for loc in (0, 1, 2):
        d2, c2 = rx.time_sync_preamble(segs, 1, loc, 2)
        for w in range(3):
            d_ref, c_ref = o.time_sync_preamble(segs[w], 1, loc, 2)
            assert d2[w] == d_ref and c2[w] == c_ref          # same samples in, same sums in the same order
    d2, _ = rx.time_sync_preamble(segs, 1, 0, 2)
    frames = np.stack([o.passband_to_baseband(wins[w], which=1)[bases[w] + int(d2[w])::4][: (o.preamble_nsymb + o.Nsymb) * o.Nofdm]
                       for w in range(3)])
    f_gpu = rx.freq_sync(frames[:, 16:])
    for w in range(3):
        f_ref = o.freq_sync(frames[w, 16:])
        assert f_gpu[w] == f_ref, (f_gpu[w], f_ref)          # bit-identical: FFTs and sum on the device in the reference's order, atan on the host


@pytest.mark.gpu
def test_capture_window_to_payload_chain_on_gpu():
    """receive_byte's happy path (telecom_system.cc:676-1345) assembled from the GPU building blocks:
    passband window -> time-sync filter -> coarse + fine Schmidl-Cox -> data filter + decimate at the found
    delay -> Moose -> [re-mix if needed] -> hot path. The payload must come back, and every integer decision
    (delays, iterations, bytes) must equal what the CPU oracle gets on the same windows."""
    from mercury_amd import RxPhy
    cfg = 8
    o = Oracle(cfg, 50)
    W = 6
    rx = RxPhy(cfg, max_batch=W)
    delays = [5000, 7777, 12345, 30011, 41234, 60000]
    wins, payloads = zip(*[_window(o, d, noise=0.03, seed=50 + i, frame_idx=100 + i) for i, d in enumerate(delays)])
    wins = np.stack(wins)
    sym = o.Nofdm * 4
    bbi = rx.passband_to_baseband(wins, CARRIER, which=0)
    coarse, _ = rx.time_sync_preamble(bbi, 100)
    bases = [(max(1, int(c) // sym) - 1) * sym for c in coarse]
    segs = np.stack([bbi[w, bases[w]: bases[w] + (o.preamble_nsymb + 4) * sym] for w in range(W)])
    fine, _ = rx.time_sync_preamble(segs, 1, 0, 2)
    start = np.array([bases[w] + int(fine[w]) for w in range(W)], np.int32)
    nfr = (o.preamble_nsymb + o.Nsymb) * o.Nofdm
    bb = rx.passband_to_baseband(wins, CARRIER, which=1, start=start, count=nfr, decimation=4)
    foff = rx.freq_sync(bb[:, 16:])
    assert np.abs(foff).max() < 2.0                     # no carrier offset was applied
    out = rx.receive(bb[:, o.preamble_nsymb * o.Nofdm:])
    for w in range(W):
        assert delays[w] - 64 <= int(start[w]) <= delays[w]    # inside the guard interval (16 samples x 4)
        assert out["stats"]["message_decoded"][w] == 1
        assert np.array_equal(out["payload"][w][: o.payload_bytes], payloads[w].astype(np.uint8))
        # the same chain on the CPU oracle
        bbi_o = o.passband_to_baseband(wins[w], which=0)
        c_o, _ = o.time_sync_preamble(bbi_o, 100)
        base_o = (max(1, c_o // sym) - 1) * sym
        f_o, _ = o.time_sync_preamble(bbi_o[base_o: base_o + (o.preamble_nsymb + 4) * sym], 1, 0, 2)
        assert (c_o, base_o + f_o) == (int(coarse[w]), int(start[w]))
        bb_o = o.passband_to_baseband(wins[w], which=1)[base_o + f_o::4][:nfr]
        ref = o.rx(bb_o[o.preamble_nsymb * o.Nofdm:], oraclelib.FLAGS_RECEIVE_BYTE)
        assert out["stats"]["iterations_done"][w] == ref["iterations"]
        assert np.array_equal(out["payload"][w], ref["bytes"].astype(np.uint8))


# ---- MFSK synchroniser / ACK-BREAK pattern detector ------------------------------------------------------------
def _pattern_windows(orc, which, W, nsym, sigmas, rng, slot0=5):
    """W capture windows of interpolated baseband: noise + the known tone pattern (zero-order hold x4, so the
    reference's decimation recovers the symbol samples) starting at a different symbol slot in each window."""
    pat = np.repeat(orc.mfsk_pattern(which) / 16.0, 4)
    out, slots = [], []
    for w in range(W):
        sigma = sigmas[w % len(sigmas)]
        buf = sigma * (rng.standard_normal(nsym * 1088) + 1j * rng.standard_normal(nsym * 1088))
        slot = slot0 + 3 * w
        buf[slot * 1088: slot * 1088 + pat.size] += pat
        out.append(buf)
        slots.append(slot)
    return np.stack(out), slots


@pytest.mark.parametrize("cfg", [100, 101, 102])
def test_time_sync_mfsk_oracle_vs_ref_and_finds_the_preamble(cfg):
    orc = oraclelib.Oracle(cfg)
    rng = np.random.default_rng(cfg)
    bufs, slots = _pattern_windows(orc, 0, 4, 40, (0.05, 0.5, 1.0, 2.0), rng)
    for w in range(4):
        d = orc.time_sync_mfsk(bufs[w])
        assert d == slots[w] * 1088                                   # low enough noise: the preamble slot is found
        assert orc.time_sync_mfsk(bufs[w], slots[w] + 1) != d         # anti-re-decode start skips it (telecom_system.cc:683-686)
        if oraclelib.RefLib.available():
            ref = oraclelib.RefLib(cfg)
            assert ref.time_sync_mfsk(bufs[w]) == d
            assert ref.time_sync_mfsk(bufs[w], slots[w] + 1) == orc.time_sync_mfsk(bufs[w], slots[w] + 1)


@pytest.mark.parametrize("cfg", [8, 100])
def test_detect_ack_pattern_oracle_vs_ref(cfg):
    orc = oraclelib.Oracle(cfg)
    rng = np.random.default_rng(50 + cfg)
    for which in (1, 2):
        assert orc.mfsk_pattern(which).size == 16 * 272
        bufs, _ = _pattern_windows(orc, which, 3, 36, (0.05, 1.0, 3.0), rng)
        for w in range(3):
            m, n = orc.detect_ack_pattern(bufs[w], which)
            other, _ = orc.detect_ack_pattern(bufs[w], 3 - which)
            assert m > other                                          # ACK and BREAK tone sets do not alias
            if w == 0:
                assert n == 16 and m > 15.5
            if oraclelib.RefLib.available():
                ref = oraclelib.RefLib(cfg)
                assert ref.mfsk_pattern(which).tobytes() == orc.mfsk_pattern(which).tobytes()
                assert ref.detect_ack_pattern(bufs[w], which) == (m, n)
    assert orc.detect_ack_pattern(np.zeros(15 * 1088, np.complex128)) == (0.0, 0)   # shorter than the pattern, ofdm.cc:2075


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [100, 101, 102])
def test_gpu_time_sync_mfsk_matches_oracle(cfg):
    from mercury_amd import RxPhy
    orc = oraclelib.Oracle(cfg)
    rng = np.random.default_rng(7 * cfg)
    bufs, slots = _pattern_windows(orc, 0, 6, 44, (0.05, 0.5, 1.0, 2.0, 4.0, 8.0), rng)
    rx = RxPhy(cfg, max_batch=1)
    for start in (0, 9):
        got = rx.time_sync_mfsk(bufs, start)
        want = [orc.time_sync_mfsk(b, start) for b in bufs]
        assert list(got) == want, (cfg, start)
    assert list(rx.time_sync_mfsk(bufs)[:3]) == [s * 1088 for s in slots[:3]]
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 100])
def test_gpu_detect_ack_pattern_matches_oracle(cfg):
    from mercury_amd import RxPhy
    orc = oraclelib.Oracle(cfg)
    rng = np.random.default_rng(11 * cfg + 1)
    rx = RxPhy(cfg, max_batch=1)
    for which in (1, 2):
        bufs, _ = _pattern_windows(orc, which, 5, 40, (0.05, 1.0, 2.0, 3.0, 5.0), rng)
        for pattern in (1, 2):
            metric, matched = rx.detect_ack_pattern(bufs, pattern)
            for w in range(5):
                m, n = orc.detect_ack_pattern(bufs[w], pattern)
                assert metric[w] == m and matched[w] == n, (cfg, which, pattern, w)      # bit-exact metric
    m, n = rx.detect_ack_pattern(np.zeros((2, 10 * 1088), np.complex128))
    assert list(m) == [0.0, 0.0] and list(n) == [0, 0]
    rx.close()


@pytest.mark.gpu
def test_gpu_detect_ack_pattern_from_passband_matches_oracle_chain():
    """telecom_system.cc:1628-1655: passband -> FIR_rx_data baseband -> detector, fused on the device; against the oracle's
    two calls. The mixer's device cos/sin differ from glibc in the last ulp, so the metric is compared to 1e-9."""
    from mercury_amd import RxPhy
    cfg = 8
    orc = oraclelib.Oracle(cfg)
    rng = np.random.default_rng(77)
    n = 40 * 1088
    wins = []
    for which, noise in ((1, 0.01), (2, 0.05), (1, 0.3)):
        pat = np.repeat(orc.mfsk_pattern(which) / 16.0 * np.sqrt(0.1), 4)            # interpolated baseband at TX power
        t = np.arange(pat.size)
        pb = (pat.real * np.cos(2 * np.pi * CARRIER * t / 48000.0) + pat.imag * np.sin(2 * np.pi * CARRIER * t / 48000.0)) * np.sqrt(2.0)
        x = rng.standard_normal(n) * noise
        x[9 * 1088: 9 * 1088 + pb.size] += pb
        wins.append(x)
    wins = np.stack(wins)
    rx = RxPhy(cfg, max_batch=1)
    for pattern in (1, 2):
        metric, matched = rx.detect_ack_pattern_from_passband(wins, CARRIER, pattern)
        for w in range(3):
            bbi = orc.passband_to_baseband(wins[w], CARRIER, 1, 1)
            m, k = orc.detect_ack_pattern(bbi, pattern)
            assert abs(metric[w] - m) <= 1e-9 * max(1.0, m) and matched[w] == k, (pattern, w, metric[w], m)
    m1, k1 = rx.detect_ack_pattern_from_passband(wins[:1], CARRIER, 1)
    assert k1[0] == 16 and m1[0] > 12.0                                               # the ACK pattern is found in its window
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [100, 101, 102])
def test_gpu_mfsk_sync_kernel_equals_host_statement_on_adversarial_energies(cfg):
    """The device search of time_sync_mfsk (mgpu_mfsk_sync_kernel) against the host statement that the oracle comparisons pin, on slot
    energies built to hit the corner cases of the arg-max: exact ties between start slots (the first must win), all-zero slots (add
    nothing), slots whose total is zero / negative zero / NaN / infinite, a search start past every candidate, buffers whose last symbols do
    not fit (the sum stops early), random energies."""
    from mercury_amd import RxPhy
    rx = RxPhy(cfg, max_batch=1)
    rng = np.random.default_rng(SEED + 9 + cfg)
    Nc, sym = 50, rx.Nofdm * 4
    for nslots in (rx.preamble_nsymb, rx.preamble_nsymb + 1, 37, 300, 648):
        W = 24
        e = rng.random((W, nslots, Nc)) * np.exp(rng.uniform(-20, 5, (W, nslots, 1)))
        e[1] = 0.0
        e[2] = 1.0                                                     # every start slot ties
        e[3, ::2] = 0.0
        e[4, nslots // 2] = np.nan
        e[5, nslots // 3] = np.inf
        e[6] = np.tile(e[6, :1], (nslots, 1))                          # periodic: many exact ties
        e[7, :, :] = -0.0
        e[8] = np.round(e[8] * 4) / 4                                  # coarse values: ties by rounding
        for size in (nslots * sym, nslots * sym - 1, nslots * sym - sym // 2, (nslots - 1) * sym + 5):
            for ss in (None, np.zeros(W, np.int32), rng.integers(-3, nslots + 3, W).astype(np.int32)):
                a = rx.debug_mfsk_sync(e, size, ss, 0)
                b = rx.debug_mfsk_sync(e, size, ss, 1)
                assert np.array_equal(a, b), (cfg, nslots, size, None if ss is None else ss.tolist(), a.tolist(), b.tolist())
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 100])
def test_gpu_p2b_sliding_tap_kernels_equal_generic_kernel_bit_for_bit(cfg):
    """passband_to_baseband with the taps sliding through registers (4 adjacent outputs per lane, sync.hip) against the generic kernel the
    oracle comparisons pin: every output bit-identical — whole windows at decimation 1 (both filters), frames cut at a start offset with
    decimation 4 (counts that fill no block, starts near both ends of the window so that taps fall outside the input), per-window
    carriers (device sincos) and a shared one (host table), a window shorter than the filter."""
    from mercury_amd import RxPhy
    rx = RxPhy(cfg, max_batch=1)
    rng = np.random.default_rng(SEED + 5 + cfg)
    n = 20000
    wins = rng.standard_normal((5, n)) * np.exp(rng.uniform(-4, 2, (5, 1)))
    cases = []
    for carriers in (np.full(5, CARRIER), CARRIER + rng.uniform(-20, 20, 5)):
        for which in (0, 1):
            cases.append(dict(passband=wins, carrier_hz=carriers, which=which))
        for count in (1, 255, 256, 257, 1500):
            cases.append(dict(passband=wins, carrier_hz=carriers, which=1, start=np.array([0, 3, n - 4 * count - 7, max(0, n - 4 * count + 40), 1234], np.int32),
                              count=count, decimation=4))
    cases.append(dict(passband=wins[:, :20], carrier_hz=CARRIER, which=0))
    cases.append(dict(passband=wins[:, :1025], carrier_hz=CARRIER, which=1))
    try:
        for kw in cases:
            rx.debug_p2b_variant(0)
            a = rx.passband_to_baseband(**kw)
            rx.debug_p2b_variant(-1)
            b = rx.passband_to_baseband(**kw)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)), {k: v for k, v in kw.items() if k != "passband"}
    finally:
        rx.debug_p2b_variant(-1)
    rx.close()


@pytest.mark.gpu
def test_gpu_span_energies_lane_per_span_equals_wave_per_span_and_the_sequential_sum():
    """The span energies of receive_byte's gates / recoveries (telecom_system.cc:758-766, :826-834, :1044-1066): both kernels (one
    wavefront per span; one lane per span, used from 4096 spans per launch) against the sequential sum in sample order — spans anywhere
    in a window, clipped at its end, starting at or past it (no terms), short and full lengths, a span count that fills no wavefront."""
    from mercury_amd import RxPhy
    rx = RxPhy(8, max_batch=1)
    rng = np.random.default_rng(SEED + 77)
    W, size = 7, 5000
    z = (rng.standard_normal((W, size)) + 1j * rng.standard_normal((W, size))) * np.exp(rng.uniform(-8, 3, (W, 1)))
    for n, length in ((1, 1088), (63, 1088), (64, 17), (700, 1088), (333, 640), (130, 1)):
        wv = rng.integers(0, W, n).astype(np.int32)
        off = rng.integers(0, size - 1088, n).astype(np.int32)
        off[::5] = size - rng.integers(0, 1200, off[::5].size)          # clipped at the window end
        off[1::11] = size + rng.integers(0, 5, off[1::11].size)          # at / past the end: no terms
        ref_s = np.zeros(n)
        ref_c = np.zeros(n, np.int32)
        for j in range(n):
            m = int(np.clip(size - off[j], 0, length))
            x = z[wv[j], off[j]: off[j] + m]
            terms = x.real * x.real + x.imag * x.imag
            ref_s[j] = np.cumsum(terms)[-1] if m else 0.0               # cumsum adds in order
            ref_c[j] = m
        for variant in (0, 1):
            s_, c_ = rx.debug_span_energy(z, wv, off, length, variant)
            assert np.array_equal(c_, ref_c), (n, length, variant)
            assert np.array_equal(s_.view(np.uint64), ref_s.view(np.uint64)), (n, length, variant)
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 16])
def test_gpu_fine_search_shared_products_equal_dense_kernel_on_every_candidate(cfg):
    """The many-window fine search (step 1: a lane owns 4 or 8 adjacent candidates and shares each sample's products between them, sync.hip)
    against the dense kernel that the oracle comparisons pin: every candidate's metric bit-identical — the search window receive_byte
    uses ((preamble + 4) symbols), windows whose candidate count is not a multiple of the lanes' share, one candidate, silent stretches,
    sub-windows with their own start and length."""
    from mercury_amd import RxPhy
    rx = RxPhy(cfg, max_batch=1)
    rng = np.random.default_rng(SEED + 50 + cfg)
    sym = rx.Nofdm * 4
    L = rx.preamble_nsymb * sym
    for W, size in [(3, L + 4 * sym), (40, L + 4 * sym), (2, L + 1), (2, L + 1023), (5, L + 1024), (5, L + 1025), (2, L + 2055), (300, L + 700)]:
        z = (rng.standard_normal((W, size)) + 1j * rng.standard_normal((W, size))) * np.exp(rng.uniform(-6, 2, (W, 1)))
        z[0, : size // 3] = 0.0                                        # the "no signal" branch of the metric (sums below 0.001)
        a = rx.debug_tsync_metric(z, 1, variant=0)
        for v in (1, 2, -1):
            b = rx.debug_tsync_metric(z, 1, variant=v)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)), (cfg, W, size, v)
    W, size = 40, L + 4 * sym + 300
    z = rng.standard_normal((W, size)) + 1j * rng.standard_normal((W, size))
    start = rng.integers(0, 300, W).astype(np.int32)
    sub = np.array([int(rng.integers(0, size - start[w] + 1)) for w in range(W)], np.int32)
    sub[::7] = L                                                    # no candidate at all
    sub[1::7] = L + 1                                               # exactly one
    a = rx.debug_tsync_metric(z, 1, variant=0, start=start, sub_size=sub)
    for v in (1, 2):
        b = rx.debug_tsync_metric(z, 1, variant=v, start=start, sub_size=sub)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), v
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 16])
def test_gpu_streaming_coarse_metric_equals_staged_kernel_on_every_candidate(cfg):
    """The many-window coarse search (one wavefront streams a window through an LDS ring, sync.hip) against the staged kernel that the
    oracle comparisons above pin: every candidate's metric bit-identical, for whole capture windows, short windows (fewer candidates than
    lanes in flight, one candidate), several windows per launch (pieces of a window per wavefront) and many windows (one piece)."""
    from mercury_amd import RxPhy
    rx = RxPhy(cfg, max_batch=1)
    rng = np.random.default_rng(SEED + cfg)
    L = rx.preamble_nsymb * rx.Nofdm * 4
    full = rx.receive_buffer_samples()
    for W, size in [(3, full), (40, full), (2, L + 1), (2, L + 100 * 7 + 3), (5, L + 100 * 45), (5, L + 100 * 46 + 50), (1100, L + 100 * 60 + 1)]:
        z = (rng.standard_normal((W, size)) + 1j * rng.standard_normal((W, size))) * np.exp(rng.uniform(-6, 2, (W, 1)))
        z[0, : size // 3] = 0.0                                        # the "no signal" branch of the metric (sums below 0.001)
        a = rx.debug_tsync_metric(z, 100, variant=0)
        b = rx.debug_tsync_metric(z, 100, variant=1)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64)), (cfg, W, size)
        assert np.array_equal(rx.debug_tsync_metric(z, 100).view(np.uint64), a.view(np.uint64))
    # sub-windows with their own start and length, as receive_byte's recoveries search them (some too short for a single candidate)
    W, size = 70, full
    z = rng.standard_normal((W, size)) + 1j * rng.standard_normal((W, size))
    start = rng.integers(0, size // 2, W).astype(np.int32)
    sub = np.array([int(rng.integers(0, size - start[w] + 1)) for w in range(W)], np.int32)
    sub[::7] = L                                                    # no candidate at all
    sub[1::7] = np.minimum(L + 1, size - start[1::7])               # exactly one
    a = rx.debug_tsync_metric(z, 100, variant=0, start=start, sub_size=sub)
    b = rx.debug_tsync_metric(z, 100, variant=1, start=start, sub_size=sub)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    rx.close()


@pytest.mark.gpu
def test_gpu_peak_selection_kernel_matches_the_reference_selection():
    """The wavefront-per-window peak selection (sync.hip) against a literal restatement of ofdm.cc:1943-1964 (overwrite, not swap) on rows
    built to hit its corners: negative maxima (an implicit zero between / behind the candidates wins), exact zeros and ties (the first
    occurrence wins), later passes, steps 1 / 4 / 100, buffers that end right behind the last candidate or well after it, NaN start entries."""
    import ctypes as C
    from mercury_amd import RxPhy
    from test_host_logic import _reference_selection
    rx = RxPhy(8, max_batch=1)
    rng = np.random.default_rng(SEED)
    pool = np.array([-1.0, -0.25, -0.25, 0.0, 0.0, 0.125, 0.5, 0.5, 0.75, 1.0])
    for step, ncmax in ((1, 200), (4, 70), (100, 9), (1, 1), (100, 130)):
        n = 96
        ncand = rng.integers(1, ncmax + 1, n).astype(np.int32)
        vals = np.zeros((n, ncmax))
        size = np.zeros(n, np.int32)
        loc = rng.integers(0, 4, n).astype(np.int32)
        ntrials = 3
        for w in range(n):
            kind = w % 4
            row = rng.choice(pool, ncand[w]) if kind < 2 else rng.standard_normal(ncand[w])
            if kind == 1:
                row = -np.abs(row) - (w % 3 == 0)                       # all non-positive
            if kind == 3 and w % 8 == 3:
                row[0] = np.nan
            vals[w, : ncand[w]] = row
            size[w] = (ncand[w] - 1) * step + 1 + (0 if w % 5 == 0 else int(rng.integers(0, 3 * step + 2)))
        size = np.maximum(size, ntrials).astype(np.int32)
        delay = np.zeros(n, np.int32)
        corr = np.zeros(n, np.float64)
        rc = rx.lib.mgpu_debug_select_peak(rx.h, vals.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(ncmax), ncand.ctypes.data_as(C.c_void_p),
                                           size.ctypes.data_as(C.c_void_p), loc.ctypes.data_as(C.c_void_p), C.c_int(step), C.c_int(ntrials),
                                           delay.ctypes.data_as(C.c_void_p), corr.ctypes.data_as(C.c_void_p))
        assert rc == 0
        for w in range(n):
            d_ref, c_ref = _reference_selection(vals[w, : ncand[w]].copy(), step, int(size[w]), int(loc[w]), ntrials)
            assert delay[w] == d_ref and (corr[w] == c_ref or (np.isnan(corr[w]) and np.isnan(c_ref))), (step, w, ncand[w], size[w], loc[w], delay[w], d_ref, corr[w], c_ref)
    rx.close()
