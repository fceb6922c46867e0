"""cl_telecom_system::transmit_byte (telecom_system.cc:342-556; SURVEY.md §8 row f4, the TX mirror up to the audio samples).

Pinning chain: the compiled reference objects composed as transmit_byte / transmit_bit compose them
(oracle/ref_harness.cc:mref_transmit_byte) -> tests/golden/golden_tx.json -> the C restatement (morc_transmit_byte) ->
the GPU path (include/mercury_tx.h, csrc/tx.hip), every link bit for bit."""
import importlib.util
import json
import os

import numpy as np
import pytest

import oraclelib
from oraclelib import CARRIER, NO_FILTER_MESSAGE, SINGLE_MESSAGE, Oracle

HERE = os.path.dirname(os.path.abspath(__file__))
needs_ref = pytest.mark.skipif(not oraclelib.RefLib.available(), reason="oracle/_ref not built (no /root/reference here)")


def _make_golden():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


MG = _make_golden()


@pytest.mark.parametrize("cfg", MG.TX_CFGS)
def test_oracle_transmit_byte_matches_the_reference_fixture(cfg):
    want = json.load(open(os.path.join(HERE, "golden", "golden_tx.json")))[str(cfg)]
    got = json.loads(json.dumps(MG.tx_case(Oracle(cfg), cfg)))      # same code path, the oracle as `lib`
    assert got == want


@needs_ref
@pytest.mark.parametrize("cfg", [1, 3, 7, 9, 12, 13, 15])
def test_oracle_transmit_byte_matches_reference_build_on_the_other_modes(cfg):
    orc, ref = Oracle(cfg), oraclelib.RefLib(cfg)
    rng = np.random.default_rng(cfg)
    for loc in (SINGLE_MESSAGE, NO_FILTER_MESSAGE):
        msg = rng.integers(0, 256, int(rng.integers(1, orc.payload_bytes + 1))).astype(np.int32)
        start = int(rng.integers(0, 2 ** 40))
        assert np.array_equal(orc.transmit_byte(msg, message_location=loc, start_sample=start),
                              ref.transmit_byte(msg, message_location=loc, start_sample=start))


def test_oracle_transmit_filters_have_the_reference_shape():
    """97 taps at 48 kHz for a 1 kHz transition band. FIR_tx1 (high-pass at the lower band edge, Hamming) halves the amplitude at
    its cut-off and passes the carrier; FIR_tx2 (low-pass at the upper band edge, the reference's periodic Blackman window) halves
    it at the upper edge."""
    import ctypes as C
    f = Oracle(8).lib.morc_tx_fir_taps
    f.restype = C.c_int
    lo, hi = CARRIER - oraclelib.BANDWIDTH / 2, CARRIER + oraclelib.BANDWIDTH / 2

    def gain(t, hz):
        return abs(np.sum(t * np.exp(-2j * np.pi * hz / 48000.0 * np.arange(t.size))))

    taps = np.zeros(128)
    assert f(C.c_double(CARRIER), C.c_int(0), taps.ctypes.data_as(C.c_void_p)) == 97
    t1 = taps[:97].copy()
    assert np.allclose(t1, t1[::-1], rtol=0, atol=1e-16)                      # linear phase (the window is symmetric to rounding)
    assert 0.45 < gain(t1, lo) < 0.6 and 0.95 < gain(t1, CARRIER) < 1.05 and gain(t1, 0.0) < gain(t1, lo)
    assert f(C.c_double(CARRIER), C.c_int(1), taps.ctypes.data_as(C.c_void_p)) == 97
    t2 = taps[:97].copy()
    assert 0.45 < gain(t2, hi) < 0.55 and 0.95 < gain(t2, CARRIER) < 1.05 and gain(t2, hi + 1500) < 0.01


def test_oracle_transmit_then_receive_round_trip():
    """The transmitted audio (clipped, filtered) placed in a capture window is what receive_byte decodes. The receiver's audio
    gain is 2: at unit gain the filtered 0.1 W frame sits just under receive_byte's 0.001 energy gate (telecom_system.cc:808-830)."""
    for cfg in (8, 16, 101):
        orc = Oracle(cfg)
        msg = np.random.default_rng(cfg).integers(0, 256, orc.payload_bytes).astype(np.int32)
        pb = orc.transmit_byte(msg)
        n = orc.buffer_samples()
        x = np.random.default_rng(1).standard_normal(n) * 1e-3
        d = 9 * orc.Nofdm * 4 + 123
        x[d: d + pb.size] += 2.0 * pb
        r = orc.receive_byte(x)
        assert r["message_decoded"] == 1 and np.array_equal(r["payload"], msg), cfg


# ---- GPU ---------------------------------------------------------------------------------------------------------------------
ALL_CFGS = list(range(17)) + [100, 101, 102]


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_gpu_transmit_byte_matches_oracle(cfg):
    from mercury_amd import RxPhy
    orc = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=8)
    assert rx.transmit_frame_samples() == (orc.preamble_nsymb + orc.Nsymb) * orc.Nofdm * 4
    rng = np.random.default_rng(900 + cfg)
    F = 4
    pls = rng.integers(0, 256, (F, orc.payload_bytes)).astype(np.uint8)
    nbytes = np.array([orc.payload_bytes, 1, orc.payload_bytes // 2, 0], np.int32)
    for loc, start, nb in ((SINGLE_MESSAGE, 0, None), (NO_FILTER_MESSAGE, 0, None), (SINGLE_MESSAGE, 2 ** 33 + 5, nbytes)):
        got = rx.transmit_byte(pls, CARRIER, nbytes=nb, message_location=loc, start_sample=start)
        for f in range(F):
            msg = pls[f, : (orc.payload_bytes if nb is None else nb[f])].astype(np.int32)
            want = orc.transmit_byte(msg, message_location=loc, start_sample=start)
            assert np.array_equal(got[f], want), (cfg, loc, start, f)
    # non-default power, clipping levels and carrier
    kw = dict(message_location=NO_FILTER_MESSAGE, output_power_watt=0.25, preamble_papr_cut=5.0, data_papr_cut=6.0)
    got = rx.transmit_byte(pls[:1], 1650.0, **kw)
    assert np.array_equal(got[0], orc.transmit_byte(pls[0].astype(np.int32), carrier=1650.0, **kw))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 8, 13, 16, 100])
def test_gpu_transmit_bit_is_what_transmit_byte_calls(cfg):
    """cl_telecom_system::transmit_bit (telecom_system.cc:384-556; mgpu_transmit_bit_batch): transmit_byte packs the message bytes LSB first,
    appends the CRC16 and calls it (telecom_system.cc:343-383) - the same data bits handed to transmit_bit directly give the same audio."""
    from mercury_amd import RxPhy
    orc = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=4)
    rng = np.random.default_rng(1900 + cfg)
    pls = rng.integers(0, 256, (3, orc.payload_bytes)).astype(np.uint8)
    bits = np.stack([orc.payload_to_bits(pls[f].astype(np.int32)) for f in range(3)]).astype(np.uint8)
    for loc in (SINGLE_MESSAGE, NO_FILTER_MESSAGE):
        by_byte = rx.transmit_byte(pls, CARRIER, message_location=loc)
        by_bit = rx.transmit_bit(bits, CARRIER, message_location=loc)
        assert np.array_equal(by_bit, by_byte), (cfg, loc)
        assert np.array_equal(by_bit[1], orc.transmit_byte(pls[1].astype(np.int32), message_location=loc)), (cfg, loc)
    rx.close()


@pytest.mark.parametrize("cfg", [c for c in MG.TX_CFGS if c < 100])
def test_library_pre_equalization_channel_matches_the_reference_fixture(cfg):
    """cl_telecom_system::get_pre_equalization_channel (telecom_system.cc:3108-3145): the library computes the table on the host
    (mgpu_host_pre_equalization_channel, no GPU needed); it must be the reference's, bit for bit, for both carriers of the fixture
    (tests/golden/golden_tx.json, generated from the reference's objects), and the oracle's."""
    import ctypes as C
    from mercury_amd import load_library
    lib = load_library()
    want = json.load(open(os.path.join(HERE, "golden", "golden_tx.json")))[str(cfg)]
    orc = Oracle(cfg)
    for tag, fc in (("pre_eq", CARRIER), ("pre_eq_1650", 1650.0)):
        h = np.zeros(50, np.complex128)
        assert lib.mgpu_host_pre_equalization_channel(C.c_int(cfg), C.c_double(fc), h.ctypes.data_as(C.c_void_p)) == 0
        assert MG.digest(h) == want[tag][0], (cfg, tag)
        assert np.array_equal(h, orc.get_pre_equalization_channel(fc))
        assert 1.0 < np.abs(h).min() and np.abs(h).max() < 6.0           # far from "all ones": the band edges need 4.8x
    assert lib.mgpu_host_pre_equalization_channel(C.c_int(100), C.c_double(CARRIER), h.ctypes.data_as(C.c_void_p)) == 1      # MFSK: none
    assert lib.mgpu_host_pre_equalization_channel(C.c_int(8), C.c_double(CARRIER), None) == 1


@needs_ref
@pytest.mark.parametrize("cfg", [1, 5, 9, 12, 15])
def test_oracle_pre_equalization_matches_reference_build_on_the_other_modes(cfg):
    orc, ref = Oracle(cfg), oraclelib.RefLib(cfg)
    for fc in (CARRIER, 1234.5):
        assert np.array_equal(orc.get_pre_equalization_channel(fc), ref.get_pre_equalization_channel(fc))
    msg = np.random.default_rng(cfg).integers(0, 256, orc.payload_bytes).astype(np.int32)
    for loc in (SINGLE_MESSAGE, NO_FILTER_MESSAGE):
        assert np.array_equal(orc.transmit_byte(msg, message_location=loc, start_sample=5, pre_equalize=True),
                              ref.transmit_byte(msg, message_location=loc, start_sample=5, pre_equalize=True))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 8, 11, 13, 16])
def test_gpu_transmit_byte_with_pre_equalization_matches_oracle(cfg):
    """transmit_bit's pre-equalisation (telecom_system.cc:474-494): with the table installed the audio equals the oracle's (pinned
    to the reference's objects by the fixture above), for the filtered / unfiltered / stream forms and another carrier; removing
    the table gives the plain audio again."""
    from mercury_amd import RxPhy
    orc = Oracle(cfg)
    F = 3
    rx = RxPhy(cfg, max_batch=F)
    rng = np.random.default_rng(900 + cfg)
    pls = rng.integers(0, 256, (F, rx.payload_stride)).astype(np.uint8)
    plain = rx.transmit_byte(pls, CARRIER, message_location=SINGLE_MESSAGE)
    for fc in (CARRIER, 1650.0):
        h = rx.pre_equalization_channel(fc)
        assert np.array_equal(h, orc.get_pre_equalization_channel(fc))
        rx.set_pre_equalization_channel(h)
        for loc, start in ((SINGLE_MESSAGE, 0), (NO_FILTER_MESSAGE, 2 ** 33 + 5)):
            got = rx.transmit_byte(pls, fc, message_location=loc, start_sample=start)
            for f in range(F):
                want = orc.transmit_byte(pls[f, : orc.payload_bytes].astype(np.int32), carrier=fc, message_location=loc, start_sample=start,
                                         pre_equalize=True)
                assert np.array_equal(got[f], want), (cfg, fc, loc, f)
    assert not np.array_equal(got[0], plain[0])
    rx.set_pre_equalization_channel(None)
    assert np.array_equal(rx.transmit_byte(pls, CARRIER, message_location=SINGLE_MESSAGE), plain)
    rx.close()


@needs_ref
@pytest.mark.parametrize("cfg", [3, 8, 13, 102])
def test_oracle_overlap_save_message_locations_match_reference_objects(cfg):
    """FIRST / MIDDLE / FLUSH_MESSAGE (telecom_system.cc:559-590) restated literally on the reference's own FIR objects and
    shift_left (oracle/ref_harness.cc:mref_transmit_stream) against the C restatement: audio of every call and the 3-frame
    buffer they leave, over a FIRST stream, a MIDDLE continuation on that buffer and a FLUSH on a buffer of noise."""
    orc, ref = Oracle(cfg), oraclelib.RefLib(cfg)
    rng = np.random.default_rng(40 + cfg)
    pls = rng.integers(0, 256, (5, orc.payload_bytes)).astype(np.int32)
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    a, ba = orc.transmit_stream(pls[:3], oraclelib.FIRST_MESSAGE, start_sample=99)
    b, bb = ref.transmit_stream(pls[:3], oraclelib.FIRST_MESSAGE, start_sample=99)
    assert np.array_equal(a, b) and np.array_equal(ba, bb)
    a, ba = orc.transmit_stream(pls[3:4], oraclelib.MIDDLE_MESSAGE, buffer=ba, start_sample=99 + 3 * used)
    b, bb = ref.transmit_stream(pls[3:4], oraclelib.MIDDLE_MESSAGE, buffer=bb, start_sample=99 + 3 * used)
    assert np.array_equal(a, b) and np.array_equal(ba, bb)
    noise = rng.standard_normal(ba.size) * 0.01
    a, ba = orc.transmit_stream(pls[4:], oraclelib.FLUSH_MESSAGE, buffer=noise.copy(), nbytes=np.array([2], np.int32))
    b, bb = ref.transmit_stream(pls[4:], oraclelib.FLUSH_MESSAGE, buffer=noise.copy(), nbytes=np.array([2], np.int32))
    assert np.array_equal(a, b) and np.array_equal(ba, bb)


def test_oracle_overlap_save_returns_the_previous_frame_filtered_with_its_neighbours():
    """What the stream form means: call n returns frame n-1 (FIRST_MESSAGE: the frame itself) as it comes out of the two
    filters run over the whole unfiltered stream; and away from the frame edges that equals SINGLE_MESSAGE filtering."""
    orc = Oracle(8)
    rng = np.random.default_rng(3)
    pls = rng.integers(0, 256, (4, orc.payload_bytes)).astype(np.int32)
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    y, buf = orc.transmit_stream(pls, oraclelib.FIRST_MESSAGE, start_sample=0)
    total = y.shape[1]
    single = [orc.transmit_byte(pls[i], message_location=SINGLE_MESSAGE, start_sample=i * used) for i in range(4)]
    raw = [orc.transmit_byte(pls[i], message_location=NO_FILTER_MESSAGE, start_sample=i * used) for i in range(4)]
    for n, src in ((0, 0), (1, 0), (2, 1), (3, 2)):            # FIRST returns its own frame, every later call the previous one
        assert np.array_equal(y[n][200: total - 200], single[src][200: total - 200]), n
    assert not np.array_equal(y[2][:100], single[1][:100])      # the edges see the neighbouring frames instead of silence
    assert np.array_equal(buf, np.concatenate([raw[2], raw[3], raw[3]]))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 8, 11, 16, 100, 102])
def test_gpu_overlap_save_message_locations_match_oracle(cfg):
    """mgpu_transmit_byte_batch with MGPU_FIRST / MIDDLE / FLUSH_MESSAGE: the batch stands for F consecutive calls; audio and the
    buffer carried between calls must equal the oracle's (pinned to the reference's objects above and by golden_tx.json) bit for
    bit, whether the calls come one by one or batched, from a fresh buffer or from one installed with mgpu_transmit_buffer."""
    from mercury_amd import RxPhy
    from mercury_amd.physical_layer import FIRST_MESSAGE, FLUSH_MESSAGE, MIDDLE_MESSAGE
    orc = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=8)
    rng = np.random.default_rng(500 + cfg)
    pls = rng.integers(0, 256, (6, orc.payload_bytes)).astype(np.uint8)
    ipl = pls.astype(np.int32)
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    assert not rx.transmit_buffer().any()                                              # a new context starts from zeros
    want, wbuf = orc.transmit_stream(ipl[:4], oraclelib.FIRST_MESSAGE, start_sample=7)
    got = rx.transmit_byte(pls[:4], CARRIER, message_location=FIRST_MESSAGE, start_sample=7, phase_continuous=1)
    assert np.array_equal(got, want) and np.array_equal(rx.transmit_buffer(), wbuf)
    # one call at a time, the way the reference is driven
    for i in (4, 5):
        w1, wbuf = orc.transmit_stream(ipl[i: i + 1], oraclelib.MIDDLE_MESSAGE if i == 4 else oraclelib.FLUSH_MESSAGE, buffer=wbuf,
                                       start_sample=7 + i * used)
        g1 = rx.transmit_byte(pls[i: i + 1], CARRIER, message_location=MIDDLE_MESSAGE if i == 4 else FLUSH_MESSAGE, start_sample=7 + i * used)
        assert np.array_equal(g1, w1) and np.array_equal(rx.transmit_buffer(), wbuf), i
    # an installed buffer, short messages
    noise = rng.standard_normal(wbuf.size) * 0.01
    rx.transmit_buffer(noise)
    nb = np.array([1, orc.payload_bytes, 0], np.int32)
    w2, wbuf = orc.transmit_stream(ipl[:3], oraclelib.MIDDLE_MESSAGE, buffer=noise.copy(), nbytes=nb, start_sample=0)
    g2 = rx.transmit_byte(pls[:3], CARRIER, nbytes=nb, message_location=MIDDLE_MESSAGE, start_sample=0, phase_continuous=1)
    assert np.array_equal(g2, w2) and np.array_equal(rx.transmit_buffer(), wbuf)
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 100])
def test_gpu_transmit_byte_phase_continuous_equals_consecutive_calls(cfg):
    """phase_continuous: message f starts where message f-1 stopped, as cl_ofdm::passband_start_sample runs on from call to call."""
    from mercury_amd import RxPhy
    orc = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=8)
    pls = np.random.default_rng(cfg).integers(0, 256, (3, orc.payload_bytes)).astype(np.uint8)
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    got = rx.transmit_byte(pls, CARRIER, start_sample=1000, phase_continuous=1)
    for f in range(3):
        assert np.array_equal(got[f], orc.transmit_byte(pls[f].astype(np.int32), start_sample=1000 + f * used)), f


@needs_ref
@pytest.mark.parametrize("cfg", [8, 16, 101])
def test_oracle_send_batch_signal_path_matches_reference_objects(cfg):
    """cl_arq_controller::send_batch's DSP (arq_common.cc:2224-2248): unfiltered frames with the carrier running on, edge frames
    repeated as padding, both transmit filters over the concatenation."""
    orc, ref = Oracle(cfg), oraclelib.RefLib(cfg)
    rng = np.random.default_rng(cfg)
    pl = rng.integers(0, 256, (3, orc.payload_bytes)).astype(np.int32)
    nb = np.array([orc.payload_bytes, 5, orc.payload_bytes // 2], np.int32)
    assert np.array_equal(orc.transmit_batch(pl, nb, start_sample=777), ref.transmit_batch(pl, nb, start_sample=777))


def test_oracle_send_batch_middle_frame_equals_filtering_with_its_neighbours():
    """What the padding buys: away from the batch edges a frame's filtered samples do not depend on how the batch was cut."""
    orc = Oracle(8)
    pl = np.random.default_rng(5).integers(0, 256, (4, orc.payload_bytes)).astype(np.int32)
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    whole = orc.transmit_batch(pl)
    tail = orc.transmit_batch(pl[1:], start_sample=used)          # the same frames 1..3 with the same carrier phase
    assert np.array_equal(whole[2:], tail[1:]) and np.array_equal(whole[1][200:], tail[0][200:]) and not np.array_equal(whole[1][:48], tail[0][:48])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 8, 16, 100, 101])
def test_gpu_send_batch_signal_path_matches_oracle(cfg):
    from mercury_amd import RxPhy
    from mercury_amd.physical_layer import BATCH_MESSAGE
    orc = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=8)
    rng = np.random.default_rng(300 + cfg)
    F = 5
    pls = rng.integers(0, 256, (F, orc.payload_bytes)).astype(np.uint8)
    nb = np.array([orc.payload_bytes, 3, orc.payload_bytes // 2, orc.payload_bytes, 0], np.int32)
    got = rx.transmit_byte(pls, CARRIER, nbytes=nb, message_location=BATCH_MESSAGE, start_sample=4242)
    assert np.array_equal(got, orc.transmit_batch(pls.astype(np.int32), nb, start_sample=4242))
    one = rx.transmit_byte(pls[:1], CARRIER, message_location=BATCH_MESSAGE)              # a batch of one: padded with itself
    assert np.array_equal(one, orc.transmit_batch(pls[:1].astype(np.int32)))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [100, 101, 102])
def test_gpu_transmit_byte_mfsk_control_frames(cfg):
    from mercury_amd import RxPhy
    orc = Oracle(cfg)
    orc.set_ctrl_mode(1)
    rx = RxPhy(cfg, max_batch=4, mfsk_ctrl_mode=True)
    pls = np.random.default_rng(cfg).integers(0, 256, (2, orc.payload_bytes)).astype(np.uint8)
    for loc in (SINGLE_MESSAGE, NO_FILTER_MESSAGE):
        got = rx.transmit_byte(pls, CARRIER, message_location=loc)
        for f in range(2):
            want = orc.transmit_byte(pls[f].astype(np.int32), message_location=loc)
            assert np.array_equal(got[f], want), (cfg, loc, f)
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    assert (used < got.shape[1]) == (cfg != 102)                          # ROBUST_2 has no shorter control frame
    assert not got[:, used + 60:].any()                                   # silence behind the short frame (filter tails aside)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 8, 13, 16, 100, 102])
def test_gpu_loopback_transmit_then_receive_byte(cfg):
    """GPU transmit_byte -> capture windows (unknown delay, noise) -> GPU receive_byte gives the messages back."""
    from mercury_amd import RxPhy
    rx = RxPhy(cfg, max_batch=8)
    W = 6
    rng = np.random.default_rng(50 + cfg)
    pls = rng.integers(0, 256, (W, rx.payload_bytes)).astype(np.uint8)
    pb = rx.transmit_byte(pls, CARRIER)
    n = rx.receive_buffer_samples()
    wins = rng.standard_normal((W, n)) * 2e-3
    for w in range(W):
        d = int(rng.integers(2 * rx.Nofdm * 4, n - pb.shape[1] - 2 * rx.Nofdm * 4))
        wins[w, d: d + pb.shape[1]] += 2.0 * pb[w]          # receiver audio gain, see the oracle round trip above
    r = rx.receive_byte(wins, CARRIER)
    ok = r["stats"]["message_decoded"] == 1
    orc = Oracle(cfg)
    assert ok.tolist() == [orc.receive_byte(wins[w])["message_decoded"] == 1 for w in range(W)]    # the CPU chain agrees window by window
    if cfg in (0, 8, 100, 102):
        assert ok.all()
    else:       # 32QAM / zero-forcing modes: clipping + filter ringing + a quarter-sample timing error cost some frames, on the CPU too
        assert ok.sum() >= W // 2
    assert np.array_equal(r["payload"][ok, : rx.payload_bytes], pls[ok])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [100, 101])
def test_gpu_loopback_mfsk_control_frames(cfg):
    """Short MFSK control frames (set_mfsk_ctrl_mode): GPU transmit -> capture windows -> GPU receive_byte, and the CPU chain agrees."""
    from mercury_amd import RxPhy
    rx = RxPhy(cfg, max_batch=4, mfsk_ctrl_mode=True)
    orc = Oracle(cfg)
    orc.set_ctrl_mode(1)
    W = 3
    rng = np.random.default_rng(70 + cfg)
    pls = rng.integers(0, 256, (W, rx.payload_bytes)).astype(np.uint8)
    pb = rx.transmit_byte(pls, CARRIER)
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    n = rx.receive_buffer_samples()
    wins = rng.standard_normal((W, n)) * 2e-3
    sym = rx.Nofdm * 4
    for w in range(W):
        d = int(rng.integers(6, (n - used) // sym - 2)) * sym + int(rng.integers(0, 48))      # past symbol 4 (receive_byte's bounds) and within the guard interval of a
        # symbol slot: time_sync_mfsk resolves whole slots only (ofdm.cc:2058)
        wins[w, d: d + used] += pb[w, :used]
    r = rx.receive_byte(wins, CARRIER)
    assert r["stats"]["message_decoded"].tolist() == [1] * W
    assert np.array_equal(r["payload"][:, : rx.payload_bytes], pls)
    for w in range(W):
        ref = orc.receive_byte(wins[w])
        assert ref["message_decoded"] == 1 and ref["delay"] == r["stats"]["delay"][w] and np.array_equal(ref["payload"], pls[w])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 100])
def test_gpu_ack_and_break_patterns_match_oracle_and_are_detected(cfg):
    """generate_ack_pattern_passband / generate_break_pattern_passband: bit-exact samples, and the library's own detector
    (mgpu_detect_ack_pattern_from_passband) finds each pattern — and not the other one — in a noisy buffer."""
    from mercury_amd import RxPhy
    orc = Oracle(cfg)
    rx = RxPhy(cfg, max_batch=4)
    for which in (1, 2):
        for kw in (dict(), dict(start_sample=987654321, output_power_watt=0.05, data_papr_cut=3.0)):
            got = rx.generate_ack_pattern_passband(which, CARRIER, **kw)
            assert np.array_equal(got, orc.generate_ack_pattern_passband(which, **kw)), (cfg, which, kw)
    rng = np.random.default_rng(cfg)
    n = 40 * orc.Nofdm * 4
    for which in (1, 2):
        x = rng.standard_normal(n) * 0.02
        p = rx.generate_ack_pattern_passband(which, CARRIER)
        x[7000: 7000 + p.size] += p
        m, k = rx.detect_ack_pattern_from_passband(x, CARRIER, pattern=which)
        mo, ko = rx.detect_ack_pattern_from_passband(x, CARRIER, pattern=3 - which)
        # the pattern starts off the detector's symbol raster, so each tone's energy is split over two slots: metric ~6 of 16
        assert float(m[0]) > 4 and int(k[0]) >= 12 and int(ko[0]) < 6 and float(mo[0]) < 0.5 * float(m[0]), (cfg, which, m, k, mo, ko)


@pytest.mark.gpu
def test_gpu_symbol_mod_is_the_unnormalised_ifft_with_guard_interval():
    from mercury_amd import RxPhy
    rx = RxPhy(8, max_batch=4)
    rng = np.random.default_rng(3)
    car = rng.standard_normal((7, 50)) + 1j * rng.standard_normal((7, 50))
    y = rx.symbol_mod(car)
    assert y.shape == (7, 272)
    bins = np.zeros((7, 256), np.complex128)
    bins[:, 231:256] = car[:, :25]                  # zero_padder (ofdm.cc:379-400), start_shift = 1
    bins[:, 1:26] = car[:, 25:]
    t = np.fft.ifft(bins, axis=1) * 256
    assert np.allclose(y[:, 16:], t, atol=1e-12) and np.array_equal(y[:, :16], y[:, 256:])
    # and symbol_demod undoes it exactly up to rounding
    back = rx.stage_symbol_demod(y) if hasattr(rx, "stage_symbol_demod") else None
    if back is not None:
        assert np.allclose(back, car, atol=1e-12)


@pytest.mark.gpu
def test_gpu_transmit_byte_bad_arguments():
    from mercury_amd import MgpuError, RxPhy
    rx = RxPhy(8, max_batch=4)
    pl = np.zeros((1, rx.payload_bytes), np.uint8)
    with pytest.raises(MgpuError):
        rx.transmit_byte(pl, CARRIER, message_location=5)                 # no such message location
    with pytest.raises(MgpuError):
        rx.transmit_byte(pl[:, :10], CARRIER)                              # rows shorter than the frame's payload
    with pytest.raises(MgpuError):
        rx.transmit_byte(pl, CARRIER, nbytes=np.array([rx.payload_bytes + 1], np.int32))   # "message too long.. not sent."
    with pytest.raises(MgpuError):
        rx.transmit_byte(pl, 30000.0)                                      # carrier above Nyquist
    assert rx.transmit_byte(pl, CARRIER).shape == (1, rx.transmit_frame_samples())   # the context still works
