"""SURVEY.md §8 rows f2 and a22 against the reference ITSELF: oracle/_ref/libmercury_ref_ts.so is the reference's own cl_telecom_system
(telecom_system.cc, main.cc, gui/gui_main.cc, audioio/audioio.c and the DSP units compiled unmodified from /root/reference; what stays
undefined and why: oracle/ref_ts_harness.cc). Rounds 1-3 believed telecom_system.cc could not be built in this image and checked the
restatement of receive_byte's control flow (oracle/mercury_oracle.c:morc_receive_byte, which the GPU's mgpu_receive_byte_batch is tested
against window for window) only by reading; round 4 found that it compiles with the reference's own include directories, so here

  * cl_telecom_system::load_configuration (telecom_system.cc:2487-3025) fills the mode table the oracle, the library and SURVEY.md §0 report;
  * cl_telecom_system::receive_byte (telecom_system.cc:646-1503) runs on randomised capture windows - a frame at a random delay and level,
    noise only, two frames, a frame cut off by the end of the window, carrier offsets, last-good delay / frequency offset carried in,
    the +-30 Hz coarse search on and off, 1-3 sync trials, MFSK frames at a known delay - and every output (delay, trial count,
    iteration count, CRC, decoded flag, SNR, frequency offset, Schmidl-Cox metric, signal level: the doubles bit for bit; the payload;
    the cross-call state) equals morc_receive_byte's.

Host-only. Runs where the .so exists (built here by `make -C oracle ref`; it travels to the GPU box with the other checkers)."""
import json
import os

import numpy as np
import pytest

import oraclelib
from oraclelib import LinkState, Oracle, RefTelecomSystem

pytestmark = pytest.mark.skipif(not RefTelecomSystem.available(), reason="oracle/_ref/libmercury_ref_ts.so not built (needs /root/reference)")

ALL_CFGS = list(range(17)) + [100, 101, 102]
INT_FIELDS = ("iterations_done", "crc", "all_zeros", "message_decoded", "delay", "sync_trials", "frame_overflow_symbols")
FLOAT_FIELDS = ("snr_db", "freq_offset", "coarse_metric", "signal_strength_dbm")
STATE_FIELDS = ("delay_of_last_decoded_message", "freq_offset_of_last_decoded_message", "mfsk_search_start", "fixed_delay_plus_one")


@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_mode_table_is_what_the_reference_load_configuration_computes(cfg):
    ref = RefTelecomSystem(cfg)
    orc = Oracle(cfg)
    for k in ("K", "P", "N", "Nsymb", "Nc", "Nfft", "Ngi", "Nofdm", "nData", "nBits", "nVirtual", "nReal", "bit_blk", "tf_blk",
              "preamble_nsymb", "payload_bytes"):
        assert ref.info[k] == getattr(orc, k), (cfg, k, ref.info[k], getattr(orc, k))
    assert ref.buffer_samples() == orc.buffer_samples()
    if cfg < 100:
        assert (ref.info["M"], ref.info["nPilots"], ref.info["estimator"], ref.info["amp_restore"], ref.info["ls_window"]) == (
            orc.M, orc.nPilots, orc.estimator, orc.amp_restore, orc.ls_window), cfg
        here = os.path.dirname(os.path.abspath(__file__))
        tab = json.load(open(os.path.join(here, "golden", "survey_mode_table.json")))["modes"][str(cfg)]
        for k, v in tab.items():
            if k in ref.info:
                assert ref.info[k] == v, (cfg, k)
    ref.close()


def windows(orc, rng, W):
    """The soak's window generator (tests/tools/soak_receive_byte.py) + per-window call parameters."""
    n = orc.buffer_samples()
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    for w in range(W):
        noise = float(10 ** rng.uniform(-3, -0.3))
        kind = str(rng.choice(["frame", "frame", "frame", "noise", "two", "edge", "silence", "tone", "clipped", "far"]))
        if w == 0:
            noise, kind = 1e-3, "frame"                                 # one window every mode decodes
        if kind == "silence":
            noise = float(10 ** rng.uniform(-12, -8))                   # below the reference's 0.001 norm threshold: every metric is forced to 0
        x = rng.standard_normal(n) * noise
        if kind == "tone":                                              # a carrier-like interferer over the frame
            x += float(rng.uniform(0.05, 1.0)) * np.sin(2 * np.pi * float(rng.uniform(300, 2700)) * np.arange(n) / 48000.0)
        pl = rng.integers(0, 256, orc.payload_bytes)
        pb = orc.transmit_byte(pl.astype(np.int32), message_location=int(rng.choice([3, 4]))) * float(rng.uniform(0.8, 4.0))
        d = -1
        if kind == "far":                                               # the frame mixed up by an oscillator error far beyond the fine search
            t = np.arange(pb.size) / 48000.0
            pb = pb * np.cos(2 * np.pi * float(rng.choice([-25.0, 18.0, 40.0])) * t)
        if kind == "clipped":
            pb = np.clip(pb, -0.3 * np.abs(pb).max(), 0.3 * np.abs(pb).max())
        if kind in ("frame", "two", "tone", "clipped", "far"):
            d = int(rng.integers(0, n - used))
            x[d: d + used] += pb[:used]
        if kind == "two" and n > 2 * used + 5000:
            d2 = int(rng.integers(0, n - used))
            x[d2: d2 + used] += pb[:used]
        if kind == "edge":
            d = n - int(rng.integers(used // 4, used))
            x[d:] += pb[: n - d]
        call = dict(trials_max=int(rng.choice([1, 2, 2, 3])), use_last_time=int(rng.random() < 0.8), use_last_freq=int(rng.random() < 0.8),
                    coarse_freq_sync=int(rng.random() < 0.25))
        state = (-1, 0.0, 0, 0)
        if rng.random() < 0.3:                                           # a link that has decoded before
            state = (int(rng.integers(0, n // 2)), float(rng.uniform(-3, 3)), 0, 0)
        if orc.mfsk_M and kind == "frame" and rng.random() < 0.3:       # the BER test's / overflow recapture's known delay (MFSK modes only)
            state = (state[0], state[1], 0, d + 1)
        elif orc.mfsk_M and rng.random() < 0.3:                         # anti-re-decode: the search starts behind the previous preamble
            state = (state[0], state[1], int(rng.integers(0, 40)), 0)
        df = float(rng.choice([0.0, 0.0, 2.5, -7.0]))                   # receiver's carrier against the transmitter's
        if w == 0:
            call, state, df = dict(trials_max=2, use_last_time=1, use_last_freq=1, coarse_freq_sync=0), (-1, 0.0, 0, 0), 0.0
        yield kind, x, call, state, oraclelib.CARRIER + df


def compare_one(orc, ref, x, carrier, call, state):
    sa, sb = LinkState(*state), LinkState(*state)
    a = orc.receive_byte(x, carrier=carrier, state=sa, **call)
    b = ref.receive_byte(x, carrier=carrier, state=sb, **call)
    diff = [k for k in INT_FIELDS if a[k] != b[k]]
    diff += [k for k in FLOAT_FIELDS if np.float64(a[k]).view(np.uint64) != np.float64(b[k]).view(np.uint64) and not (a[k] != a[k] and b[k] != b[k])]
    diff += ["state." + k for k in STATE_FIELDS if getattr(a["state"], k) != getattr(b["state"], k)]
    # `out` is written only by trials that reach the decoder (the reference leaves the caller's array alone otherwise; both start from zeros here)
    if not np.array_equal(a["payload"], b["payload"]):
        diff.append("payload")
    return diff, a, b


@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_receive_byte_control_flow_equals_the_reference(cfg):
    orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
    rng = np.random.default_rng(7000 + cfg)
    decoded = 0
    for w, (kind, x, call, state, carrier) in enumerate(windows(orc, rng, 8)):
        diff, a, b = compare_one(orc, ref, x, carrier, call, state)
        assert not diff, (cfg, w, kind, call, state, diff, [a.get(k) for k in diff if k in a], [b.get(k) for k in diff if k in b])
        decoded += b["message_decoded"]
    assert decoded >= 1, cfg
    ref.close()


GEOMETRIES = [(8, dict(Nsymb=20, Dy=5)), (0, dict(Nsymb=40, Dy=5)), (13, dict(Nsymb=10, Dy=5)),      # the reference's LOW_DENSITY frames (telecom_system.cc:1828-1865)
              (4, dict(Nsymb=45, Dy=3)), (16, dict(Nsymb=8, Dy=4)), (8, dict(Nsymb=18, Dy=9))]


@pytest.mark.parametrize("cfg,geo", GEOMETRIES)
def test_explicit_frame_geometry_equals_the_reference(cfg, geo):
    """ofdm_Nsymb / ofdm_pilot_configurator_Dy set on the reference's own object the way its callers would (the public members of
    default_configurations_telecom_system, in front of load_configuration): the sizes init() derives and receive_byte on randomised windows
    against the oracle created with the same two numbers (morc_create_geometry) - the restatement the GPU's explicit geometry is tested against."""
    orc, ref = Oracle(cfg, explicit=geo), RefTelecomSystem(cfg, geometry=geo)
    for k in ("K", "P", "N", "Nsymb", "Nc", "Nfft", "Ngi", "Nofdm", "nData", "nBits", "nVirtual", "nReal", "bit_blk", "tf_blk", "preamble_nsymb", "payload_bytes"):
        assert ref.info[k] == getattr(orc, k), (cfg, geo, k, ref.info[k], getattr(orc, k))
    assert (ref.info["Nsymb"], ref.info["nPilots"]) == (geo["Nsymb"], orc.nPilots)
    assert ref.buffer_samples() == orc.buffer_samples()
    rng = np.random.default_rng(7100 + cfg)
    decoded = 0
    for w, (kind, x, call, state, carrier) in enumerate(windows(orc, rng, 6)):
        diff, a, b = compare_one(orc, ref, x, carrier, call, state)
        assert not diff, (cfg, geo, w, kind, call, state, diff, [a.get(k) for k in diff if k in a], [b.get(k) for k in diff if k in b])
        decoded += b["message_decoded"]
    assert decoded >= 1, (cfg, geo)
    ref.close()


def test_consecutive_windows_with_the_state_carried_over():
    """A short RX_SHM session: the state one call leaves goes into the next (last good delay and frequency offset steer the next sync)."""
    cfg = 8
    orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
    rng = np.random.default_rng(99)
    sa, sb = LinkState(-1, 0.0, 0, 0), LinkState(-1, 0.0, 0, 0)
    for w, (kind, x, call, _, carrier) in enumerate(windows(orc, rng, 10)):
        call = dict(call, use_last_time=1, use_last_freq=1)
        a = orc.receive_byte(x, carrier=carrier, state=sa, **call)
        b = ref.receive_byte(x, carrier=carrier, state=sb, **call)
        for k in INT_FIELDS + FLOAT_FIELDS:
            assert a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]), (w, kind, k, a[k], b[k])
        for k in STATE_FIELDS:
            assert getattr(sa, k) == getattr(sb, k), (w, kind, k)
    ref.close()


# ---- the other members of cl_telecom_system the C-ABI mirrors, against the real object ------------------------------------------------
@pytest.mark.parametrize("cfg", [0, 3, 8, 11, 13, 16, 100, 102])
def test_transmit_byte_pre_equalization_and_patterns_equal_the_reference(cfg):
    """cl_telecom_system::transmit_byte (telecom_system.cc:343-634) on the object load_configuration left (carrier, 0.1 W, PAPR cuts, the
    pre-equalisation table it measured, the mixer phase at a start sample): filtered and unfiltered audio bit for bit; the table itself
    (get_pre_equalization_channel :3108-3145); generate_ack / break_pattern_passband (:1589-1630)."""
    orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
    assert ref.carrier() == oraclelib.CARRIER
    rng = np.random.default_rng(300 + cfg)
    for loc, start in ((oraclelib.SINGLE_MESSAGE, 0), (oraclelib.NO_FILTER_MESSAGE, 12345), (oraclelib.SINGLE_MESSAGE, 2 ** 33 + 7)):
        msg = rng.integers(0, 256, orc.payload_bytes).astype(np.int32)
        want = ref.transmit_byte(msg, message_location=loc, start_sample=start)
        got = orc.transmit_byte(msg, message_location=loc, start_sample=start, pre_equalize=cfg < 100)
        assert np.array_equal(got, want), (cfg, loc, start, float(np.abs(got - want).max()))
    if cfg < 100:
        assert np.array_equal(orc.get_pre_equalization_channel(), ref.pre_equalization_channel()), cfg
    for which in (1, 2):
        want = ref.generate_pattern(which, start_sample=777)
        got = orc.generate_ack_pattern_passband(which, start_sample=777)
        assert want.size == got.size and np.array_equal(got, want), (cfg, which)
    ref.close()


@pytest.mark.parametrize("cfg", [8, 100])
def test_pattern_detection_and_signal_level_equal_the_reference(cfg):
    """detect_ack / break_pattern_from_passband (telecom_system.cc:1633-1680: FIR_rx_data baseband -> cl_ofdm::detect_ack_pattern) and
    measure_signal_only (:1520-1541) on noisy audio holding a pattern: metric, matched tones and the level in dBm, as doubles."""
    orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
    rng = np.random.default_rng(400 + cfg)
    n = ref.buffer_samples()
    for which, noise in ((1, 0.01), (2, 0.05), (1, 0.3), (2, 1e-4)):
        x = rng.standard_normal(n) * noise
        pat = ref.generate_pattern(which, start_sample=0)
        d = int(rng.integers(0, n - pat.size))
        x[d: d + pat.size] += pat
        for probe in (1, 2):
            bbi = orc.passband_to_baseband(x, oraclelib.CARRIER, 1, 1)
            assert orc.detect_ack_pattern(bbi, probe) == ref.detect_pattern(x, probe), (cfg, which, probe)
        lvl = ref.measure_signal_only(x)
        a = orc.receive_byte(x)                                              # signal_stregth_dbm is the same measurement (:678)
        assert a["signal_strength_dbm"] == lvl, (cfg, which, a["signal_strength_dbm"], lvl)
    ref.close()


def test_configuration_bookkeeping_of_the_reference_object():
    """What load_configuration / return_to_last_configuration (telecom_system.cc:2487-2492, :3027-3034) leave in current_configuration and
    last_configuration - the state include/mercury_gpu.hpp's mirror reproduces: after return_to_last_configuration() the two
    members read as BEFORE the call while the mode actually loaded is the former last_configuration."""
    ref = RefTelecomSystem(8)
    assert ref.load_configuration(3) == (3, 8, 48, 400)
    assert ref.load_configuration(-1) == (3, 8, 24, 600)                   # mode 8 is loaded again; the members still say 3 / 8
    assert ref.load_configuration(-1) == (3, 8, 24, 600)
    assert ref.load_configuration(3) == (3, 8, 24, 600)                    # "already current" by the members (:2489-2492): nothing is loaded
    assert ref.load_configuration(16)[:2] == (16, 3)
    ref.close()


@pytest.mark.parametrize("cfg", [100, 101, 102])
def test_mfsk_control_frames_equal_the_reference(cfg):
    """set_mfsk_ctrl_mode (telecom_system.cc:1572-1585): the short control frames of the ROBUST modes through transmit_byte and receive_byte
    on both sides - the audio bit for bit, and every receive_byte output on windows that hold such a frame, noise, or a frame whose tail lies
    beyond the capture window (frame_overflow_symbols)."""
    orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
    orc.set_ctrl_mode(True)
    assert ref.set_ctrl_mode(True) == orc.active_nsymb
    rng = np.random.default_rng(500 + cfg)
    msg = rng.integers(0, 256, orc.payload_bytes).astype(np.int32)
    # A control frame fills only (preamble + ctrl_nsymb) symbols of passband_data_tx; the reference filters and returns total_frame_size
    # samples all the same, so what lies behind the frame is whatever the buffer held before - the previous full frame, or for a new object
    # whatever `new double[]` handed out (zeros in a fresh process, a freed buffer's content in this one). The ARQ layer plays only the
    # control frame's samples; the comparison covers them up to the reach of the two 97-tap transmit filters in front of that tail.
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4 - 2 * 96
    want = ref.transmit_byte(msg, message_location=oraclelib.SINGLE_MESSAGE)
    assert np.array_equal(orc.transmit_byte(msg, message_location=oraclelib.SINGLE_MESSAGE)[:used], want[:used])
    decoded = 0
    for w, (kind, x, call, state, carrier) in enumerate(windows(orc, rng, 10)):
        diff, a, b = compare_one(orc, ref, x, carrier, call, state)
        assert not diff, (cfg, w, kind, call, state, diff, [a.get(k) for k in diff if k in a], [b.get(k) for k in diff if k in b])
        decoded += b["message_decoded"]
    assert decoded >= 1
    ref.close()


@pytest.mark.parametrize("cfg", [100, 102])
def test_overlap_save_message_locations_equal_the_reference(cfg):
    """transmit_byte's FIRST / MIDDLE / FLUSH_MESSAGE (telecom_system.cc:559-596: the frame goes into a three-frame buffer, two frames around
    its middle are filtered, the buffer shifts left) on the real object, call by call, against the restatement's transmit_stream: the audio of
    every call and the buffer it leaves. The MFSK modes: the stream form of the restatement applies no pre-equalisation, like the reference's
    MFSK path (:475)."""
    orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
    rng = np.random.default_rng(600 + cfg)
    total = (orc.preamble_nsymb + orc.Nsymb) * orc.Nofdm * 4
    buf = rng.standard_normal(3 * total) * 0.01                      # whatever the buffer held: both sides start from the same content
    ref.transmit_buffer(buf)
    start = 99
    for loc in (oraclelib.FIRST_MESSAGE, oraclelib.MIDDLE_MESSAGE, oraclelib.MIDDLE_MESSAGE, oraclelib.FLUSH_MESSAGE, oraclelib.FIRST_MESSAGE):
        msg = rng.integers(0, 256, (1, orc.payload_bytes)).astype(np.int32)
        got, buf = orc.transmit_stream(msg, loc, buffer=buf, start_sample=start)
        want = ref.transmit_byte(msg[0], message_location=loc, start_sample=start)
        assert np.array_equal(got[0], want), (cfg, loc)
        assert np.array_equal(buf, ref.transmit_buffer()), (cfg, loc)
        start += total
    ref.close()


def test_get_configuration_thresholds_of_the_mirror_are_the_reference_function():
    """char cl_telecom_system::get_configuration(double SNR) (telecom_system.cc:3036-3106) on a fine SNR grid against the threshold table of
    include/mercury_gpu.hpp's mirror (the gear-shift rule a drop-in caller relies on)."""
    import re
    hpp = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mercury_gpu.hpp")).read()
    above = [float(v) for v in re.search(r"static const double above\[15\] = \{([^}]*)\}", hpp).group(1).split(",")]
    assert len(above) == 15

    def mirror(snr):
        for i, th in enumerate(above):
            if snr > th:
                return 15 - i
        return 0

    ref = RefTelecomSystem(0)
    ref.lib.mrefts_get_configuration.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_double]
    for snr in list(np.arange(-12.0, 16.0, 0.125)) + above + [v + 1e-9 for v in above] + [v - 1e-9 for v in above]:
        assert ref.lib.mrefts_get_configuration(ref.h, float(snr)) == mirror(float(snr)), snr
    ref.close()


# ---- rows a1-a19 through the reference's own BER loop ------------------------------------------------------------------------------------
STAGES = ("grid", "eq", "syms", "llr_demod", "llr_ldpc")


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


@pytest.mark.parametrize("cfg", list(range(17)))
def test_hot_path_stages_equal_the_reference_ber_loop(cfg):
    """cl_telecom_system::baseband_test_EsN0 (telecom_system.cc:95-229), the reference's own driver of the hot path: one
    frame per call, and the noisy samples it received go through the restatement (FLAGS_BASEBAND_TEST: no AGC, variance from the un-equalised
    pilots). Every stage the real run left in data_container - carrier grid, equalised grid, de-interleaved symbols, demapper LLRs, decoder-input
    LLRs - is bit-identical, and so are the decoder's hard decisions, at the operating point and in the noise (frames that never converge)."""
    from conftest import OPERATING_ESN0
    orc, ref = Oracle(cfg), RefTelecomSystem(cfg)
    for esn0 in (OPERATING_ESN0[cfg], OPERATING_ESN0[cfg] + 1.0, OPERATING_ESN0[cfg] - 3.0):
        r = ref.baseband_test_one_frame(esn0)
        o = orc.rx(r["baseband"], oraclelib.FLAGS_BASEBAND_TEST)
        for k in STAGES:
            assert _same_bits(o[k], r[k]), (cfg, esn0, k)
        assert np.array_equal(o["bits"][: orc.nReal], r["decoded_bits"]), (cfg, esn0, o["iterations"])
        assert r["err"][0] == orc.nReal and r["err"][1] == int((r["data_bits"] != r["decoded_bits"]).sum()), (cfg, esn0, r["err"])
    ref.close()
