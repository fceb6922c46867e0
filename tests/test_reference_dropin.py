"""The drop-in, executed from the reference's side (INTEGRATION.md 1.2 / 1.3b in compiled form).

oracle/_ref/libmercury_ref_ts_gpu.so is the reference's own cl_telecom_system - the same unmodified object code as libmercury_ref_ts.so -
in which the methods section 8b lists (cl_ofdm::symbol_demod ... cl_ldpc::decode, the free deinterleaver / bit_to_byte / CRC16 functions and
cl_telecom_system::receive_byte) were made weak with objcopy and are defined again by oracle/ref_ts_gpu_harness.cc on top of
libmercury_gpu.so's C-ABI. Here the reference's OWN callers run over them:

  * cl_telecom_system::baseband_test_EsN0 (telecom_system.cc:95-229), the BER loop: same cl_error_rate counts and the same stage buffers in
    data_container as the untouched object given the same random streams, on all 17 OFDM modes;
  * cl_telecom_system::receive_byte (telecom_system.cc:646-1503) with its hot span (:1132-1345) on the GPU methods, and as a whole on
    mgpu_receive_byte_batch: the same st_receive_stats, payload and cross-call members as the untouched object on all 20 modes;
  * cl_telecom_system::RX_RAND_process_main (telecom_system.cc:2102-2190): the reference's receive loop prints the bytes the reference's
    transmit_byte sent, the same text as the untouched object prints.

In SHADOW mode every replaced call additionally runs the original machine code on the same inputs and compares the outputs bit for bit.
Test infrastructure only: the product never links these libraries."""
import re

import numpy as np
import pytest

import oraclelib
from oraclelib import (CARRIER, MODE_REFERENCE, MODE_SHADOW, MODE_STAGES, MODE_WHOLE, LinkState, Oracle, RefTelecomSystem, RefTelecomSystemGpu)
from test_receive_byte_vs_reference import ALL_CFGS, FLOAT_FIELDS, INT_FIELDS, STATE_FIELDS, windows

pytestmark = pytest.mark.skipif(not (RefTelecomSystem.available() and RefTelecomSystemGpu.available()),
                                reason="oracle/_ref/libmercury_ref_ts[_gpu].so not built (needs /root/reference)")

STAGE_KEYS = ("baseband", "grid", "eq", "syms", "llr_demod", "llr_ldpc", "data_bits", "decoded_bits", "err")
CHAIN = ("symbol_demod", "channel_estimator", "channel_equalizer", "measure_variance", "deframer", "deinterleaver_c128", "deinterleaver_f32",
         "psk_demod", "ldpc_decode")
TAIL = ("bit_energy_dispersal", "bit_to_byte", "crc16")


# ---- without a GPU: the interposition itself --------------------------------------------------------------------------------------------

@pytest.mark.parametrize("cfg", [0, 8, 16, 101])
def test_interposed_library_in_pass_through_mode_is_the_reference(cfg):
    """mode 0: every replaced symbol is reached (the counters move) and falls through to the original machine code, so the object behaves
    exactly as libmercury_ref_ts.so's - the BER loop's buffers and receive_byte on randomised windows."""
    a, b = RefTelecomSystem(cfg), RefTelecomSystemGpu(cfg, MODE_REFERENCE)
    c0 = b.counters()
    assert c0["symbol_demod"][0] == (1000 if cfg < 100 else 0)          # get_pre_equalization_channel's calls (telecom_system.cc:3129-3133) came through
    if cfg < 100:
        for esn0 in (3.0, -15.0):
            a.seed(11 + cfg); ra = a.baseband_test_one_frame(esn0)
            b.seed(11 + cfg); rb = b.baseband_test_one_frame(esn0)
            for k in STAGE_KEYS:
                assert np.array_equal(ra[k], rb[k]), (cfg, esn0, k)
    orc = Oracle(cfg)
    rng = np.random.default_rng(9000 + cfg)
    for w, (kind, x, call, state, carrier) in enumerate(windows(orc, rng, 4)):
        sa, sb = LinkState(*state), LinkState(*state)
        qa, qb = a.receive_byte(x, carrier=carrier, state=sa, **call), b.receive_byte(x, carrier=carrier, state=sb, **call)
        for k in INT_FIELDS + FLOAT_FIELDS:
            assert qa[k] == qb[k] or (qa[k] != qa[k] and qb[k] != qb[k]), (cfg, w, kind, k)
        assert np.array_equal(qa["payload"], qb["payload"])
    c = b.counters()
    assert c["receive_byte"][0] == 4 and c["ldpc_decode"][0] >= 1 and all(v[1] == 0 and v[2] == 0 for v in c.values()), c
    a.close(); b.close()


# ---- on the GPU -----------------------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("cfg", list(range(17)))
def test_reference_ber_loop_runs_on_the_gpu_methods(cfg):
    """cl_telecom_system::baseband_test_EsN0(EsN0, 1), the reference's own object code, over the GPU-backed methods: cl_error_rate's counts
    and every stage buffer it left in data_container equal the untouched object's, frame for frame, at four Es/N0 points (one of them
    non-converging); each replaced call was served by the GPU and agreed bit for bit with the original machine code run beside it."""
    a, b = RefTelecomSystem(cfg), RefTelecomSystemGpu(cfg, MODE_STAGES | MODE_SHADOW)
    assert b.counters()["symbol_demod"][:2] == (1000, 1000)            # get_pre_equalization_channel was served by the GPU as well
    assert np.array_equal(a.pre_equalization_channel(), b.pre_equalization_channel())
    b.counters(reset=True)
    from conftest import OPERATING_ESN0
    op = OPERATING_ESN0[cfg]
    frames = 0
    for esn0 in (op, op + 1.0, op - 1.0, -15.0):
        for rep in range(2):
            s = 1000 * cfg + 10 * int(esn0 + 20) + rep
            a.seed(s); ra = a.baseband_test_one_frame(esn0)
            b.seed(s); rb = b.baseband_test_one_frame(esn0)
            for k in STAGE_KEYS:
                assert np.array_equal(ra[k], rb[k]), (cfg, esn0, rep, k)
            frames += 1
    c = b.counters()
    info = a.info
    assert c["symbol_demod"] == (frames * info["Nsymb"], frames * info["Nsymb"], 0), c
    for m in CHAIN[1:]:
        assert c[m][0] == c[m][1] >= frames and c[m][2] == 0, (m, c[m])
    if info["amp_restore"]:
        assert c["restore_channel_amplitude"] == (frames, frames, 0) and c["channel_equalizer_without_amplitude_restoration"] == (frames, frames, 0), c
    assert b.error() == ""
    a.close(); b.close()


def _compare_receive(cfg, w, kind, qa, qb, sa, sb, exact_doubles=True):
    for k in INT_FIELDS:
        assert qa[k] == qb[k], (cfg, w, kind, k, qa[k], qb[k])
    for k in FLOAT_FIELDS:
        same = qa[k] == qb[k] or (qa[k] != qa[k] and qb[k] != qb[k])
        assert same or (not exact_doubles and abs(qa[k] - qb[k]) <= 1e-9 * max(1.0, abs(qa[k]))), (cfg, w, kind, k, qa[k], qb[k])
    assert np.array_equal(qa["payload"], qb["payload"]), (cfg, w, kind)
    for k in STATE_FIELDS:
        va, vb = getattr(sa, k), getattr(sb, k)
        assert va == vb or (not exact_doubles and k.startswith("freq") and abs(va - vb) <= 1e-9), (cfg, w, kind, k, va, vb)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_reference_receive_byte_hot_span_runs_on_the_gpu_methods(cfg):
    """cl_telecom_system::receive_byte's own control flow (synchroniser, gates, retry loop: the reference's object code) with the span
    :1132-1345 on the GPU-backed methods - INTEGRATION.md 1.2 as the reference executes it. Randomised capture windows (frames at unknown
    delay / offset, noise, two frames, cut-off frames, carried-in link state): every st_receive_stats field bit for bit, the payload and the
    cross-call members equal the untouched object's."""
    a, b = RefTelecomSystem(cfg), RefTelecomSystemGpu(cfg, MODE_STAGES | MODE_SHADOW)
    b.counters(reset=True)
    orc = Oracle(cfg)
    rng = np.random.default_rng(8200 + cfg)
    decoded = 0
    for w, (kind, x, call, state, carrier) in enumerate(windows(orc, rng, 10)):
        sa, sb = LinkState(*state), LinkState(*state)
        qa, qb = a.receive_byte(x, carrier=carrier, state=sa, **call), b.receive_byte(x, carrier=carrier, state=sb, **call)
        _compare_receive(cfg, w, kind, qa, qb, sa, sb)
        decoded += qa["message_decoded"]
    c = b.counters()
    assert decoded >= 1 and c["receive_byte"] == (10, 0, 0)             # receive_byte itself stayed the reference's
    for m in ("symbol_demod", "ldpc_decode", "deinterleaver_f32") + TAIL:
        assert c[m][0] == c[m][1] >= 1 and c[m][2] == 0, (cfg, m, c[m])
    if cfg < 100:
        for m in CHAIN + ("automatic_gain_control",):
            assert c[m][0] == c[m][1] >= 1 and c[m][2] == 0, (cfg, m, c[m])
    assert b.error() == ""
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ALL_CFGS)
def test_reference_receive_byte_replaced_wholesale(cfg):
    """cl_telecom_system::receive_byte = mgpu_receive_byte_batch inside the reference's object: what its callers get
    back and what it leaves in the receive_stats member equal the untouched object's; doubles bit for bit where the device restates the
    host's libm (the mixer / Moose / signal-level doubles: else to 1e-9, as tests/test_receive_byte.py)."""
    a, b = RefTelecomSystem(cfg), RefTelecomSystemGpu(cfg, MODE_WHOLE)
    orc = Oracle(cfg)
    rng = np.random.default_rng(8350 + cfg)
    decoded = 0
    for w, (kind, x, call, state, carrier) in enumerate(windows(orc, rng, 12)):
        sa, sb = LinkState(*state), LinkState(*state)
        qa, qb = a.receive_byte(x, carrier=carrier, state=sa, **call), b.receive_byte(x, carrier=carrier, state=sb, **call)
        _compare_receive(cfg, w, kind, qa, qb, sa, sb, exact_doubles=False)
        decoded += qa["message_decoded"]
    assert decoded >= 1 and b.counters()["receive_byte"][:2] == (12, 12) and b.error() == ""
    a.close(); b.close()


def _decoded_lines(text):
    """what RX_RAND_process_main itself prints for a decoded frame (telecom_system.cc:2130-2146): the iteration count, the bytes, the statistics"""
    m = re.search(r"Frame decoded in (\d+) iterations\. Data: \n((?:0x[0-9a-f]+, )*)\n(.*)", text)
    return None if not m else (int(m.group(1)), [int(t, 16) for t in m.group(2).replace(",", " ").split()], m.group(3).strip())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,mode", [(8, MODE_WHOLE), (8, MODE_STAGES), (13, MODE_WHOLE), (2, MODE_STAGES), (16, MODE_WHOLE), (100, MODE_WHOLE), (101, MODE_STAGES)])
def test_reference_rx_rand_loop_decodes_what_the_reference_sent(cfg, mode):
    """A bounded RX_RAND session: cl_telecom_system::transmit_byte's frames (the reference's transmitter) in noisy capture windows that slide
    by a few symbols per call, through cl_telecom_system::RX_RAND_process_main - the reference's receive loop, object code untouched - once
    on the untouched methods and once GPU-backed. The loop prints the sent bytes; the text of every decoded window (iterations, bytes,
    sync_trial / peak location / freq_offset / SNR / signal strength as the loop formats them) and the loop's frames_to_read are the same."""
    a, b = RefTelecomSystemGpu(cfg, MODE_REFERENCE), RefTelecomSystemGpu(cfg, mode)
    n = a.buffer_samples()
    rng = np.random.default_rng(70 + cfg)
    sent = rng.integers(0, 256, a.payload_bytes).astype(np.int32)
    pb = a.transmit_byte(sent)
    symbol = a.info["Nofdm"] * 4
    stream = rng.standard_normal(5 * n) * 0.01
    base, slide = 2 * n, max(1, n // symbol // 14) * symbol
    start = base + n // 2 + 3 * symbol + 77
    stream[start: start + pb.size] += pb
    fa = fb = 0
    decoded, attempted = 0, False
    for step in range(14):                                                # the capture window slides over the frame, as the capture thread would move it
        off = base + n // 2 - (7 - step) * slide
        x = stream[off: off + n]
        ta, fa = a.rx_rand_process_main(x, fa)
        tb, fb = b.rx_rand_process_main(x, fb)
        da, db = _decoded_lines(ta), _decoded_lines(tb)
        assert (da is None) == (db is None) and fa == fb, (cfg, step, da, db, fa, fb)
        if da:
            assert da[0] == db[0] and da[1] == db[1], (cfg, step)
            assert da[2] == db[2] or mode == MODE_WHOLE, (cfg, step, da[2], db[2])   # the statistics line: 6 significant digits of the doubles
            assert da[1] == list(sent), (cfg, step)
            decoded += 1
        ha, hb = a.held_receive_stats(), b.held_receive_stats()
        attempted = attempted or ha["iterations_done"] != -1              # crc / all_zeros: not set by the constructor (telecom_system.cc:38-51),
        for k in a.RAW_INTS:                                              # first written by the first trial that reaches the decoder (:1319-1341)
            if k in ("crc", "all_zeros") and not attempted:
                continue
            assert ha[k] == hb[k], (cfg, step, k, ha[k], hb[k])
        fa = fb = 0                                                       # the capture thread counts frames_to_read down to 0 before the next call
    assert decoded >= 2, (cfg, decoded)
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seq", [(8, 100, 9), (7, 101, 12, 8), (13, 102, 15, 16, 100, 13), (0, 100, 6, 10)])
def test_gear_shifts_through_an_mfsk_mode_leave_the_reference_pre_equalisation_table(seq):
    """load_configuration OFDM(A) -> MFSK -> OFDM(B). The reference re-measures pre_equalization_channel on the last hop whatever
    B's modulation (the MFSK load sets the sticky reinit flag, telecom_system.cc:2682-2691; init() measures and clears it only in an OFDM mode,
    :1954-1958); the C++ mirror (mgpu::cl_rx_phy, include/mercury_gpu.hpp) caches the table per (modulation, preamble, carrier, seeds) and must
    drop that key on the MFSK load. The real object and the mirror go through the same sequence; after every OFDM load the tables are equal."""
    import ctypes as C
    ref = RefTelecomSystem(seq[0])
    lib = C.CDLL(oraclelib.REF_TS_GPU_SO, mode=1)
    lib.mmirror_create.restype = C.c_void_p
    m = C.c_void_p(lib.mmirror_create(C.c_int(seq[0]), C.c_int(50)))
    assert m.value
    for cfg in seq[1:] + (seq[0],):
        ref.load_configuration(cfg)
        assert lib.mmirror_load_configuration(m, C.c_int(cfg)) == 0
        if cfg < 100:
            out = np.zeros(50, np.complex128)
            assert lib.mmirror_pre_equalization_channel(m, out.ctypes.data_as(C.c_void_p)) == 50
            assert np.array_equal(out, ref.pre_equalization_channel()), (seq, cfg)
    lib.mmirror_destroy(m)
    ref.close()
