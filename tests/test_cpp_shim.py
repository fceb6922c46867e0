"""The C++ host-side mirror of the reference's classes (include/mercury_gpu.hpp) driven from a C++
program the way telecom_system.cc drives cl_ldpc / receive_byte; results checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oraclelib
from conftest import OPERATING_ESN0, SEED

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = tmp_path / "shim_test"
    lib = os.path.join(ROOT, "mercury_amd")
    subprocess.run(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"),
                    "-o", str(exe), "-L", lib, "-lmercury_gpu", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_cpp_shim_compiles_against_the_header(tmp_path):
    """CPU: the C++14 host code (the reference's language level) compiles and links against the C-ABI."""
    assert _build(tmp_path).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 13])
def test_cpp_shim_matches_oracle(tmp_path, cfg):
    exe = _build(tmp_path)
    orc = oraclelib.Oracle(cfg, 50)
    op = OPERATING_ESN0[cfg]
    snrs = [op + 1, op + 2, -15.0, 60.0]
    F = len(snrs)
    frames = [orc.gen_frame(SEED, 900 + i, oraclelib.noise_amp_for(s))[0] for i, s in enumerate(snrs)]
    bb = np.stack(frames)
    refs = [orc.rx(b, oraclelib.FLAGS_RECEIVE_BYTE) for b in bb]
    llr = np.stack([r["llr_ldpc"] for r in refs])
    (tmp_path / "bb.bin").write_bytes(bb.tobytes())
    (tmp_path / "llr.bin").write_bytes(llr.tobytes())
    r = subprocess.run([str(exe), str(cfg), str(F), str(tmp_path / "bb.bin"), str(tmp_path / "llr.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    pb, stride, K = orc.payload_bytes, (orc.nReal + 7) // 8, orc.K
    off = 0

    def take(n, dt):
        nonlocal off
        a = raw[off: off + n * np.dtype(dt).itemsize].view(dt)
        off += n * np.dtype(dt).itemsize
        return a

    for f in range(F):                                   # frame-at-a-time
        rec = take(4, np.int32)
        by = take(pb, np.int32)
        ref = refs[f]
        assert list(rec[:3]) == [ref["iterations"], ref["crc"], ref["all_zeros"]]
        assert np.array_equal(by, ref["bytes"][:pb])
    recs = take(4 * F, np.int32).reshape(F, 4)           # batched
    pay = take(F * stride, np.uint8).reshape(F, stride)
    for f in range(F):
        assert list(recs[f, :3]) == [refs[f]["iterations"], refs[f]["crc"], refs[f]["all_zeros"]]
        assert np.array_equal(pay[f], refs[f]["bytes"].astype(np.uint8))
    if True:                                             # cl_ldpc::decode on the same LLRs
        for f in range(F):
            it = take(1, np.int32)[0]
            bits = take(K, np.int32)
            rb, ri = orc.ldpc_decode(llr[f])
            assert it == ri and np.array_equal(bits, rb)


@pytest.mark.gpu
def test_cpp_receive_byte_keeps_link_state_across_calls(tmp_path):
    """mgpu::cl_rx_phy::receive_byte(double* data, int* out), called once per capture window like RX_SHM_process_main
    does: the last good delay / frequency offset carry over between calls (members of receive_stats, as in the
    reference), which the oracle reproduces when handed the same state."""
    from test_receive_byte import make_windows
    cfg = 8
    exe = _build(tmp_path)
    orc = oraclelib.Oracle(cfg, 50)
    true_delay = 9 * 1088 + 100
    wins = np.concatenate([make_windows(orc, [("frame", true_delay, noise, 2)], seed=5)[0] for noise in (0.01, 0.1, 0.15)])
    bb = np.zeros((1, orc.frame_samples), np.complex128)
    (tmp_path / "bb.bin").write_bytes(bb.tobytes())
    (tmp_path / "llr.bin").write_bytes(np.zeros((1, 1600), np.float32).tobytes())
    (tmp_path / "pass.bin").write_bytes(wins.tobytes())
    r = subprocess.run([str(exe), str(cfg), "1", str(tmp_path / "bb.bin"), str(tmp_path / "llr.bin"), str(tmp_path / "out.bin"),
                        str(tmp_path / "pass.bin"), "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = np.fromfile(str(tmp_path / "out.bin") + ".rb", np.int32).reshape(3, 6 + orc.payload_bytes)
    state = oraclelib.LinkState(-1, 0.0, 0)
    for w in range(3):
        ref = orc.receive_byte(wins[w], state=state)      # state is updated in place, like the reference's members
        assert list(raw[w][:5]) == [ref["iterations_done"], ref["crc"], ref["message_decoded"], ref["delay"], ref["sync_trials"]], w
        assert raw[w][5] == state.delay_of_last_decoded_message
        assert np.array_equal(raw[w][6:], ref["payload"])
    assert list(raw[:, 2]) == [1, 1, 1]                    # the noisy windows decode thanks to the first one's sync state


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 100])
def test_cpp_transmit_byte_runs_the_carrier_on_across_calls(tmp_path, cfg):
    """mgpu::cl_rx_phy::transmit_byte(int* data, int nBytes, double* out, int message_location): consecutive calls continue the
    carrier phase (cl_ofdm::passband_start_sample), a short message is zero-padded, a too-long one is not sent."""
    exe = _build(tmp_path)
    orc = oraclelib.Oracle(cfg, 50)
    pb = orc.payload_bytes
    msgs = np.random.default_rng(cfg).integers(0, 256, (4, pb)).astype(np.int32)
    (tmp_path / "bb.bin").write_bytes(np.zeros((1, orc.frame_samples), np.complex128).tobytes())
    (tmp_path / "llr.bin").write_bytes(np.zeros((1, 1600), np.float32).tobytes())
    win = np.random.default_rng(5).standard_normal(orc.buffer_samples()) * 1e-3      # message 0 in a capture window, receiver gain 2
    pb0 = orc.transmit_byte(msgs[0])
    d0 = 9 * orc.Nofdm * 4 + 77
    win[d0: d0 + pb0.size] += 2.0 * pb0
    (tmp_path / "pass.bin").write_bytes(win.tobytes())
    (tmp_path / "msgs.bin").write_bytes(msgs.tobytes())
    r = subprocess.run([str(exe), str(cfg), "1", str(tmp_path / "bb.bin"), str(tmp_path / "llr.bin"), str(tmp_path / "out.bin"),
                        str(tmp_path / "pass.bin"), "1", str(tmp_path / "msgs.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    total = (orc.preamble_nsymb + orc.Nsymb) * orc.Nofdm * 4
    used = (orc.preamble_nsymb + orc.active_nsymb) * orc.Nofdm * 4
    audio = np.fromfile(str(tmp_path / "out.bin") + ".tx", np.float64).reshape(5, total)
    pre_eq = cfg < 100            # load_configuration measures and installs pre_equalization_channel like init() does (OFDM modes)
    for m in range(4):
        want = orc.transmit_byte(msgs[m, : pb // 2] if m == 1 else msgs[m], start_sample=m * used,
                                 message_location=oraclelib.NO_FILTER_MESSAGE if m == 3 else oraclelib.SINGLE_MESSAGE, pre_equalize=pre_eq)
        assert np.array_equal(audio[m], want), m
    # transmit_bit on the bits transmit_byte makes of message 0: the same frame, four frames further along the carrier
    assert np.array_equal(audio[4], orc.transmit_byte(msgs[0], start_sample=4 * used, pre_equalize=pre_eq))
    # receive_bit on a window that holds message 0: the de-scrambled decoded bits, CRC included, LSB first
    rb = np.fromfile(str(tmp_path / "out.bin") + ".rbits", np.int32)
    assert rb[0] == 1
    assert np.array_equal(rb[1:], orc.payload_to_bits(msgs[0])[: (orc.nReal // 8) * 8])
    # the signalling calls: the ACK pattern it generates is the one it detects (and not BREAK), the level of a buffer that holds only
    # that pattern, and the control-frame switch (MFSK modes only)
    metric, m_ack, m_brk, dbm, data_nsymb, ctrl_nsymb = np.fromfile(str(tmp_path / "out.bin") + ".sig", np.float64)
    assert metric > 4 and m_ack >= 12 and m_brk < m_ack - 4      # silence around the pattern lets a few BREAK slots match by chance
    assert -40 < dbm < 20
    orc.set_ctrl_mode(1)
    assert (int(data_nsymb), int(ctrl_nsymb)) == (orc.Nsymb, orc.active_nsymb if cfg >= 100 else orc.Nsymb)
    # baseband_test_EsN0 / passband_test_EsN0 through the C++ mirror: counters of cl_error_rate, every frame of these clean points decodes
    b_frames, b_err, b_bits, p_frames, p_err, p_ok = np.fromfile(str(tmp_path / "out.bin") + ".ber", np.float64)
    assert (b_frames, b_err) == (4, 0) and b_bits > 0
    assert (p_frames, p_err, p_ok) == (3, 0, 3)


def _build_stages(tmp_path):
    exe = tmp_path / "stages_test"
    lib = os.path.join(ROOT, "mercury_amd")
    subprocess.run(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "stages_test.cpp"),
                    "-o", str(exe), "-L", lib, "-lmercury_gpu", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_cpp_per_method_mirror_compiles(tmp_path):
    assert _build_stages(tmp_path).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [0, 8, 10, 13, 16])
def test_cpp_per_method_sequence_matches_oracle(tmp_path, cfg):
    """receive_byte's front half written method by method (mgpu::cl_ofdm / cl_psk / deinterleaver, tests/cpp/stages_test.cpp)
    gives, stage by stage, what the oracle's receive_byte variant gives, bit for bit (carrier grid, channel grid, equalised grid,
    float variance, symbols, LLRs) like the fused path."""
    exe = _build_stages(tmp_path)
    orc = oraclelib.Oracle(cfg, 50)
    bb, _ = orc.gen_frame(SEED, 4000 + cfg, oraclelib.noise_amp_for(OPERATING_ESN0[cfg] + 2.0))
    (tmp_path / "bb.bin").write_bytes(bb.tobytes())
    r = subprocess.run([str(exe), str(cfg), str(tmp_path / "bb.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ref = orc.rx(bb, oraclelib.FLAGS_RECEIVE_BYTE | oraclelib.FLAG_NO_LDPC)
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    G, nData, nBits = orc.Nsymb * orc.Nc, orc.nData, orc.nBits
    off = 0

    def take(n, dt):
        nonlocal off
        a = raw[off: off + n * np.dtype(dt).itemsize].view(dt)
        off += n * np.dtype(dt).itemsize
        return a

    grid, H, eq = take(G, np.complex128), take(G, np.complex128), take(G, np.complex128)
    variance = take(1, np.float32)[0]
    syms, llr_demod, llr_deint = take(nData, np.complex128), take(nBits, np.float32), take(nBits, np.float32)
    assert grid.tobytes() == ref["grid"].tobytes()
    # stages.hip uses the same restated libm (csrc/glibc_trig.h) and the same operation order as the fused front-end: the per-method
    # path is held to the same bar - BIT-IDENTICAL wherever the host runs the libm build that was restated (tests/test_gpu_parity.py:
    # EXACT_TRIG), last-ulp phasor differences otherwise in the modes that restore the amplitude
    from test_gpu_parity import EXACT_TRIG
    exact = EXACT_TRIG or not orc.amp_restore
    rel = 0.0 if exact else 1e-12
    assert np.abs(H - ref["H"]).max() <= rel * np.abs(ref["H"]).max()
    assert np.abs(eq - ref["eq"]).max() <= rel * np.abs(ref["eq"]).max()
    if cfg < 15:     # the ZF modes' equalised-pilot variance is rounding noise (~1e-33) in this variant: not comparable
        nreal = orc.nReal
        if exact:
            assert np.float32(variance) == np.float32(ref["variance_f"])
            assert syms.tobytes() == ref["syms"].tobytes()
            assert np.array_equal(llr_demod, ref["llr_demod"], equal_nan=True)
            assert np.array_equal(llr_deint[:nreal], ref["llr_ldpc"][:nreal], equal_nan=True)
        assert abs(variance - ref["variance_f"]) <= 2e-7 * ref["variance_f"]
        assert np.abs(syms - ref["syms"]).max() <= 1e-12 * np.abs(ref["syms"]).max()
        tol = 1e-5 * np.maximum(1.0, np.abs(ref["llr_demod"]))
        assert (np.abs(llr_demod - ref["llr_demod"]) <= tol).all()
        assert (np.abs(llr_deint[:nreal] - ref["llr_ldpc"][:nreal]) <= 1e-5 * np.maximum(1.0, np.abs(ref["llr_ldpc"][:nreal]))).all()
    # integer tail: bit-exact
    nreal = orc.nReal
    if cfg >= 15:
        off = G * 16 * 3 + 4 + nData * 16 + nBits * 4 * 2
    desc, by, crc = take(nreal, np.int32), take((nreal + 7) // 8, np.int32), take(1, np.int32)[0]
    bits = np.array([(i * 7 + i // 3) & 1 for i in range(nreal)], np.int32)
    want = bits ^ orc.scrambler()[:nreal]
    assert np.array_equal(desc, want)
    packed = np.zeros((nreal + 7) // 8, np.int32)
    for i in range(nreal):
        packed[i // 8] |= int(want[i]) << (i % 8)
    assert np.array_equal(by, packed)
    assert crc == orc.crc16(packed[: nreal // 8])


def _build_b2b(tmp_path):
    exe = tmp_path / "back_to_back_test"
    lib = os.path.join(ROOT, "mercury_amd")
    subprocess.run(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "back_to_back_test.cpp"),
                    "-o", str(exe), "-L", lib, "-lmercury_gpu", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_back_to_back_test_compiles(tmp_path):
    assert _build_b2b(tmp_path).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 13, 16, 101])
def test_calls_straight_after_one_another_return_the_same_bytes(tmp_path, cfg):
    """A C++ host calls mgpu_receive_byte_batch (device windows of alternating large / tiny batch sizes, host windows through the one-piece
    and the pipelined path) and mgpu_measure_signal_only with no pause between the calls: every repetition must return the bytes of the
    paused single-window reference. Round 5's use-after-return (a call left its last kernel in flight; NOTES R5.4) made the tiny call after a
    large one fault or differ; the Python tests cannot see this class of defect because the interpreter always pauses long enough."""
    from test_receive_byte import SPECS, make_windows
    exe = _build_b2b(tmp_path)
    orc = oraclelib.Oracle(cfg)
    wins, _ = make_windows(orc, SPECS, seed=500 + cfg)
    (tmp_path / "w.bin").write_bytes(np.ascontiguousarray(wins).tobytes())
    r = subprocess.run([str(exe), str(cfg), str(tmp_path / "w.bin"), str(len(SPECS))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr[-2000:])
    assert "0 differing windows" in r.stdout, r.stdout
