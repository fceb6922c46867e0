"""GPU parity tests for the MFSK modes (ROBUST_0..2 = cfg 100..102): csrc/mfsk.hip + the shared LDPC kernels,
through the C-ABI, against the CPU oracle and the committed reference vectors (tests/golden/golden_mfsk.*).

Everything on this path is required BIT-exact, the LLRs included: the demapper is sums, products, one divide
and comparisons in the reference's order (mfsk.cc:288-390)."""
import hashlib
import json
import os

import numpy as np
import pytest

import oraclelib
from conftest import MFSK_CFGS, OPERATING_ESN0, SEED
from oraclelib import Oracle, noise_amp_for

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _rx(cfg, **kw):
    from mercury_amd import RxPhy
    return RxPhy(cfg, **kw)


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("ctrl", [0, 1])
@pytest.mark.parametrize("cfg", MFSK_CFGS)
def test_mfsk_all_stages_bit_exact(cfg, ctrl):
    orc = Oracle(cfg, 50)
    orc.set_ctrl_mode(ctrl)
    op = OPERATING_ESN0[cfg]
    snrs = [op, op + 1.0, op - 1.5, op - 3.0, -20.0, 60.0]
    frames = [orc.gen_frame(SEED, 50 + i, noise_amp_for(s)) for i, s in enumerate(snrs)]
    bb = np.stack([f[0] for f in frames])
    rx = _rx(cfg, max_iters=50, max_batch=len(snrs), mfsk_ctrl_mode=bool(ctrl))
    for n in ("K", "P", "Nsymb", "nData", "nBits", "nVirtual", "nReal", "bit_blk", "payload_bytes", "mfsk_M", "mfsk_nStreams",
              "active_nsymb", "active_nbits", "frame_samples"):
        assert getattr(rx, n) == getattr(orc, n), n
    out = rx.receive(bb, taps=True)
    n = orc.active_nsymb * orc.Nc
    for f, snr in enumerate(snrs):
        ref = orc.rx(bb[f])
        assert out["grid"][f][:n].tobytes() == ref["grid"][:n].tobytes(), (cfg, ctrl, f, "grid")
        assert out["llr_demod"][f].tobytes() == ref["llr_demod"].tobytes(), (cfg, ctrl, f, "llr_demod")
        assert out["llr_ldpc"][f].tobytes() == ref["llr_ldpc"].tobytes(), (cfg, ctrl, f, "llr_ldpc")
        st = out["stats"][f]
        assert (st["iterations_done"], st["crc"], st["all_zeros"]) == (ref["iterations"], ref["crc"], ref["all_zeros"]), (cfg, ctrl, f)
        assert np.array_equal(out["payload"][f], ref["bytes"].astype(np.uint8)), (cfg, ctrl, f)
        assert st["snr_db"] == np.float32(ref["snr_db"]) and st["variance"] == 0.0
        assert st["message_decoded"] == int(ref["all_zeros"] == 0 and ref["crc"] == 0)
        if snr == 60.0:
            assert st["message_decoded"] == 1 and np.array_equal(out["payload"][f][: orc.payload_bytes], frames[f][1].astype(np.uint8))
    assert out["stats"]["iterations_done"][4] == 51          # -20 dB never converges
    rx.close()


@pytest.mark.parametrize("cfg", MFSK_CFGS)
def test_mfsk_against_committed_reference_vectors(cfg):
    """Outputs of the reference's own objects (oracle/_ref at fixture-generation time), not of the oracle."""
    meta = json.load(open(os.path.join(HERE, "golden", "golden_mfsk.json")))["modes"][str(cfg)]
    arr = np.load(os.path.join(HERE, "golden", "golden_mfsk.npz"))
    orc = Oracle(cfg, 50)            # input generator only
    for idx, rec in enumerate(meta["frames"]):
        orc.set_ctrl_mode(rec["ctrl_mode"])
        bb, _ = orc.gen_frame(SEED, rec["frame"], noise_amp_for(rec["esn0_db"]), rec["channel"])
        assert _digest(bb) == rec["input_sha256"]
        rx = _rx(cfg, max_iters=50, max_batch=1, mfsk_ctrl_mode=bool(rec["ctrl_mode"]))
        out = rx.receive(bb[None, :], taps=True)
        n = rec["active_nsymb"] * rx.Nc
        key = "cfg%d_f%d" % (cfg, idx)
        assert _digest(out["grid"][0][:n]) == rec["sha256"]["grid"]
        assert _digest(out["llr_demod"][0]) == rec["sha256"]["llr_demod"]
        assert out["llr_ldpc"][0].tobytes() == arr[key + "_llr_ldpc"].tobytes()
        assert np.array_equal(out["payload"][0], arr[key + "_bytes"])
        st = out["stats"][0]
        assert (st["iterations_done"], st["crc"], st["all_zeros"]) == (rec["iterations"], rec["crc"], rec["all_zeros"])
        assert st["snr_db"] == np.float32(rec["snr_db"])
        rx.close()


@pytest.mark.parametrize("cfg", MFSK_CFGS)
def test_mfsk_txgen_matches_cpu_generator_and_round_trips(cfg):
    import torch
    orc = Oracle(cfg, 50)
    F = 300
    rx = _rx(cfg, max_iters=50, max_batch=F)
    dev = torch.device("cuda:0")
    bb = torch.empty((F, rx.frame_samples, 2), dtype=torch.float64, device=dev)
    pl = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
    payload = torch.empty((F, rx.payload_stride), dtype=torch.uint8, device=dev)
    stats = torch.empty((F, 24), dtype=torch.uint8, device=dev)
    na = noise_amp_for(OPERATING_ESN0[cfg] + 1.0)
    s = torch.cuda.current_stream().cuda_stream
    rx.txgen_dev(SEED, 1000, F, na, bb.data_ptr(), pl.data_ptr(), stream=s)
    rx.receive_dev(bb.data_ptr(), F, payload.data_ptr(), stats.data_ptr(), stream=s)
    torch.cuda.synchronize()
    host = bb.cpu().numpy().view(np.complex128).reshape(F, -1)
    for f in (0, 1, F - 1):
        ref_bb, ref_pl = orc.gen_frame(SEED, 1000 + f, na)
        # the AWGN term goes through device log/cos: identical algorithmically, last-ulp differences allowed
        assert np.abs(host[f] - ref_bb).max() <= 1e-9 * np.abs(ref_bb).max()
        assert np.array_equal(pl[f].cpu().numpy()[: orc.payload_bytes], ref_pl.astype(np.uint8))
    from mercury_amd import STATS_DTYPE
    st = stats.cpu().numpy().view(STATS_DTYPE).reshape(F)
    ok = st["message_decoded"] == 1
    assert ok.mean() > 0.97
    assert np.array_equal(payload.cpu().numpy()[ok], pl.cpu().numpy()[ok])      # round trip: decoded payload == sent payload
    rx.close()


def test_mfsk_minsum_and_ragged_batches():
    from mercury_amd import DEC_MINSUM
    cfg = 101
    orc = Oracle(cfg, 50)
    for F in (1, 17, 65):
        frames = [orc.gen_frame(SEED, 9000 + i, noise_amp_for(OPERATING_ESN0[cfg] + 2.0)) for i in range(F)]
        bb = np.stack([f[0] for f in frames])
        rx = _rx(cfg, max_iters=50, max_batch=F, decoder=DEC_MINSUM)
        out = rx.receive(bb)
        ok = out["stats"]["message_decoded"] == 1
        assert ok.mean() > 0.9
        for f in np.nonzero(ok)[0]:
            assert np.array_equal(out["payload"][f][: orc.payload_bytes], frames[f][1].astype(np.uint8))
        rx.close()


@pytest.mark.parametrize("cfg", [100, 101])
def test_mfsk_degenerate_inputs_behave_like_the_reference(cfg):
    """Zero, denormal, huge, NaN / Inf polluted frames through the MFSK demapper: the isfinite() guards of
    mfsk.cc:310-316, :331, :381 and the 1e-30 noise floor are reproduced bit for bit."""
    orc = Oracle(cfg, 50)
    n = orc.frame_samples
    good, _ = orc.gen_frame(5, 1, noise_amp_for(OPERATING_ESN0[cfg] + 5.0))
    cases = [np.zeros(n, np.complex128), good * 1e-300, good * 1e150, good.copy(), good.copy(), -good, good * 1e160]
    cases[3][100] = np.nan
    cases[4][200] = np.inf
    rx = _rx(cfg, max_iters=50, max_batch=len(cases))
    with np.errstate(all="ignore"):
        out = rx.receive(np.stack(cases), taps=True)
        for i, x in enumerate(cases):
            ref = orc.rx(x)
            assert out["llr_demod"][i].tobytes() == ref["llr_demod"].tobytes(), (cfg, i)      # no NaN ever reaches the LLRs
            assert out["llr_ldpc"][i].tobytes() == ref["llr_ldpc"].tobytes(), (cfg, i)
            st = out["stats"][i]
            assert (st["iterations_done"], st["crc"], st["all_zeros"]) == (ref["iterations"], ref["crc"], ref["all_zeros"]), (cfg, i)
            assert np.array_equal(out["payload"][i], ref["bytes"].astype(np.uint8)), (cfg, i)
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,cut", [(100, 1000), (101, 900), (102, 1500), (100, 1599)])
def test_test_puncture_nbits_hook_matches_oracle(cfg, cut):
    """cl_telecom_system::test_puncture_nBits (telecom_system.cc:1186-1192, the punctured-LDPC BER-test hook): demodulated LLRs
    from the cut on are erasures. Decoder-input LLRs, payload, iteration count and CRC must equal the oracle's with the same
    hook set (the oracle's hook is checked against the compiled reference objects in test_oracle_vs_ref.py)."""
    from mercury_amd import RxPhy
    orc = oraclelib.Oracle(cfg, 50)
    orc.set_test_puncture(cut)
    F = 4
    bb = np.stack([orc.gen_frame(SEED, 8100 + i, oraclelib.noise_amp_for(OPERATING_ESN0[cfg] + 2.0))[0] for i in range(F)])
    rx = RxPhy(cfg, max_batch=F, test_puncture_nbits=cut)
    out = rx.receive(bb, taps=True)
    plain = RxPhy(cfg, max_batch=F).receive(bb, taps=True)
    for f in range(F):
        ref = orc.rx(bb[f], oraclelib.FLAGS_RECEIVE_BYTE)
        assert np.array_equal(out["llr_demod"][f], ref["llr_demod"]) and np.array_equal(out["llr_ldpc"][f], ref["llr_ldpc"])
        assert not out["llr_demod"][f][cut:].any() and np.array_equal(out["llr_demod"][f][:cut], plain["llr_demod"][f][:cut])
        assert np.array_equal(out["payload"][f], ref["bytes"].astype(np.uint8))
        assert (out["stats"]["iterations_done"][f], out["stats"]["crc"][f]) == (ref["iterations"], ref["crc"])
    rx.close()
