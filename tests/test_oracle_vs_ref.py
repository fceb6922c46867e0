"""CPU tests: the C restatement against the reference's own compiled objects (oracle/_ref), when the
prebuilt library is present (it is built from /root/reference in the build container and travels as a
.so). Fresh random frames, every stage bit-identical."""
import numpy as np
import pytest

import oraclelib
from conftest import OPERATING_ESN0

pytestmark = pytest.mark.skipif(not oraclelib.RefLib.available(), reason="oracle/_ref not built (no /root/reference here)")


@pytest.mark.parametrize("cfg", [0, 4, 8, 10, 13, 16])
def test_stage_by_stage_identical(cfg):
    orc, ref = oraclelib.Oracle(cfg, 50), oraclelib.RefLib(cfg, 50)
    for n in oraclelib.INFO_FIELDS:
        if n != "dwidth":
            assert getattr(orc, n) == getattr(ref, n)
    op = OPERATING_ESN0[cfg]
    for i, snr in enumerate((op, op - 1.5, 40.0)):
        bb, pl = orc.gen_frame(991, 7 * cfg + i, oraclelib.noise_amp_for(snr), channel=i % 2)
        bits = orc.payload_to_bits(pl)
        assert np.array_equal(bits, ref.payload_to_bits(pl))
        assert orc.tx(bits, 1).tobytes() == ref.tx(bits, 1).tobytes()
        assert orc.tx(bits, 0).tobytes() == ref.tx(bits, 0).tobytes()
        for flags in (oraclelib.FLAGS_BASEBAND_TEST, oraclelib.FLAGS_RECEIVE_BYTE):
            a, b = orc.rx(bb, flags), ref.rx(bb, flags)
            for k in a:
                if isinstance(a[k], np.ndarray):
                    assert a[k].tobytes() == b[k].tobytes(), (cfg, snr, flags, k)
                elif k != "agc_gain":
                    assert a[k] == b[k] or (np.isnan(a[k]) and np.isnan(b[k])), (cfg, snr, flags, k)


@pytest.mark.parametrize("cfg", [1, 6, 12])
def test_ldpc_decode_identical(cfg):
    orc, ref = oraclelib.Oracle(cfg, 50), oraclelib.RefLib(cfg, 50)
    rng = np.random.default_rng(cfg)
    for sigma in (0.7, 1.0):
        llr = (2.0 * (1.0 + sigma * rng.standard_normal(1600)) / sigma ** 2).astype(np.float32)  # all-zero codeword
        b1, i1 = orc.ldpc_decode(llr)
        b2, i2 = ref.ldpc_decode(llr)
        assert i1 == i2 and np.array_equal(b1, b2)


@pytest.mark.parametrize("cfg", [0, 8, 16])
def test_the_reference_decoder_changes_no_bit_of_a_hard_word(cfg):
    """What the GPU decoder's hard-frame shortcut rests on (ldpc.hip, NOTES R5.8), shown on the reference's own cl_ldpc::decode: a word whose
    LLRs are all >= 200 in magnitude (what the zero-forcing modes hand over behind RX_SHM) comes back with the signs it went in with, after 0
    iterations when its parity checks hold and after all of them (max + 1) when they do not - the worst case included: every magnitude AT the
    threshold and a fifth of the signs wrong, so that the checks push against the channel values as hard as they can."""
    orc, ref = oraclelib.Oracle(cfg, 50), oraclelib.RefLib(cfg, 50)
    rng = np.random.default_rng(77 + cfg)
    for mag, wrong in ((200.0, 0.2), (200.0, 0.02), (1e30, 0.05), (np.inf, 0.05), (250.0, 0.0)):
        signs = np.where(rng.random(1600) < wrong, -1.0, 1.0)
        llr = (signs * mag).astype(np.float32)                      # the all-zero codeword with sign errors
        for dec in (orc, ref):
            bits, it = dec.ldpc_decode(llr)
            assert it == (0 if wrong == 0.0 else 51), (cfg, mag, wrong, it)
            assert np.array_equal(np.asarray(bits[: dec.K]) != 0, signs[: dec.K] < 0), (cfg, mag, wrong)


@pytest.mark.parametrize("cfg", [100, 101, 102])
def test_mfsk_modes_identical(cfg):
    """ROBUST_0..2: cl_mfsk::mod / demod (mfsk.cc:232-390) and the MFSK branch of receive_byte, full and control frames."""
    orc, ref = oraclelib.Oracle(cfg, 50), oraclelib.RefLib(cfg, 50)
    op = OPERATING_ESN0[cfg]
    for ctrl in (0, 1):
        orc.set_ctrl_mode(ctrl)
        ref.set_ctrl_mode(ctrl)
        for n in oraclelib.INFO_FIELDS:
            if n != "dwidth":
                assert getattr(orc, n) == getattr(ref, n), n
        for i, snr in enumerate((op, op - 2.0, 40.0)):
            bb, pl = orc.gen_frame(77, 10 * cfg + 3 * ctrl + i, oraclelib.noise_amp_for(snr))
            bits = orc.payload_to_bits(pl)
            assert np.array_equal(bits, ref.payload_to_bits(pl))
            assert orc.tx(bits, 1).tobytes() == ref.tx(bits, 1).tobytes()
            a, b = orc.rx(bb), ref.rx(bb)
            n = orc.active_nsymb * orc.Nc
            assert a["grid"][:n].tobytes() == b["grid"][:n].tobytes()
            for k in ("llr_demod", "llr_ldpc", "bits", "bytes"):
                assert a[k].tobytes() == b[k].tobytes(), (cfg, ctrl, snr, k)
            for k in ("iterations", "crc", "all_zeros", "snr_db"):
                assert a[k] == b[k], (cfg, ctrl, snr, k)


@pytest.mark.parametrize("cfg,cut", [(100, 1000), (101, 900), (102, 1500)])
def test_test_puncture_nbits_hook_identical(cfg, cut):
    """cl_telecom_system::test_puncture_nBits (telecom_system.cc:1186-1192): the oracle's hook against the same zeroing done on the
    compiled reference's demodulated LLRs (oracle/ref_harness.cc), down to the decoder's verdict."""
    orc, ref = oraclelib.Oracle(cfg, 50), oraclelib.RefLib(cfg, 50)
    orc.set_test_puncture(cut)
    ref.set_test_puncture(cut)
    for i, snr in enumerate((OPERATING_ESN0[cfg] + 2.0, OPERATING_ESN0[cfg] - 1.0)):
        bb, _ = orc.gen_frame(78, 10 * cfg + i, oraclelib.noise_amp_for(snr))
        a, b = orc.rx(bb), ref.rx(bb)
        assert not a["llr_demod"][cut:].any() and a["llr_demod"][:cut].any()
        for k in ("llr_demod", "llr_ldpc", "bits", "bytes"):
            assert a[k].tobytes() == b[k].tobytes(), (cfg, snr, k)
        for k in ("iterations", "crc", "all_zeros"):
            assert a[k] == b[k], (cfg, snr, k)


from conftest import EXPLICIT_COMBOS  # noqa: E402


@pytest.mark.parametrize("M,rate16,pre,est,esn0", EXPLICIT_COMBOS)
def test_explicit_configurations_identical(M, rate16, pre, est, esn0):
    """MGPU_CFG_EXPLICIT ids (explicit M / rate / preamble / estimator instead of a CONFIG row): the reference's
    classes configured with the combination (oracle/ref_harness.cc) against the C restatement, stage by stage."""
    from mercury_amd.physical_layer import cfg_explicit
    cfg = cfg_explicit(M, rate16, pre, est)
    assert cfg >= 1000
    orc, ref = oraclelib.Oracle(cfg, 50), oraclelib.RefLib(cfg, 50)
    for n in oraclelib.INFO_FIELDS:
        if n != "dwidth":
            assert getattr(orc, n) == getattr(ref, n), n
    assert (orc.M, orc.K, orc.preamble_nsymb, orc.estimator) == (M, 100 * rate16, pre, est)
    for i, snr in enumerate((esn0, esn0 - 2.5, 40.0)):
        bb, pl = orc.gen_frame(993, 11 * i, oraclelib.noise_amp_for(snr), channel=i % 2)
        bits = orc.payload_to_bits(pl)
        assert orc.tx(bits, 1).tobytes() == ref.tx(bits, 1).tobytes()
        for flags in ((oraclelib.FLAGS_BASEBAND_TEST,) if est == 0 else (oraclelib.FLAGS_BASEBAND_TEST, oraclelib.FLAGS_RECEIVE_BYTE)):
            a, b = orc.rx(bb, flags), ref.rx(bb, flags)
            for k in a:
                if isinstance(a[k], np.ndarray):
                    assert a[k].tobytes() == b[k].tobytes(), (cfg, snr, flags, k)
                elif k != "agc_gain":
                    assert a[k] == b[k] or (np.isnan(a[k]) and np.isnan(b[k])), (cfg, snr, flags, k)
        if i == 0:
            assert a["iterations"] <= 50 and np.array_equal(a["bytes"][: orc.payload_bytes], pl)      # and it decodes


def test_explicit_configuration_ids():
    from mercury_amd.physical_layer import cfg_explicit
    seen = set()
    for M in (2, 4, 8, 16, 32):
        for r in (1, 2, 3, 4, 5, 6, 8, 14):
            for p in range(1, 9):
                for e in (0, 1):
                    c = cfg_explicit(M, r, p, e)
                    assert 1000 <= c < 1640 and c not in seen
                    seen.add(c)
    assert cfg_explicit(64, 8, 1, 0) == -1 and cfg_explicit(4, 7, 1, 0) == -1 and cfg_explicit(4, 8, 0, 0) == -1 and cfg_explicit(4, 8, 9, 1) == -1
    assert oraclelib.Oracle(cfg_explicit(4, 6, 4, 1), 50).K == oraclelib.Oracle(8, 50).K          # cfg 8 is (QPSK, 6/16, 4, LS)
