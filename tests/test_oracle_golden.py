"""CPU tests: the plain-C oracle against the golden vectors generated from the compiled reference
(tests/golden/make_golden.py). Everything stored is required bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

import oraclelib
from conftest import SEED

HERE = os.path.dirname(os.path.abspath(__file__))
META = json.load(open(os.path.join(HERE, "golden", "golden_rx.json")))
ARR = np.load(os.path.join(HERE, "golden", "golden_rx.npz"))


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_known_answers():
    orc = oraclelib.Oracle(0)
    assert list(orc.prng(1, 3)) == [1804289383, 846930886, 1681692777] == META["kat"]["prng_seed1_first3"]
    assert [int(x) for x in orc.prng(0, 8)] == META["kat"]["prng_seed0_first8"]
    assert orc.crc16([ord(c) for c in "123456789"]) == 0x4B37 == META["kat"]["crc16_123456789"]
    scr = orc.scrambler()
    assert "".join(str(int(b)) for b in scr[:32]) == "10111100110101100000101100011110"   # SURVEY.md §8c anchor
    assert np.array_equal(scr.astype(np.uint8), ARR["scrambler"])


@pytest.mark.parametrize("cfg", list(range(17)))
def test_tables_and_mode_parameters(cfg):
    orc = oraclelib.Oracle(cfg)
    m = META["modes"][str(cfg)]
    for k, v in m.items():
        if k != "frames":
            assert getattr(orc, k) == v, (cfg, k)
    types = orc.frame_types()
    assert np.array_equal(types.astype(np.uint8), ARR["cfg%d_frame_types" % cfg])
    rows, cols = np.divmod(np.arange(types.size), orc.Nc)
    assert np.array_equal(types == 1, (rows - cols) % 3 == 0)          # lattice rule, SURVEY.md §8c
    assert orc.pilot_seq().tobytes() == ARR["cfg%d_pilot_seq" % cfg].tobytes()
    assert orc.constellation().tobytes() == ARR["cfg%d_constellation" % cfg].tobytes()
    assert abs(abs(orc.pilot_seq()[0]) - 1.3300000429153442) == 0.0


def test_mode_table_equals_what_the_compiled_reference_printed():
    """SURVEY.md §8 row a22, pinned independently of oracle/ref_harness.cc (which restates the 17 mode rows because telecom_system.cc
    cannot be linked here): the oracle AND the library's host-side table builder (mgpu_host_mode_info: the code mgpu_create runs) report
    the numbers the surveyor printed from the compiled reference after the real load_configuration(cfg) — SURVEY.md §0, transcribed by
    tests/golden/make_survey_mode_table.py."""
    import ctypes as C
    from mercury_amd import load_library
    from mercury_amd.physical_layer import Info
    tab = json.load(open(os.path.join(HERE, "golden", "survey_mode_table.json")))
    assert sorted(tab["modes"], key=int) == [str(c) for c in range(17)]
    lib = load_library()
    for cfg in range(17):
        want = dict(tab["modes"][str(cfg)], **tab["fixed"])
        orc = oraclelib.Oracle(cfg)
        info = Info()
        assert lib.mgpu_host_mode_info(C.c_int(cfg), C.c_int(0), C.byref(info)) == 0
        for k, v in want.items():
            if k != "E":                                   # the oracle's info carries the table widths, not the edge count
                assert getattr(orc, k) == v, ("oracle", cfg, k, getattr(orc, k), v)
            assert getattr(info, k) == v, ("library", cfg, k, getattr(info, k), v)
        assert info.payload_bytes == (want["nBits"] - want["P"] - 16) // 8 == orc.payload_bytes      # telecom_system.cc:332-340
        assert info.frame_samples == want["Nsymb"] * want["Nofdm"]
    bad = Info()
    assert lib.mgpu_host_mode_info(C.c_int(55), C.c_int(0), C.byref(bad)) != 0


@pytest.mark.parametrize("cfg", list(range(17)))
def test_rx_chain_matches_reference_vectors(cfg):
    orc = oraclelib.Oracle(cfg, 50)
    for idx, rec in enumerate(META["modes"][str(cfg)]["frames"]):
        bb, pl = orc.gen_frame(SEED, rec["frame"], oraclelib.noise_amp_for(rec["esn0_db"]), rec["channel"])
        assert digest(bb) == rec["input_sha256"], "generator drift (libm?)"
        assert digest(pl.astype(np.uint8)) == rec["payload_sha256"]
        for vname, flags in (("baseband_test", oraclelib.FLAGS_BASEBAND_TEST), ("receive_byte", oraclelib.FLAGS_RECEIVE_BYTE)):
            g = rec["variants"][vname]
            r = orc.rx(bb, flags)
            key = "cfg%d_f%d_%s" % (cfg, idx, vname)
            for k, d in g["sha256"].items():
                assert digest(r[k]) == d, (cfg, idx, vname, k)
            assert r["llr_ldpc"].tobytes() == ARR[key + "_llr_ldpc"].tobytes()
            assert np.array_equal(np.packbits(r["bits"].astype(np.uint8)), ARR[key + "_bits"])
            assert np.array_equal(r["bytes"].astype(np.uint8), ARR[key + "_bytes"])
            assert (r["iterations"], r["crc"], r["all_zeros"]) == (g["iterations"], g["crc"], g["all_zeros"])
            assert float(r["variance"]).hex() == g["variance"] and float(r["variance_f"]).hex() == g["variance_f"]
            # the reference harness recovers the AGC gain by probing one cell (after/before), so allow its rounding
            assert abs(r["agc_gain"] - float.fromhex(g["agc_gain"])) <= 1e-15 * abs(r["agc_gain"])
            if rec["esn0_db"] == 60.0:   # noiseless: payload must come back and the CRC self-check is 0
                assert r["iterations"] == 0 and r["crc"] == 0
                assert np.array_equal(r["bytes"][: orc.payload_bytes], pl)


def test_tx_rx_round_trip_and_crc_property():
    """encode -> decode returns the input; CRC16([payload | crc_lo | crc_hi]) == 0 (telecom_system.cc:1334-1341)."""
    for cfg in (0, 8, 10, 16):
        orc = oraclelib.Oracle(cfg)
        rng = np.random.default_rng(cfg)
        payload = rng.integers(0, 256, orc.payload_bytes).astype(np.int32)
        bits = orc.payload_to_bits(payload)
        frame = orc.tx(bits, 1)
        r = orc.rx(frame, oraclelib.FLAGS_BASEBAND_TEST)
        assert r["iterations"] == 0 and r["crc"] == 0 and r["all_zeros"] == 0
        assert np.array_equal(r["bytes"][: orc.payload_bytes], payload)
        assert orc.crc16(r["bytes"][: orc.nReal // 8]) == 0


def test_interleaver_tail_passthrough_modes():
    """8PSK (1599 bits, block 159 -> 9-bit tail) and 32QAM exercise the tail rule of interleaver.cc:88-91."""
    for cfg in (10, 14, 16):
        orc = oraclelib.Oracle(cfg)
        assert orc.nBits % 10 != 0 or orc.nData % 10 != 0 or cfg == 16
        payload = np.arange(orc.payload_bytes, dtype=np.int32) % 251
        r = orc.rx(orc.tx(orc.payload_to_bits(payload), 1), oraclelib.FLAGS_BASEBAND_TEST)
        assert np.array_equal(r["bytes"][: orc.payload_bytes], payload)


# ---- MFSK modes (ROBUST_0..2 = cfg 100..102): fixtures from tests/golden/make_golden.py --mfsk -----------------
META_MFSK = json.load(open(os.path.join(HERE, "golden", "golden_mfsk.json")))
ARR_MFSK = np.load(os.path.join(HERE, "golden", "golden_mfsk.npz"))


@pytest.mark.parametrize("cfg", [100, 101, 102])
def test_mfsk_rx_chain_matches_reference_vectors(cfg):
    orc = oraclelib.Oracle(cfg, 50)
    m = META_MFSK["modes"][str(cfg)]
    for k, v in m.items():
        if k != "frames":
            assert getattr(orc, k) == v, (cfg, k)
    assert (orc.M, orc.nPilots, orc.nVirtual, orc.nBits) == (200, 0, 0, 1600)      # MOD_MFSK, no pilots, no shortening
    for idx, rec in enumerate(m["frames"]):
        orc.set_ctrl_mode(rec["ctrl_mode"])
        assert (orc.active_nsymb, orc.active_nbits) == (rec["active_nsymb"], rec["active_nbits"])
        bb, pl = orc.gen_frame(SEED, rec["frame"], oraclelib.noise_amp_for(rec["esn0_db"]), rec["channel"])
        assert bb.size == rec["active_nsymb"] * orc.Nofdm
        assert digest(bb) == rec["input_sha256"], "generator drift (libm?)"
        assert digest(pl.astype(np.uint8)) == rec["payload_sha256"]
        r = orc.rx(bb)
        n = orc.active_nsymb * orc.Nc
        assert digest(r["grid"][:n]) == rec["sha256"]["grid"]
        assert digest(r["llr_demod"]) == rec["sha256"]["llr_demod"]
        key = "cfg%d_f%d" % (cfg, idx)
        assert r["llr_ldpc"].tobytes() == ARR_MFSK[key + "_llr_ldpc"].tobytes()
        assert np.array_equal(np.packbits(r["bits"].astype(np.uint8)), ARR_MFSK[key + "_bits"])
        assert np.array_equal(r["bytes"].astype(np.uint8), ARR_MFSK[key + "_bytes"])
        assert (r["iterations"], r["crc"], r["all_zeros"], r["snr_db"]) == (rec["iterations"], rec["crc"], rec["all_zeros"], rec["snr_db"])
        assert np.abs(r["llr_ldpc"]).max() <= 5.0                                   # mfsk.cc:382-384 clamp
        if rec["ctrl_mode"]:
            assert np.count_nonzero(r["llr_demod"][orc.active_nbits:]) == 0         # punctured tail, telecom_system.cc:1183-1191
        if rec["esn0_db"] == 60.0:
            assert r["crc"] == 0 and r["snr_db"] == 0.0 and np.array_equal(r["bytes"][: orc.payload_bytes], pl)
    orc.set_ctrl_mode(0)


# ---- synchroniser blocks: fixtures from tests/golden/make_golden.py --sync (outputs of the compiled reference) ---------
@pytest.mark.parametrize("cfg", [8, 10, 16, 100, 101])
def test_sync_blocks_match_reference_vectors(cfg):
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(HERE, "golden", "golden_sync.json")))[str(cfg)]
    got = json.loads(json.dumps(mg.sync_case(oraclelib.Oracle(cfg), cfg)))          # same code path, the oracle as `lib`
    assert got == want


# ---- explicit parameter sets (physical_config.cc:35-65 varied): fixtures from tests/golden/make_golden.py --explicit -------------
def test_explicit_parameter_sets_match_reference_vectors():
    """morc_create_explicit against the compiled reference configured the same way (pilot boost, LS window, pilot / scrambler /
    preamble seeds): tables, every RX stage in both variants, a transmitted frame, the pre-equalization channel."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(HERE, "golden", "golden_explicit.json")))
    assert len(want) == len(mg.EXPLICIT_CASES)
    for i, (cfg, x) in enumerate(mg.EXPLICIT_CASES):
        got = json.loads(json.dumps(mg.explicit_case(oraclelib.Oracle, cfg, x, i), sort_keys=True))
        assert got == want[i], (cfg, [k for k in got if got[k] != want[i].get(k)])


@pytest.mark.skipif(not oraclelib.RefLib.available(), reason="needs oracle/_ref (the compiled reference: build container only)")
def test_committed_fixtures_are_what_the_generator_writes_today(tmp_path):
    """VERDICT r05 housekeeping: tests/golden/make_golden.py, run against the compiled reference, must reproduce every committed fixture - the
    JSON documents key for key, the .npz archives array for array (dtype and bits) - so that the script and the files cannot drift apart
    (round 5's golden_rx.json lacked four info keys the script had learned to write). Runs where oracle/_ref exists; ~20 s."""
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    env = dict(os.environ, MERCURY_GOLDEN_OUT=str(tmp_path))
    for flag in ([], ["--mfsk"], ["--tx"], ["--sync"], ["--explicit"]):
        r = subprocess.run([sys.executable, os.path.join(here, "make_golden.py")] + flag, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    made = sorted(os.listdir(tmp_path))
    assert made == ["golden_explicit.json", "golden_mfsk.json", "golden_mfsk.npz", "golden_rx.json", "golden_rx.npz", "golden_sync.json", "golden_tx.json"]
    for name in made:
        new, old = os.path.join(tmp_path, name), os.path.join(here, name)
        if name.endswith(".json"):
            assert json.load(open(new)) == json.load(open(old)), name
        else:
            a, b = np.load(new), np.load(old)
            assert sorted(a.files) == sorted(b.files), name
            for k in a.files:
                assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), (name, k)
