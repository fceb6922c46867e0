#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container: it needs oracle/_ref/libmercury_ref.so, i.e. the reference's own
DSP objects compiled from /root/reference by oracle/Makefile. For every Mercury mode 0..16 a few
frames are synthesised (the repo's Philox generator: seed, frame index, Es/N0, channel), pushed
through the reference RX chain in both orchestration variants (baseband_test_EsN0,
telecom_system.cc:155-198, and receive_byte, :1132-1345), and the reference's outputs are stored:

  * small outputs verbatim: decoder-input LLRs (float32[1600]), hard bits, de-scrambled bytes,
    iteration count, CRC, all-zeros flag, variance (double + float), AGC gain;
  * large FP64 intermediates (carrier grid, channel estimate, equalised grid, de-interleaved
    symbols, demapper LLRs) as SHA-256 digests — the CPU oracle is required to match them bit for bit;
  * static tables per mode (pilot lattice + signs, scrambler, constellation) and KATs (PRNG, CRC).

`--mfsk` writes golden_mfsk.{npz,json} for the three MFSK modes (cfg 100..102 = ROBUST_0..2): the
M == MOD_MFSK branch of receive_byte (telecom_system.cc:1132-1192), full and short control frames.

`--sync` writes golden_sync.json: every synchroniser block (passband_to_baseband, Schmidl-Cox coarse / fine with the k-th
peak, Moose, time_sync_mfsk, the ACK / BREAK detector) of the compiled reference on one capture window per mode.

`--tx` writes golden_tx.json: cl_telecom_system::transmit_byte composed from the compiled reference's objects
(oracle/ref_harness.cc:mref_transmit_byte) for a seeded message per mode — filtered (SINGLE_MESSAGE) and unfiltered
(NO_FILTER_MESSAGE), two carrier phase origins, a short message, MFSK control frames, the ARQ sender's batch form (frames
filtered as one concatenation) and the ACK / BREAK tone patterns.

Fixtures are DATA (inputs are regenerated from the recorded seeds; a digest of the input guards
against generator drift). No reference source text is stored.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.environ.get("MERCURY_GOLDEN_OUT", HERE)     # tests/test_oracle_golden.py regenerates into a scratch directory and compares with the committed files
sys.path.insert(0, os.path.dirname(HERE))
import oraclelib  # noqa: E402
from conftest import OPERATING_ESN0, SEED  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert oraclelib.RefLib.available(), "build oracle/_ref first (make -C oracle ref)"
    arrays = {}
    meta = {"seed": SEED, "modes": {}, "kat": {}}
    ref0 = oraclelib.RefLib(0)
    meta["kat"]["prng_seed1_first3"] = [int(x) for x in ref0.prng(1, 3)]
    meta["kat"]["prng_seed0_first8"] = [int(x) for x in ref0.prng(0, 8)]
    meta["kat"]["crc16_123456789"] = ref0.crc16([ord(c) for c in "123456789"])
    for cfg in range(17):
        ref = oraclelib.RefLib(cfg, 50)
        orc = oraclelib.Oracle(cfg, 50)   # only used as the input generator here
        m = {n: getattr(ref, n) for n in oraclelib.INFO_FIELDS if n != "dwidth"}
        arrays["cfg%d_frame_types" % cfg] = ref.frame_types().astype(np.uint8)
        arrays["cfg%d_pilot_seq" % cfg] = ref.pilot_seq()
        arrays["cfg%d_constellation" % cfg] = ref.constellation()
        if cfg == 0:
            arrays["scrambler"] = ref.scrambler().astype(np.uint8)
        op = OPERATING_ESN0[cfg]
        cases = [(op, 0), (op + 1.0, 0), (-15.0, 0), (60.0, 0)]
        if cfg in (8, 16):
            cases.append((30.0, 1))       # static 2-path channel
        frames = []
        for idx, (snr, ch) in enumerate(cases):
            bb, pl = orc.gen_frame(SEED, 100 * cfg + idx, oraclelib.noise_amp_for(snr), ch)
            rec = {"frame": 100 * cfg + idx, "esn0_db": snr, "channel": ch, "input_sha256": digest(bb),
                   "payload_sha256": digest(pl.astype(np.uint8)), "variants": {}}
            for vname, flags in (("baseband_test", oraclelib.FLAGS_BASEBAND_TEST), ("receive_byte", oraclelib.FLAGS_RECEIVE_BYTE)):
                r = ref.rx(bb, flags)
                key = "cfg%d_f%d_%s" % (cfg, idx, vname)
                arrays[key + "_llr_ldpc"] = r["llr_ldpc"]
                arrays[key + "_bits"] = np.packbits(r["bits"].astype(np.uint8))
                arrays[key + "_bytes"] = r["bytes"].astype(np.uint8)
                rec["variants"][vname] = {
                    "iterations": int(r["iterations"]), "crc": int(r["crc"]), "all_zeros": int(r["all_zeros"]),
                    "variance": float(r["variance"]).hex(), "variance_f": float(r["variance_f"]).hex(),
                    "agc_gain": float(r["agc_gain"]).hex(), "mean_H": float(r["mean_H"]).hex(),
                    "sha256": {k: digest(r[k]) for k in ("grid", "H", "eq", "syms", "llr_demod", "llr_ldpc")},
                }
            frames.append(rec)
        m["frames"] = frames
        meta["modes"][str(cfg)] = m
        print("cfg", cfg, "done")
    np.savez_compressed(os.path.join(OUT, "golden_rx.npz"), **arrays)
    with open(os.path.join(OUT, "golden_rx.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote golden_rx.npz (%d arrays), golden_rx.json" % len(arrays))


def main_mfsk():
    """ROBUST_0..2 (cfg 100..102): the MFSK branch of receive_byte, full frames and short control frames."""
    assert oraclelib.RefLib.available(), "build oracle/_ref first (make -C oracle ref)"
    arrays = {}
    meta = {"seed": SEED, "modes": {}}
    for cfg in (100, 101, 102):
        ref = oraclelib.RefLib(cfg, 50)
        orc = oraclelib.Oracle(cfg, 50)   # input generator only
        m = {n: getattr(ref, n) for n in oraclelib.INFO_FIELDS if n != "dwidth"}
        op = OPERATING_ESN0[cfg]
        cases = [(op, 0, 0), (op + 1.0, 0, 0), (-20.0, 0, 0), (60.0, 0, 0)]
        if ref.mfsk_M and cfg != 102:
            cases += [(op + 1.0, 0, 1), (60.0, 0, 1)]      # short control frames (set_mfsk_ctrl_mode)
        frames = []
        for idx, (snr, ch, ctrl) in enumerate(cases):
            ref.set_ctrl_mode(ctrl)
            orc.set_ctrl_mode(ctrl)
            bb, pl = orc.gen_frame(SEED, 100 * cfg + idx, oraclelib.noise_amp_for(snr), ch)
            r = ref.rx(bb, oraclelib.FLAGS_RECEIVE_BYTE)
            key = "cfg%d_f%d" % (cfg, idx)
            arrays[key + "_llr_ldpc"] = r["llr_ldpc"]
            arrays[key + "_bits"] = np.packbits(r["bits"].astype(np.uint8))
            arrays[key + "_bytes"] = r["bytes"].astype(np.uint8)
            n = ref.active_nsymb * ref.Nc
            frames.append({"frame": 100 * cfg + idx, "esn0_db": snr, "channel": ch, "ctrl_mode": ctrl,
                           "active_nsymb": ref.active_nsymb, "active_nbits": ref.active_nbits,
                           "input_sha256": digest(bb), "payload_sha256": digest(pl.astype(np.uint8)),
                           "iterations": int(r["iterations"]), "crc": int(r["crc"]), "all_zeros": int(r["all_zeros"]),
                           "snr_db": float(r["snr_db"]),
                           "sha256": {"grid": digest(r["grid"][:n]), "llr_demod": digest(r["llr_demod"]),
                                      "llr_ldpc": digest(r["llr_ldpc"])}})
        ref.set_ctrl_mode(0)
        m["frames"] = frames
        meta["modes"][str(cfg)] = m
        print("cfg", cfg, "done")
    np.savez_compressed(os.path.join(OUT, "golden_mfsk.npz"), **arrays)
    with open(os.path.join(OUT, "golden_mfsk.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote golden_mfsk.npz (%d arrays), golden_mfsk.json" % len(arrays))


def sync_case(lib, cfg):
    """One capture window through every synchroniser block of `lib` (RefLib when generating, Oracle when checking)."""
    from oraclelib import CARRIER
    rng = np.random.default_rng(7000 + cfg)
    payload = rng.integers(0, 256, lib.payload_bytes)
    pb = lib.tx_passband(lib.payload_to_bits(payload))
    sym = lib.Nofdm * 4
    n = 60 * sym if cfg < 100 else pb.size + 40 * sym
    delay = 9 * sym + 321
    win = rng.standard_normal(n) * 0.02
    win[delay: delay + pb.size] += pb
    rec = {"passband_sha256": digest(pb), "window_sha256": digest(win), "fir_taps": [digest(lib.fir_taps(0)), digest(lib.fir_taps(1))]}
    bbi = lib.passband_to_baseband(win, CARRIER, 1, 0)
    rec["p2b_time_sync_sha256"] = digest(bbi)
    rec["p2b_data_sha256"] = digest(lib.passband_to_baseband(win, CARRIER + 3.0, 1, 1))
    rec["p2b_data_decim4_sha256"] = digest(lib.passband_to_baseband(win[777:], CARRIER, 4, 1))
    if cfg >= 100:
        rec["mfsk_delay"] = [lib.time_sync_mfsk(bbi), lib.time_sync_mfsk(bbi, 12)]
        rec["mfsk_preamble_sha256"] = digest(lib.mfsk_pattern(0))
    else:
        d, c = lib.time_sync_preamble(bbi, 100)
        rec["coarse"] = [d, float(c).hex()]
        ps = max(1, d // sym)
        seg = bbi[(ps - 1) * sym: (ps - 1) * sym + (lib.preamble_nsymb + 4) * sym]
        rec["fine"] = [[dd, float(cc).hex()] for dd, cc in (lib.time_sync_preamble(seg, 1, loc, 2) for loc in (0, 1, 2))]
        fine = (ps - 1) * sym + rec["fine"][0][0]
        bb = lib.passband_to_baseband(win, CARRIER, 1, 1)[fine::4]
        rec["moose_hz"] = float(lib.freq_sync(bb[16:])).hex()
        rec["preamble_sha256"] = digest(lib.preamble())
    for which in (1, 2):
        pat = np.repeat(lib.mfsk_pattern(which) / 16.0, 4)
        buf = rng.standard_normal(40 * sym) * 0.5 + 1j * rng.standard_normal(40 * sym) * 0.5
        buf[7 * sym: 7 * sym + pat.size] += pat
        m, k = lib.detect_ack_pattern(buf, which)
        rec["ack_%d" % which] = [float(m).hex(), k, digest(lib.mfsk_pattern(which))]
    return rec


TX_CFGS = (0, 5, 8, 11, 14, 16, 100, 101, 102)


def tx_case(lib, cfg):
    """transmit_byte outputs of `lib` (the compiled reference when generating, the oracle when checking) as digests plus
    a few samples in hex; the message bytes are regenerated from the seed."""
    rng = np.random.default_rng(7000 + cfg)
    nb = (lib.nReal - 16) // 8
    msg = rng.integers(0, 256, nb).astype(np.int32)
    rec = {"message": digest(msg)}
    cases = [("filtered", msg, dict(message_location=oraclelib.SINGLE_MESSAGE)),
             ("unfiltered", msg, dict(message_location=oraclelib.NO_FILTER_MESSAGE)),
             ("late_phase", msg, dict(message_location=oraclelib.SINGLE_MESSAGE, start_sample=3 * 10 ** 9 + 17)),
             ("short", msg[: nb // 3], dict(message_location=oraclelib.NO_FILTER_MESSAGE, carrier=1650.0, output_power_watt=0.25,
                                            preamble_papr_cut=5.0, data_papr_cut=6.0))]
    for name, m, kw in cases:
        y = lib.transmit_byte(m, **kw)
        rec[name] = [digest(y), [float(v).hex() for v in y[[0, 1, 777, y.size // 2, y.size - 1]]]]
    if cfg < 100:
        # pre_equalization_channel as init() computes it (telecom_system.cc:1954-1958, :3108-3145) for two carriers, and the audio
        # transmit_bit makes with it (:474-494)
        for tag, fc in (("pre_eq", oraclelib.CARRIER), ("pre_eq_1650", 1650.0)):
            h = lib.get_pre_equalization_channel(fc)
            rec[tag] = [digest(h), [float(v).hex() for v in h.view(np.float64)[[0, 1, 50, 51, 98, 99]]]]
        y = lib.transmit_byte(msg, message_location=oraclelib.SINGLE_MESSAGE, pre_equalize=True)
        rec["filtered_pre_eq"] = [digest(y), [float(v).hex() for v in y[[0, 1, 777, y.size // 2, y.size - 1]]]]
        y = lib.transmit_byte(msg[: nb // 3], message_location=oraclelib.NO_FILTER_MESSAGE, carrier=1650.0, start_sample=99, pre_equalize=True)
        rec["unfiltered_pre_eq_1650"] = [digest(y), [float(v).hex() for v in y[[0, 1, 777, y.size // 2, y.size - 1]]]]
    if cfg >= 100:
        lib.set_ctrl_mode(1)
        y = lib.transmit_byte(msg, message_location=oraclelib.SINGLE_MESSAGE)
        rec["ctrl"] = [digest(y), int(np.count_nonzero(y))]
        lib.set_ctrl_mode(0)
    # the signal path of cl_arq_controller::send_batch (arq_common.cc:2224-2248): three messages, one of them short
    pl3 = rng.integers(0, 256, (3, nb)).astype(np.int32)
    rec["send_batch"] = digest(lib.transmit_batch(pl3, np.array([nb, 2, nb // 2], np.int32), start_sample=31337))
    # FIRST / MIDDLE / FLUSH_MESSAGE overlap-save filtering (telecom_system.cc:559-590): four calls starting with FIRST_MESSAGE on a
    # fresh buffer, then two more as MIDDLE_MESSAGE and one FLUSH_MESSAGE on the buffer they left (one message short)
    pl7 = rng.integers(0, 256, (7, nb)).astype(np.int32)
    used = (lib.preamble_nsymb + lib.active_nsymb) * lib.Nofdm * 4
    y1, buf = lib.transmit_stream(pl7[:4], oraclelib.FIRST_MESSAGE, start_sample=4242)
    y2, buf = lib.transmit_stream(pl7[4:6], oraclelib.MIDDLE_MESSAGE, buffer=buf, nbytes=np.array([nb, 3], np.int32), start_sample=4242 + 4 * used)
    y3, buf = lib.transmit_stream(pl7[6:], oraclelib.FLUSH_MESSAGE, buffer=buf, start_sample=4242 + 6 * used)
    rec["stream"] = [digest(y1), digest(y2), digest(y3), digest(buf), [float(v).hex() for v in y1[1, [0, 1, 777, y1.shape[1] // 2, y1.shape[1] - 1]]]]
    # generate_ack_pattern_passband / generate_break_pattern_passband (telecom_system.cc:1589-1689)
    rec["ack"] = digest(lib.generate_ack_pattern_passband(1))
    rec["break"] = digest(lib.generate_ack_pattern_passband(2, start_sample=10 ** 7 + 1, output_power_watt=0.05, data_papr_cut=3.0))
    assert lib.transmit_byte(np.zeros(nb + 1, np.int32)) is None       # "message too long.. not sent."
    return rec


def main_tx():
    assert oraclelib.RefLib.available(), "build oracle/_ref first (make -C oracle ref)"
    meta = {str(cfg): tx_case(oraclelib.RefLib(cfg), cfg) for cfg in TX_CFGS}
    with open(os.path.join(OUT, "golden_tx.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote golden_tx.json")


def main_sync():
    assert oraclelib.RefLib.available(), "build oracle/_ref first (make -C oracle ref)"
    meta = {str(cfg): sync_case(oraclelib.RefLib(cfg), cfg) for cfg in (8, 10, 16, 100, 101)}
    with open(os.path.join(OUT, "golden_sync.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote golden_sync.json")


# Explicit parameter sets (mgpu_create_explicit / morc_create_explicit): values physical_config.cc:35-65 holds for every mode, varied.
EXPLICIT_CASES = [
    # (cfg, parameters)
    (8, dict(pilot_boost=1.7, ls_window=9, pilot_seed=5, scrambler_seed=7, preamble_seed=3)),
    (0, dict(pilot_boost=1.0, ls_window=4, pilot_seed=11, scrambler_seed=0, preamble_seed=1)),
    (13, dict(pilot_boost=1.33, ls_window=14, pilot_seed=0, scrambler_seed=123456, preamble_seed=99)),
    (16, dict(pilot_boost=2.5, ls_window=20, pilot_seed=77, scrambler_seed=1, preamble_seed=2)),      # ZF estimator: the window is unused
    (4, dict(pilot_boost=0.8, ls_window=1, pilot_seed=3, scrambler_seed=9, preamble_seed=4)),          # a window of one cell
    # frame geometry (round 6): ofdm_Nsymb / ofdm_pilot_configurator_Dy as load_configuration copies them (telecom_system.cc:2775-2778)
    (8, dict(Nsymb=20, Dy=5)),                                                   # the reference's LOW_DENSITY option, QPSK (telecom_system.cc:1828-1836, :1857-1865)
    (0, dict(Nsymb=40, Dy=5, ls_window=9, pilot_seed=2)),                        # LOW_DENSITY, BPSK
    (13, dict(Nsymb=10, Dy=5, pilot_boost=1.5)),                                 # LOW_DENSITY, 16QAM
    (16, dict(Nsymb=8, Dy=4)),                                                   # 32QAM, zero-forcing estimator, 100 virtual bits
    (4, dict(Nsymb=45, Dy=3)),                                                   # the default lattice on a shorter frame: 100 virtual bits
    (8, dict(Nsymb=18, Dy=9, ls_window=21)),                                     # two pilots per column: every data row extrapolates or spans 9 rows
]


def explicit_case(lib, cfg, x, idx):
    """One explicit parameter set through `lib` (RefLib or Oracle): tables, the RX chain's outputs on two frames, one transmitted frame."""
    gen = oraclelib.Oracle(cfg, 50, explicit=x)      # input generator (its frames carry the set's pilots / scrambler)
    o = lib(cfg, 50, explicit=x)
    rec = {"cfg": cfg, "params": x, "ls_window": o.ls_window, "pilot_seq": digest(o.pilot_seq()), "scrambler": digest(o.scrambler().astype(np.uint8)),
           "frames": []}
    op = OPERATING_ESN0[cfg]
    for k, snr in enumerate((op + 1.0, 40.0)):
        bb, pl = gen.gen_frame(SEED, 5000 + 10 * idx + k, oraclelib.noise_amp_for(snr), 0)
        fr = {"frame": 5000 + 10 * idx + k, "esn0_db": snr, "input_sha256": digest(bb), "variants": {}}
        for vname, flags in (("baseband_test", oraclelib.FLAGS_BASEBAND_TEST), ("receive_byte", oraclelib.FLAGS_RECEIVE_BYTE)):
            r = o.rx(bb, flags)
            fr["variants"][vname] = {"iterations": int(r["iterations"]), "crc": int(r["crc"]), "variance_f": float(r["variance_f"]).hex(),
                                     "sha256": {q: digest(r[q]) for q in ("grid", "H", "eq", "syms", "llr_demod", "llr_ldpc", "bytes")}}
        if k == 0:
            fr["tx_sha256"] = digest(o.tx(o.payload_to_bits(pl), 1))
        rec["frames"].append(fr)
    if cfg != 16:
        rec["pre_equalization_channel"] = digest(o.get_pre_equalization_channel(oraclelib.CARRIER))
    return rec


def main_explicit():
    assert oraclelib.RefLib.available(), "build oracle/_ref first (make -C oracle ref)"
    out = [explicit_case(oraclelib.RefLib, cfg, x, i) for i, (cfg, x) in enumerate(EXPLICIT_CASES)]
    with open(os.path.join(OUT, "golden_explicit.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote golden_explicit.json")


if __name__ == "__main__":
    if "--explicit" in sys.argv:
        main_explicit() # explicit parameter sets (boost, LS window, seeds)
    elif "--tx" in sys.argv:
        main_tx()       # transmit_byte (the TX mirror up to the audio samples)
    elif "--sync" in sys.argv:
        main_sync()     # synchroniser blocks and the MFSK sync / ACK detector
    elif "--mfsk" in sys.argv:
        main_mfsk()     # separate fixture files: the OFDM fixtures are not regenerated
    else:
        main()
