"""The multi-GPU pool (include/mercury_pool.h, SURVEY.md §8 row e): frame-range sharding, one context + host thread per device,
no collectives. CPU: the shard arithmetic, the ABI and the loud failure without a GPU. GPU: a pool of several contexts on
device 0 must return exactly what one context returns, for ragged batch sizes, from Python and from a C++14 program."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oraclelib
from conftest import OPERATING_ESN0, SEED

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_every_batch():
    from mercury_amd import pool_shard
    from mercury_amd.sharding import frame_range, owner_of
    for F in (0, 1, 2, 7, 8, 9, 63, 64, 65, 4096, 1000003):
        for G in (1, 2, 3, 4, 8, 16):
            nxt = 0
            for g in range(G):
                a, n = pool_shard(F, G, g)
                assert a == nxt and n >= 0
                assert (a, a + n) == tuple(frame_range(g, G, F))       # the same rule bench.py's ranks use
                nxt = a + n
                for f in {a, a + n - 1} if n else ():
                    assert owner_of(f, G, F) == g
            assert nxt == F


def test_pool_fails_loudly_without_a_gpu():
    import torch
    from mercury_amd import MgpuError, RxPool
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(MgpuError) as e:
        RxPool(8, [0, 0], max_batch=4)
    assert "mgpu_pool_create failed (2)" in str(e.value)


def test_pool_rejects_bad_arguments():
    from mercury_amd import load_library
    from mercury_amd.physical_layer import Config
    lib = load_library()
    h = C.c_void_p()
    cfg = Config(8, 50, 1, 1, 1, 0, 4, 0.0, 0)
    devs = (C.c_int * 2)(0, 0)
    assert lib.mgpu_pool_create(C.byref(cfg), devs, 0, C.byref(h)) == 1
    assert lib.mgpu_pool_create(C.byref(cfg), devs, 17, C.byref(h)) == 1
    assert lib.mgpu_pool_create(None, devs, 2, C.byref(h)) == 1
    assert lib.mgpu_pool_rx_batch(None, None, 0, None, None) == 1
    assert lib.mgpu_pool_size(None) == 0


def _cpp(tmp_path):
    exe = tmp_path / "pool_test"
    lib = os.path.join(ROOT, "mercury_amd")
    subprocess.run(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "pool_test.cpp"),
                    "-o", str(exe), "-L", lib, "-lmercury_gpu", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-pthread"], check=True)
    return exe


def test_cpp_pool_program_compiles(tmp_path):
    assert _cpp(tmp_path).exists()


@pytest.mark.gpu
@pytest.mark.parametrize("F,n_ctx", [(1, 2), (5, 2), (37, 2), (37, 3), (64, 4)])
def test_cpp_pool_equals_single_context(tmp_path, F, n_ctx):
    exe = _cpp(tmp_path)
    cfg = 8
    orc = oraclelib.Oracle(cfg, 50)
    snrs = [OPERATING_ESN0[cfg] + 1.0, -15.0, OPERATING_ESN0[cfg], 60.0]
    bb = np.stack([orc.gen_frame(SEED, 4000 + i, oraclelib.noise_amp_for(snrs[i % 4]))[0] for i in range(F)])
    (tmp_path / "bb.bin").write_bytes(bb.tobytes())
    r = subprocess.run([str(exe), str(cfg), str(F), str(tmp_path / "bb.bin"), str(tmp_path / "out.bin"), str(n_ctx)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    raw = (tmp_path / "out.bin").read_bytes()
    half = len(raw) // 2
    assert raw[:half] == raw[half:], "pool output differs from the single-context output"
    stride = (orc.nReal + 7) // 8
    pay = np.frombuffer(raw[: F * stride], np.uint8).reshape(F, stride)
    for f in (0, F // 2, F - 1):                     # and that output is the oracle's
        ref = orc.rx(bb[f], oraclelib.FLAGS_RECEIVE_BYTE)
        assert np.array_equal(pay[f], ref["bytes"].astype(np.uint8))


@pytest.mark.gpu
def test_python_pool_rx_ldpc_and_receive_byte_equal_single_context():
    from mercury_amd import RxPhy, RxPool
    cfg, F = 5, 23
    orc = oraclelib.Oracle(cfg, 50)
    bb = np.stack([orc.gen_frame(SEED, 5000 + i, oraclelib.noise_amp_for(OPERATING_ESN0[cfg] + (1.0 if i % 3 else -20.0)))[0] for i in range(F)])
    one = RxPhy(cfg, max_batch=F)
    pool = RxPool(cfg, [0, 0, 0], max_batch=F)
    a, b = one.receive(bb, want_llr=True), pool.receive(bb)
    assert np.array_equal(a["payload"], b["payload"]) and a["stats"].tobytes() == b["stats"].tobytes()
    k = pool.counters()
    assert k["frames"] == F and sum(k["device_frames"]) == F and k["decoded"] == int((a["stats"]["message_decoded"] != 0).sum())
    assert k["ldpc_iterations"] == int(np.minimum(a["stats"]["iterations_done"], 50).sum())
    bits1, it1 = one.ldpc_decode(a["llr_ldpc"])
    bits2, it2 = pool.ldpc_decode(a["llr_ldpc"])
    assert np.array_equal(bits1, bits2) and np.array_equal(it1, it2)
    # capture windows through receive_byte: windows are independent, so the split must not show
    from test_receive_byte import SPECS, make_windows
    from oraclelib import CARRIER
    wins, _ = make_windows(orc, SPECS[:7], seed=77)
    r1, r2 = one.receive_byte(wins, CARRIER), pool.receive_byte(wins, CARRIER)
    assert np.array_equal(r1["payload"], r2["payload"]) and r1["stats"].tobytes() == r2["stats"].tobytes()
    assert r1["state"].tobytes() == r2["state"].tobytes()
    assert int((r1["stats"]["message_decoded"] != 0).sum()) >= 3
    one.close()
    pool.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,F,devices", [(8, 37, [0, 0]), (16, 10, [0, 0, 0]), (8, 3, [0, 0, 0, 0])])
def test_pool_device_resident_shards_equal_single_context(cfg, F, devices):
    """mgpu_pool_rx_batch_dev / mgpu_pool_ldpc_batch_dev / mgpu_pool_txgen_dev: every device's shard lies in its
    own memory (allocated through the C-ABI, no torch involved), outputs stay there. A pool of contexts on device 0 must give, shard
    by shard, exactly the bytes one context gives for the whole batch - generated frames, payload, stats, hard bits and iterations -
    and the merged counters must add up."""
    from mercury_amd import RxPhy, RxPool, STATS_DTYPE
    agc, vs = (0, 0) if cfg in (15, 16) else (1, 1)
    one = RxPhy(cfg, max_batch=F, agc=agc, variance_source=vs)
    pool = RxPool(cfg, devices, max_batch=F, agc=agc, variance_source=vs)
    G = len(devices)
    shards = pool.shard(F)
    counts = [n for _, n in shards]
    assert sum(counts) == F
    fs, ps = one.frame_samples, one.payload_stride
    d_bb = [pool.device_malloc(g, max(1, counts[g]) * fs * 16) for g in range(G)]
    d_sent = [pool.device_malloc(g, max(1, counts[g]) * ps) for g in range(G)]
    d_pay = [pool.device_malloc(g, max(1, counts[g]) * ps) for g in range(G)]
    d_st = [pool.device_malloc(g, max(1, counts[g]) * 24) for g in range(G)]
    amp = oraclelib.noise_amp_for(OPERATING_ESN0[cfg] + 0.5)
    pool.txgen_dev(SEED, 9000, counts, amp, d_bb, d_sent)
    pool.receive_dev(d_bb, counts, d_pay, d_st)
    k = pool.counters()
    bb = np.concatenate([pool.copy_to_host(g, np.zeros((counts[g], fs), np.complex128), d_bb[g]) for g in range(G) if counts[g]])
    pay = np.concatenate([pool.copy_to_host(g, np.zeros((counts[g], ps), np.uint8), d_pay[g]) for g in range(G) if counts[g]])
    st = np.concatenate([pool.copy_to_host(g, np.zeros(counts[g], STATS_DTYPE), d_st[g]) for g in range(G) if counts[g]])
    # the single context on the same frames: generated on the device from the same (seed, frame numbers), received from host memory
    orc = oraclelib.Oracle(cfg, 50)
    sent = np.concatenate([pool.copy_to_host(g, np.zeros((counts[g], ps), np.uint8), d_sent[g]) for g in range(G) if counts[g]])
    for f in (0, F // 2, F - 1):                                      # frame numbering runs across the devices (samples: up to libm ulps in the noise)
        ref_bb, ref_pl = orc.gen_frame(SEED, 9000 + f, amp)
        assert np.array_equal(sent[f][: orc.payload_bytes], ref_pl.astype(np.uint8))
        assert np.abs(bb[f] - ref_bb).max() <= 1e-9 * np.abs(ref_bb).max()
    ref = one.receive(bb, want_llr=True)
    assert np.array_equal(pay, ref["payload"]) and st.tobytes() == ref["stats"].tobytes()
    assert k["frames"] == F and k["device_frames"] == counts and k["decoded"] == int((ref["stats"]["message_decoded"] != 0).sum())
    assert k["ldpc_iterations"] == int(np.minimum(ref["stats"]["iterations_done"], 50).sum())
    assert k["decoded"] >= F // 2
    # decoder only, LLRs uploaded shard by shard
    d_llr = [pool.device_malloc(g, max(1, counts[g]) * 1600 * 4) for g in range(G)]
    d_bits = [pool.device_malloc(g, max(1, counts[g]) * one.K) for g in range(G)]
    d_it = [pool.device_malloc(g, max(1, counts[g]) * 4) for g in range(G)]
    for g, (a, n) in enumerate(shards):
        if n:
            pool.copy_to_device(g, d_llr[g], ref["llr_ldpc"][a: a + n])
    pool.ldpc_decode_dev(d_llr, counts, d_bits, d_it)
    bits = np.concatenate([pool.copy_to_host(g, np.zeros((counts[g], one.K), np.uint8), d_bits[g]) for g in range(G) if counts[g]])
    its = np.concatenate([pool.copy_to_host(g, np.zeros(counts[g], np.int32), d_it[g]) for g in range(G) if counts[g]])
    b1, i1 = one.ldpc_decode(ref["llr_ldpc"])
    assert np.array_equal(bits, b1) and np.array_equal(its, i1)
    assert pool.counters()["ldpc_iterations"] == int(np.minimum(i1, 50).sum())
    for g in range(G):
        for p in (d_bb[g], d_sent[g], d_pay[g], d_st[g], d_llr[g], d_bits[g], d_it[g]):
            pool.device_free(g, p)
    one.close()
    pool.close()


@pytest.mark.gpu
def test_one_thread_two_contexts_graph_order_and_device_scope():
    """(high) the single-frame hipGraph must survive a larger batch reallocating the context's input buffer
    (order F=1, F=max, F=1 on a fresh context); (medium) every entry point runs on the context's own device whatever the
    calling thread's current device is."""
    import torch
    from mercury_amd import RxPhy
    cfg, F = 8, 9
    orc = oraclelib.Oracle(cfg, 50)
    bb = np.stack([orc.gen_frame(SEED, 6000 + i, oraclelib.noise_amp_for(OPERATING_ESN0[cfg] + 1.0))[0] for i in range(F)])
    rx = RxPhy(cfg, max_batch=F)
    first = rx.receive(bb[:1])
    allf = rx.receive(bb)
    again = rx.receive(bb[:1])
    last = rx.receive(bb[F - 1:])
    assert np.array_equal(first["payload"], again["payload"]) and first["stats"].tobytes() == again["stats"].tobytes()
    assert np.array_equal(allf["payload"][:1], first["payload"]) and np.array_equal(allf["payload"][F - 1:], last["payload"])
    ref = orc.rx(bb[0], oraclelib.FLAGS_RECEIVE_BYTE)
    assert np.array_equal(first["payload"][0], ref["bytes"].astype(np.uint8))
    if torch.cuda.device_count() > 1:
        torch.cuda.set_device(1)
        other = rx.receive(bb)
        assert np.array_equal(other["payload"], allf["payload"])
        assert torch.cuda.current_device() == 1
        torch.cuda.set_device(0)
    rx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [8, 16, 101])
def test_pipelined_host_entry_point_equals_one_launch(cfg, monkeypatch):
    """mgpu_rx_batch for F > 1 runs the batch in chunks on two streams (copy of chunk i+1 under the kernels of chunk i);
    the result must be byte-identical to the one-launch path, whatever the chunk size (ragged last chunk, ZF modes with
    their per-frame equalised-symbol workspace, MFSK modes)."""
    from mercury_amd import RxPhy
    F = 37
    orc = oraclelib.Oracle(cfg, 50)
    agc, vs = (0, 0) if cfg in (15, 16) else (1, 1)
    snr = OPERATING_ESN0[cfg]
    bb = np.stack([orc.gen_frame(SEED, 7000 + i, oraclelib.noise_amp_for(snr + (1.0 if i % 4 else -25.0)))[0] for i in range(F)])
    rx = RxPhy(cfg, max_batch=F, agc=agc, variance_source=vs)
    monkeypatch.setenv("MERCURY_NO_PIPELINE", "1")
    ref = rx.receive(bb)
    monkeypatch.delenv("MERCURY_NO_PIPELINE")
    for chunk in ("7", "16", "36", "64"):
        monkeypatch.setenv("MERCURY_RX_CHUNK", chunk)
        out = rx.receive(bb)
        assert np.array_equal(out["payload"], ref["payload"]) and out["stats"].tobytes() == ref["stats"].tobytes(), (cfg, chunk)
    monkeypatch.delenv("MERCURY_RX_CHUNK")
    out = rx.receive(bb)
    assert np.array_equal(out["payload"], ref["payload"]) and out["stats"].tobytes() == ref["stats"].tobytes()
    assert int((ref["stats"]["message_decoded"] != 0).sum()) >= F // 2
    rx.close()


@pytest.mark.gpu
def test_device_props_and_numa_placed_staging():
    """SURVEY.md §8 row e, placement: the device's properties come from the HIP runtime (what bench.py prices the vector unit with), its
    NUMA node from sysfs; page-locked memory allocated near the device works like any other input buffer, and a pool reports the same
    node for its contexts' devices (its workers bind themselves there, its contexts' staging is allocated there)."""
    from mercury_amd import RxPhy, RxPool, device_props
    from mercury_amd.physical_layer import pinned_empty
    p = device_props(0)
    assert p["compute_units"] == 256 and p["wavefront_size"] == 64 and p["gcn_arch"].startswith("gfx950")
    assert p["lds_bytes_per_cu"] == 160 * 1024 and p["hbm_bytes"] > 200e9 and 1_000_000 < p["clock_khz"] < 3_000_000
    assert len(p["pci_bus_id"]) >= 7 and p["numa_node"] >= -1
    lib = RxPhy(8, max_batch=1).lib
    lib.mgpu_host_numa_node_of_pci.argtypes = [C.c_char_p]
    assert lib.mgpu_host_numa_node_of_pci(p["pci_bus_id"].encode()) == p["numa_node"]
    orc = oraclelib.Oracle(8, 50)
    frames = [orc.gen_frame(SEED, 900 + f, oraclelib.noise_amp_for(OPERATING_ESN0[8] + 1.0)) for f in range(6)]
    bb = np.stack([f[0] for f in frames])
    near = pinned_empty(bb.shape, np.complex128, device=0)
    near[...] = bb
    rx = RxPhy(8, max_batch=8)
    a, b = rx.receive(bb), rx.receive(near)
    assert np.array_equal(a["payload"], b["payload"]) and a["stats"].tobytes() == b["stats"].tobytes()
    for f in range(6):
        assert np.array_equal(a["payload"][f][: orc.payload_bytes], frames[f][1].astype(np.uint8))
    rx.close()
    pool = RxPool(8, [0, 0], max_batch=8)
    assert pool.numa_nodes() == [p["numa_node"]] * 2
    out = pool.receive(near)
    assert np.array_equal(out["payload"], a["payload"])
    pool.close()
