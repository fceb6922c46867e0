"""CPU tests of the host logic around the kernels: frame sharding (single process and a 2-rank gloo
world), the byte model used by bench.py, and the build recipe."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_ranges_partition_the_index_space():
    from mercury_amd.sharding import frame_range, owner_of
    for total in (0, 1, 7, 4096, 1048576, 12345):
        for world in (1, 2, 3, 8):
            ranges = [frame_range(r, world, total) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            for (a, b), (c, d) in zip(ranges, ranges[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
            for f in (0, total // 3, total - 1):
                if 0 <= f < total:
                    r = owner_of(f, world, total)
                    assert ranges[r][0] <= f < ranges[r][1]
    with pytest.raises(ValueError):
        frame_range(2, 2, 10)


def _rank_main(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from mercury_amd.sharding import frame_range, merge_counters
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frame_range(rank, world, 1001)
    # every rank "decodes" its own frames: the stand-in work is a checksum of the frame indices it owns
    local = {"frames": hi - lo, "iterations": 50 * (hi - lo), "checksum": float(sum(range(lo, hi))), "seconds": 0.5 + rank}
    merged = merge_counters(local, dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, lo, hi, merged))


def test_two_rank_gloo_world_shards_and_merges():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, m0), (r1, lo1, hi1, m1) = out
    assert (lo0, hi1) == (0, 1001) and hi0 == lo1
    assert m0 == m1
    assert m0["frames"] == 1001 and m0["iterations"] == 50 * 1001
    assert m0["checksum"] == float(sum(range(1001)))      # a checksum of checksums: nothing lost, nothing doubled
    assert m0["seconds"] == 1.5                            # MAX over ranks


def test_algorithmic_byte_model_matches_survey():
    sys.path.insert(0, ROOT)
    import types
    import bench
    rx = types.SimpleNamespace(Nsymb=24, Nofdm=272, E=5616, N=1600, payload_stride=75)
    ldpc, total, b_iter = bench.algorithmic_bytes(rx, 50, 1)
    assert b_iter == 96256                                   # SURVEY.md §8d, rate 6/16
    assert abs(total - 4.92e6) < 0.02e6                      # 4.92 MB / frame at 50 iterations
    rx8 = types.SimpleNamespace(Nsymb=24, Nofdm=272, E=6049, N=1600, payload_stride=100)
    assert bench.algorithmic_bytes(rx8, 50, 1)[2] == 103184  # rate 8/16
    assert bench.usable_cores() >= 1


def test_bench_contract_line_is_compact_whatever_the_full_record_holds(capsys):
    """round 5's 20 KB bench line was cut by the driver's stdout tail (BENCH_r05.parsed = null). The contract record is
    now built from the full record by bench.compact_line and is the LAST stdout line: at most 4096 bytes, valid JSON, with every contract
    key, `roofline` and `cpu_baseline` - from canned inputs: round 5's committed 20 KB line, the same with an 8-rank block, and with every
    optional block blown up."""
    sys.path.insert(0, ROOT)
    import argparse
    import copy
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_spa_cfg8.json")))
    assert len(json.dumps(full)) > 16000                     # the record that broke the driver's parse
    contract = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
    big = copy.deepcopy(full)
    big["n_gpus"] = 8
    big["per_rank"] = [dict(full["per_rank"][0], rank=r) for r in range(8)]
    big["waterfall_point"] = copy.deepcopy(full["operating_point"])
    huge = copy.deepcopy(big)
    huge["config"]["workload"] *= 3
    huge["cpu_baseline"]["sample"] *= 8
    huge["extras_per_gpu"]["junk"] = ["x" * 100] * 1000
    for rec in (full, big, huge):
        c = bench.compact_line(rec)
        text = json.dumps(c)
        assert len(text) <= bench.COMPACT_LIMIT == 4096, len(text)
        back = json.loads(text)
        assert all(k in back for k in contract) and back["config"]["workload"]
        assert abs(back["value"] / rec["value"] - 1) < 1e-6 and back["steps"] == rec["steps"] and back["warmup"] == rec["warmup"] and back["n_gpus"] == rec["n_gpus"]
        rf = back["roofline"]
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5 and "traffic" in rf
        assert rf["secondary"]["bound"] == "valu_issue" and set(rf["secondary"]) == {"bound", "frac"}      # the issue-bound reading lives only there
        cb = back["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["gpu_vs_cpu_mismatches"] == 0 and cb["reference_1core_frames_per_s"] > 0
    c = bench.compact_line(big)
    assert len(c["per_device"]) == 8 and c["operating_point"]["kernel_ms"]["ldpc"] > 0 and c["waterfall_point"]["value"] > 0
    # emit(): the compact record is the last line on stdout, the full one comes before it
    bench.emit(full, argparse.Namespace(line="both"))
    out = [l for l in capsys.readouterr().out.splitlines() if l]
    assert len(out) == 2 and json.loads(out[0])["bench_full_record"] and json.loads(out[1]) == bench.compact_line(full) and len(out[1]) <= 4096


def test_table_blob_is_wellformed():
    import struct
    blob = open(os.path.join(ROOT, "mercury_amd", "data", "mercury_ldpc_tables.bin"), "rb").read()
    magic, ver, n = struct.unpack_from("<4sII", blob, 0)
    assert magic == b"MLDP" and ver == 1 and n == 8
    off, seen = 12, {}
    for _ in range(n):
        K, P, N, E, cw, vw = struct.unpack_from("<6I", blob, off)
        off += 24
        cdeg = np.frombuffer(blob, "u1", P, off); off += P
        C = np.frombuffer(blob, "<u2", E, off); off += 2 * E
        vdeg = np.frombuffer(blob, "u1", N, off); off += N
        V = np.frombuffer(blob, "<u2", E, off); off += 2 * E
        assert K + P == N == 1600 and cdeg.sum() == E == vdeg.sum() and cdeg.max() == cw and vdeg.max() == vw
        assert C.max() < N and V.max() < P
        seen[K] = E
    assert off == len(blob)
    # SURVEY.md §0 graph statistics
    assert seen == {100: 3574, 200: 3859, 300: 4439, 400: 4651, 500: 5409, 600: 5616, 800: 6049, 1400: 6604}


# ---- the library's own host logic, through its GPU-free entry points ------------------------------------------------------------
def _reference_selection(cand, step, size, loc, ntrials):
    """ofdm.cc:1926-1964 as written: an array of `size` zeros with the candidate metrics at every `step`-th index, then the
    overwrite-not-swap partial sort."""
    vals = np.zeros(size)
    vals[: len(cand) * step: step][: len(cand)] = cand
    locs = -np.ones(size, int)
    if loc >= ntrials:
        loc = ntrials - 1
    for j in range(ntrials):
        locs[j] = j
        for i in range(j + 1, size):
            if vals[i] > vals[j]:
                vals[j] = vals[i]
                locs[j] = i
    return int(locs[loc]), float(vals[loc])


def test_host_select_peak_matches_the_reference_selection():
    import ctypes as C
    from hypothesis import given, settings, strategies as st
    from mercury_amd import load_library
    lib = load_library()

    @settings(max_examples=300, deadline=None)
    @given(st.integers(1, 12), st.integers(1, 7), st.integers(0, 5), st.integers(1, 4), st.integers(0, 3), st.data())
    def check(ncand, step, extra, ntrials, loc, data):
        size = (ncand - 1) * step + 1 + extra
        ntrials = min(ntrials, size)
        cand = np.array(data.draw(st.lists(st.sampled_from([-1.0, -0.25, 0.0, 0.125, 0.5, 0.5, 0.75, 1.0]), min_size=ncand, max_size=ncand)))
        delay, corr = C.c_int(-7), C.c_double(-7)
        rc = lib.mgpu_host_select_peak(cand.ctypes.data_as(C.c_void_p), C.c_int(ncand), C.c_int(step), C.c_int(size), C.c_int(loc), C.c_int(ntrials),
                                       C.byref(delay), C.byref(corr))
        assert rc == 0
        assert (delay.value, corr.value) == _reference_selection(cand, step, size, loc, ntrials), (cand, step, size, loc, ntrials)

    check()


def test_host_filter_designs_and_preamble_match_the_oracle():
    import ctypes as C
    import oraclelib
    from mercury_amd import load_library
    lib = load_library()
    orc = oraclelib.Oracle(8)
    taps, n = np.zeros(128), C.c_int(0)
    for which in (0, 1):
        assert lib.mgpu_host_fir_taps(C.c_int(which), C.c_double(0.0), taps.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
        assert np.array_equal(taps[: n.value], orc.fir_taps(which))
    f = orc.lib.morc_tx_fir_taps
    f.restype = C.c_int
    for carrier in (oraclelib.CARRIER, 1650.0):
        for which in (0, 1):
            want = np.zeros(128)
            nw = f(C.c_double(carrier), C.c_int(which), want.ctypes.data_as(C.c_void_p))
            assert lib.mgpu_host_fir_taps(C.c_int(2 + which), C.c_double(carrier), taps.ctypes.data_as(C.c_void_p), C.byref(n)) == 0
            assert n.value == nw == 97 and np.array_equal(taps[:97], want[:97])
    for cfg in (0, 8, 13, 16):
        o = oraclelib.Oracle(cfg)
        out, ns = np.zeros((8, 50), np.complex128), C.c_int(0)
        assert lib.mgpu_host_preamble_carriers(C.c_int(cfg), out.ctypes.data_as(C.c_void_p), C.byref(ns)) == 0
        assert ns.value == o.preamble_nsymb and np.array_equal(out[: ns.value].ravel(), o.preamble())
    assert lib.mgpu_host_preamble_carriers(C.c_int(55), out.ctypes.data_as(C.c_void_p), C.byref(ns)) != 0


def test_numa_placement_lookups_without_a_gpu(tmp_path, monkeypatch):
    """SURVEY.md §8 row e, host side: the sysfs lookups behind mgpu_alloc_host_near / the pool's worker placement — NUMA node of a PCI
    address, CPUs of a node — against a fabricated sysfs tree; without a device mgpu_device_props_get reports MGPU_ERR_DEVICE."""
    import ctypes as C
    from mercury_amd import load_library
    from mercury_amd.physical_layer import DeviceProps
    lib = load_library()
    lib.mgpu_host_numa_node_of_pci.argtypes = [C.c_char_p]
    root = tmp_path / "sys"
    (root / "bus/pci/devices/0000:c1:00.0").mkdir(parents=True)
    (root / "bus/pci/devices/0000:c1:00.0/numa_node").write_text("1\n")
    (root / "bus/pci/devices/0000:05:00.0").mkdir(parents=True)
    (root / "bus/pci/devices/0000:05:00.0/numa_node").write_text("-1\n")
    (root / "devices/system/node/node1").mkdir(parents=True)
    (root / "devices/system/node/node1/cpulist").write_text("16-19,48-49,63\n")
    monkeypatch.setenv("MERCURY_SYSFS_ROOT", str(root))
    assert lib.mgpu_host_numa_node_of_pci(b"0000:C1:00.0") == 1          # HIP spells the address in upper case, sysfs in lower case
    assert lib.mgpu_host_numa_node_of_pci(b"0000:05:00.0") == -1         # the platform reports no affinity
    assert lib.mgpu_host_numa_node_of_pci(b"0000:ff:00.0") == -1         # no such device
    cpus = (C.c_int * 16)()
    n = lib.mgpu_host_numa_cpus(C.c_int(1), cpus, C.c_int(16))
    assert n == 7 and list(cpus[:7]) == [16, 17, 18, 19, 48, 49, 63]
    assert lib.mgpu_host_numa_cpus(C.c_int(5), cpus, C.c_int(16)) == 0
    import torch
    if not torch.cuda.is_available():
        p = DeviceProps()
        assert lib.mgpu_device_props_get(C.c_int(0), C.byref(p)) == 2   # MGPU_ERR_DEVICE


def test_bench_dry_run_prints_the_shard_map_without_a_gpu():
    """`bench.py --gpus 8 --dry-run`: the driver's 8-GPU launch, as a shard map - disjoint, contiguous, complete global frame ranges per rank
    and input batch - produced without touching a device."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--frames", "4096", "--dry-run"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["dry_run"] and j["n_gpus"] == 8 and j["collectives_on_the_data_path"] == 0 and j["frames_per_step_total"] == 8 * 4096
    assert [k["rank"] for k in j["ranks"]] == list(range(8)) and len({k["device"] for k in j["ranks"]}) == 8
    for b in range(2):
        at = b * 8 * 4096
        for k in j["ranks"]:
            lo, hi = k["global_frames_by_input_batch"][b]
            assert lo == at and hi - lo == 4096 == k["frames_per_step"]
            at = hi


def test_fp32_decoder_placement_keeps_its_gathers_off_each_others_banks():
    """The bank-aware placement of the fp32 decoders' posteriors and edges (tables.cpp; round 4) by the LDS's own rule — a 32-lane group of a
    4-byte gather costs one cycle plus one per extra address on a bank: a random placement costs about 2.8 (check pass) and 4-7 (variable
    update) cycles per group, the placed tables must stay below 2.0 / 2.2 on every rate, and the rate-14/16 graph's large groups reach 1.0."""
    import ctypes as C
    from mercury_amd import load_library
    lib = load_library()
    out = (C.c_double * 4)()
    for cfg in (0, 1, 2, 3, 4, 8, 13, 16):
        assert lib.mgpu_host_layout_stats(C.c_int(cfg), out) == 0
        assert 1.0 <= out[0] < 2.0 and 1.0 <= out[1] < 2.2, (cfg, list(out))
        assert 0.6 < out[3] <= 1.0
    assert lib.mgpu_host_layout_stats(C.c_int(16), out) == 0 and out[1] == 1.0
