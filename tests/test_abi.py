"""CPU tests of the drop-in boundary: the C-ABI library loads, exports exactly what include/mercury_gpu.h and
include/mercury_shm.h declare, and refuses loudly (error code + message, no crash, no CPU fallback) when it cannot run."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    names = set()
    for header in ("mercury_gpu.h", "mercury_shm.h", "mercury_rxloop.h", "mercury_stages.h", "mercury_tx.h", "mercury_pool.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(mgpu_[a-z_0-9]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    from mercury_amd import EXPORTED_SYMBOLS, load_library
    lib = load_library()
    declared = _header_functions()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "declared in include/*.h but not exported: " + name
    assert sorted(EXPORTED_SYMBOLS) == declared


def test_struct_layouts_match_header():
    from mercury_amd import STATS_DTYPE
    from mercury_amd.physical_layer import Config, Info, TransmitConfig
    assert C.sizeof(TransmitConfig) == 56       # 5 doubles, uint64, 2 ints
    assert STATS_DTYPE.itemsize == 24           # 4 ints + 2 floats
    assert C.sizeof(Config) == 40               # 7 ints, 1 float, mfsk_ctrl_mode, test_puncture_nBits
    assert C.sizeof(Info) == 4 * 32
    from mercury_amd.physical_layer import ExplicitParams
    assert C.sizeof(ExplicitParams) == 44       # float, 2 ints, 3 unsigned, 5 ints (mgpu_explicit_params)


def test_explicit_parameters_are_validated_before_any_device_work():
    """mgpu_create_explicit: Nc / Nfft / Dx other than the reference's are MGPU_ERR_UNSUPPORTED, bad values MGPU_ERR_ARG, a frame geometry
    (Dy, Nsymb) whose data cells do not fit a codeword MGPU_ERR_TABLES - on any machine, before any device work."""
    from mercury_amd import load_library
    from mercury_amd.physical_layer import Config, ExplicitParams
    lib = load_library()
    h = C.c_void_p()
    good = Config(8, 50, 1, 1, 1, 0, 16, 0.0)
    for xp, want in ((ExplicitParams(0.0, 0, 0, 0, 0, 0, 64, 0, 0, 0, 0), 4), (ExplicitParams(0.0, 0, 0, 0, 0, 0, 0, 512, 0, 0, 0), 4),
                     (ExplicitParams(0.0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0), 4),
                     (ExplicitParams(0.0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 0), 3),        # mode 8's 24 symbols with Dy = 4: 1800 bits in the data cells
                     (ExplicitParams(0.0, 0, 0, 0, 0, 0, 0, 0, 0, 5, 24), 3),       # LOW_DENSITY lattice on the HIGH_DENSITY frame length: 1920 bits
                     (ExplicitParams(0.0, 0, 0, 0, 0, 0, 0, 0, 0, 3, 6), 3),        # too short: fewer bits than the code's parity
                     (ExplicitParams(0.0, 0, 0, 0, 0, 0, 0, 0, 0, -1, 0), 1), (ExplicitParams(0.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 256), 1),
                     (ExplicitParams(0.0, 23, 0, 0, 0, 0, 0, 0, 0, 0, 0), 1), (ExplicitParams(-1.0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), 1)):
        rc = lib.mgpu_create_explicit(C.byref(good), C.byref(xp), C.byref(h))
        assert rc == want and not h.value, (rc, want)
        assert lib.mgpu_last_error(None)


def test_bad_arguments_return_error_codes():
    from mercury_amd import load_library
    from mercury_amd.physical_layer import Config
    lib = load_library()
    h = C.c_void_p()
    for bad in (Config(17, 50, 1, 1, 1, 0, 16, 0.0), Config(-1, 50, 1, 1, 1, 0, 16, 0.0), Config(99, 50, 1, 1, 1, 0, 16, 0.0),
                Config(103, 50, 1, 1, 1, 0, 16, 0.0),
                Config(8, 0, 1, 1, 1, 0, 16, 0.0), Config(8, 50, 7, 1, 1, 0, 16, 0.0), Config(8, 50, 1, 1, 1, 0, 0, 0.0)):
        rc = lib.mgpu_create(C.byref(bad), C.byref(h))
        assert rc == 1 and not h.value
        assert lib.mgpu_last_error(None)
    assert lib.mgpu_create(None, C.byref(h)) == 1
    assert lib.mgpu_get_info(None, None) == 1


def test_no_gpu_means_loud_failure_not_fallback():
    """On the CPU-only build container mgpu_create must fail with MGPU_ERR_DEVICE; on a GPU box it succeeds."""
    from mercury_amd import MgpuError, RxPhy
    import torch
    if torch.cuda.is_available():
        rx = RxPhy(8, max_batch=2)
        assert rx.K == 600 and rx.E == 5616 and rx.payload_bytes == 73 and rx.frame_samples == 24 * 272
        rx.close()
    else:
        with pytest.raises(MgpuError) as e:
            RxPhy(8, max_batch=2)
        assert "mgpu_create failed (2)" in str(e.value)


def test_corrupt_table_blob_is_reported(tmp_path, monkeypatch):
    from mercury_amd import load_library
    from mercury_amd.physical_layer import Config
    lib = load_library()
    p = tmp_path / "bad.bin"
    p.write_bytes(b"MLDP" + b"\x00" * 40)
    monkeypatch.setenv("MERCURY_LDPC_TABLES", str(p))
    h = C.c_void_p()
    rc = lib.mgpu_create(C.byref(Config(8, 50, 1, 1, 1, 0, 4, 0.0)), C.byref(h))
    assert rc == 3 and not h.value
    assert b"LDPC" in lib.mgpu_last_error(None)


def test_host_libm_selfcheck_reports_the_pinned_platform():
    """mgpu_host_libm_selfcheck (host-only): this image is the platform the device restates (x86-64 glibc 2.35), so the host's tanh / atanh /
    atan / sincos and the restatement compiled for the host agree on every checked argument; the report says what was evaluated."""
    import ctypes as C
    import platform

    class Report(C.Structure):
        _fields_ = [("evaluated", C.c_longlong * 4), ("differed", C.c_longlong * 4), ("first", C.c_double * 4), ("differing", C.c_int),
                    ("libc_version", C.c_char * 32)]
    from mercury_amd import load_library
    lib = load_library()
    r = Report()
    n = lib.mgpu_host_libm_selfcheck(C.byref(r))
    assert n == r.differing and all(e > 100000 for e in r.evaluated), (n, list(r.evaluated))
    assert r.libc_version.decode() == "glibc " + platform.libc_ver()[1]
    if platform.libc_ver() == ("glibc", "2.35") and platform.machine() == "x86_64":
        assert n == 0 and list(r.differed) == [0, 0, 0, 0], list(r.differed)
    assert lib.mgpu_host_libm_selfcheck(None) == n                      # cached, NULL allowed


def test_fp64_decoder_kernels_compile_without_register_spills():
    """The fp64 decoder kernels run eight wavefronts per SIMD (64 vector registers each). Round 5 measured what a spill costs there: the
    resident variable record going to scratch was +6 % on the headline, a spill inside the bin loop more. The compiler's own report for
    every rate's kernel must say ScratchSize 0 (cross-compiled here, no GPU needed)."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_count
    lines = isa_count.assembly("ldpc.hip")
    for ne in (4, 5, 6, 7, 8):
        kern = "mgpu_ldpc_spa_kernel_ne%d" % ne
        start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
        end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
        foot = "\n".join(lines[end:end + 60])
        assert int(re.search(r"; ScratchSize: (\d+)", foot).group(1)) == 0, (kern, "spills")
        assert int(re.search(r"; NumVgprs: (\d+)", foot).group(1)) <= 64, kern
        assert not any("scratch_" in l for l in lines[start:end]), kern
        # the product walk's LDS reads must stay single ds_read_b64 (the compiler pairs them into ds_read2_b64 unless told not to, and the
        # LDS pipeline takes 8 cycles for one of those against 2 x 2.2: profiles/r05_lds_mask.json; rate 14/16 9.15 -> 8.19 ms). A toolchain that
        # starts pairing them again fails HERE rather than silently costing 10 %.
        body = [l.split()[0] for l in lines[start:end] if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.rstrip().endswith(":")]
        # (two ds_read2_b64 exist outside the walk, in the start-up / epilogue code; the two copies of the bin loop - first pass, steady state -
        # carry one read per unrolled walk step each: 2 x 16, at rate 14/16 2 x 48)
        assert body.count("ds_read2_b64") <= 2 and body.count("ds_read2st64_b64") == 0, (kern, "the walk's LDS reads were paired", body.count("ds_read2_b64"))
        assert body.count("ds_read_b64") >= 2 * (48 if ne == 8 else 16), (kern, body.count("ds_read_b64"))


def test_fp64_decoder_bin_loop_keeps_its_scalar_unit_savings():
    """Round 6 (profiles/NOTES.md R6.8, R6.10): a scalar instruction costs the fp64 decoder's bin loop three times a vector one, and the loop has no
    scalar register to spare - one more value carried across it and the allocator rematerialises fdlibm's double constants as literal pairs in
    every bin. What that round bought is pinned on the compiler's own assembly (cross-compiled here): the steady-state loop of the headline
    kernel holds at most 185 scalar instructions (round 5: 232), its polynomial constants are v_mov_b32 literals INSIDE the loop (not hoisted into
    registers the loop does not have, not turned back into s_mov_b32 pairs), and the kernel uses one kernel argument for the look thresholds."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_count
    lines = isa_count.assembly("ldpc.hip")
    kern = "mgpu_ldpc_spa_kernel_ne6"
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    loops, key = {}, None
    for l in lines[start:end]:
        if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
            m = re.search(r"Header=(BB\d+_\d+) Depth=2", l)
            key = m.group(1) if m else None
            continue
        t = l.strip()
        if key and t and t[0] not in ";.":
            loops.setdefault(key, []).append(t)
    steady = max(loops.values(), key=len)                     # the steady-state bin loop (tanh + walk + atanh); the other one is the first pass
    salu = sum(1 for t in steady if t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop")))
    assert salu <= 185, salu
    lits = [t for t in steady if t.startswith("v_mov_b32") and "0x" in t]
    assert len(lits) >= 18, len(lits)                         # nine double constants of tanh / atanh as word pairs (SPA_VCONST 575: invln2, Q1..Q5, 3, 6, 8 Lp1)
    assert sum(1 for t in steady if t.startswith("s_mov_b32") and ", 0x" in t) <= 8, [t for t in steady if t.startswith("s_mov_b32") and ", 0x" in t]
