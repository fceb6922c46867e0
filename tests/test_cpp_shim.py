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
