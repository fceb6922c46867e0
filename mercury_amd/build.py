"""Build the in-tree HIP library ``mercury_amd/libmercury_gpu.so`` for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so
travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libmercury_gpu.so")
TABLES = os.path.join(HERE, "data", "mercury_ldpc_tables.bin")

HIP_SOURCES = ["api.hip", "rxloop.hip", "stages_api.hip", "stages.hip", "frontend.hip", "mfsk.hip", "ldpc.hip", "txgen.hip", "tx.hip", "stats.hip", "sync.hip"]
CXX_SOURCES = ["tables.cpp", "shm_transport.cpp", "pool.cpp", "numa.cpp", "libm_check.cpp"]
HEADERS = ["device_tables.h", "tables.hpp", "numa.hpp", "spa_math.h", "fft256.h", "fe_math.h", "glibc_trig.h", "glibc_trig_tables.h", "ctx.hpp", os.path.join(ROOT, "include", "mercury_gpu.h"),
           os.path.join(ROOT, "include", "mercury_shm.h"), os.path.join(ROOT, "include", "mercury_rxloop.h"), os.path.join(ROOT, "include", "mercury_stages.h"),
           os.path.join(ROOT, "include", "mercury_tx.h"), os.path.join(ROOT, "include", "mercury_pool.h")]

# -ffp-contract=off: the reference runs without FMA contraction (baseline x86-64); the FP64 front-end
# and the sum-product decoder reproduce its roundings exactly, which an fma() would break.
# -amdgpu-atomic-optimizer-strategy=None: the kernels' few LDS atomics are issued by one lane of a wavefront (the decoder's bin counter,
# the front-end's work queue); the optimiser's wave-reduction around them (mbcnt, bcnt, readfirstlane, add and a wait right behind the
# atomic) is eight vector instructions and an exposed LDS round trip per call for nothing.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
               "-Wno-unused-result", "-Wno-deprecated-declarations"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _sources_digest():
    h = hashlib.sha256()
    for name in HIP_SOURCES + CXX_SOURCES + HEADERS + [TABLES, os.path.abspath(__file__)]:
        path = name if os.path.isabs(name) else os.path.join(CSRC, name)
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()


def decoder_digest():
    """Digest of what the LDPC decoder kernels are compiled from (sources + flags): profiles/ stamps its PMC passes with it and
    bench.py only quotes a profile whose stamp matches the library it is timing."""
    h = hashlib.sha256()
    for name in ("ldpc.hip", "spa_math.h", "device_tables.h"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def _write_blob_c(path):
    with open(TABLES, "rb") as f:
        data = f.read()
    with open(path, "w") as f:
        f.write("/* generated from mercury_amd/data/mercury_ldpc_tables.bin by mercury_amd/build.py */\n")
        f.write("const unsigned long mgpu_ldpc_blob_size = %dUL;\n" % len(data))
        f.write("const unsigned char mgpu_ldpc_blob[] __attribute__((aligned(16))) = {\n")
        for i in range(0, len(data), 32):
            f.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")
        f.write("};\n")


def build(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link the shared library. Returns its path."""
    os.makedirs(BUILD, exist_ok=True)
    stamp = os.path.join(BUILD, "digest.txt")
    digest = _sources_digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in HIP_SOURCES + CXX_SOURCES:
        obj = os.path.join(BUILD, src + ".o")
        cmd = [hipcc] + HIPCC_FLAGS + ["-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".cpp"):
            cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-x", "c++", "-c", os.path.join(CSRC, src), "-o", obj]
            if src == "libm_check.cpp":      # the host's libm must be CALLED, not folded (fma() stays a libm call too: correct on any x86-64)
                cmd[1:1] = ["-fno-builtin"]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    blob_c = os.path.join(BUILD, "ldpc_blob.c")
    _write_blob_c(blob_c)
    blob_o = os.path.join(BUILD, "ldpc_blob.o")
    subprocess.run(["gcc", "-O1", "-fPIC", "-c", blob_c, "-o", blob_o], check=True)
    objs.append(blob_o)
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("== %s failed ==\n%s\n" % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-lrt"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
