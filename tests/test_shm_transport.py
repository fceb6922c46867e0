"""Transport on the output side of the RX path: the POSIX shared-memory byte ring that
RX_SHM mode publishes decoded payloads through (telecom_system.cc:2326-2333) — csrc/shm_transport.cpp against
the reference's own implementation (source/common/ring_buffer_posix.cc, compiled unmodified into oracle/_ref)
sharing the same memory objects, and against the reference's unmodified example client (examples/receiver.c).
Host-only code: these tests run without a GPU."""
import ctypes as C
import os
import subprocess
import threading
import time

import numpy as np
import pytest

import oraclelib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECEIVER = os.path.join(ROOT, "oracle", "_ref", "receiver")
needs_ref = pytest.mark.skipif(not oraclelib.RefLib.available(), reason="oracle/_ref not built")
_n = [0]


def _name():
    _n[0] += 1
    return ("/mgpu-test-%d-%d" % (os.getpid(), _n[0])).encode()


@pytest.fixture(scope="module")
def lib():
    from mercury_amd import load_library
    l = load_library()
    l.mgpu_shm_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    l.mgpu_shm_connect.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    for f in ("close", "destroy", "clear"):
        getattr(l, "mgpu_shm_" + f).argtypes = [C.c_void_p]
        getattr(l, "mgpu_shm_" + f).restype = None
    for f in ("used", "free", "capacity"):
        getattr(l, "mgpu_shm_" + f).argtypes = [C.c_void_p]
        getattr(l, "mgpu_shm_" + f).restype = C.c_size_t
    l.mgpu_shm_write.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    l.mgpu_shm_read.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    l.mgpu_shm_read_all.argtypes = [C.c_void_p, C.c_char_p]
    l.mgpu_shm_read_all.restype = C.c_long
    l.mgpu_shm_publish_decoded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return l


@pytest.fixture(scope="module")
def ref():
    r = C.CDLL(oraclelib.REF_SO)
    r.circular_buf_init_shm.argtypes = [C.c_size_t, C.c_char_p]
    r.circular_buf_init_shm.restype = C.c_void_p
    r.circular_buf_connect_shm.argtypes = [C.c_size_t, C.c_char_p]
    r.circular_buf_connect_shm.restype = C.c_void_p
    r.circular_buf_destroy_shm.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p]
    r.write_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    r.read_buffer.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    r.read_buffer_all.argtypes = [C.c_void_p, C.c_char_p]
    for f in ("size_buffer", "circular_buf_free_size", "circular_buf_capacity"):
        getattr(r, f).argtypes = [C.c_void_p]
        getattr(r, f).restype = C.c_size_t
    return r


def _create(lib, name, size):
    h = C.c_void_p()
    assert lib.mgpu_shm_create(name, size, C.byref(h)) == 0
    return h


def test_exports_and_argument_checks(lib):
    h = C.c_void_p()
    assert lib.mgpu_shm_create(None, 64, C.byref(h)) == 1
    assert lib.mgpu_shm_create(b"/x", 0, C.byref(h)) == 1
    assert lib.mgpu_shm_connect(b"/mgpu-does-not-exist", 64, C.byref(h)) == 1 and not h.value
    ring = _create(lib, _name(), 64)
    assert lib.mgpu_shm_write(ring, b"x" * 65, 65) == 1              # larger than the ring could ever hold
    assert (lib.mgpu_shm_capacity(ring), lib.mgpu_shm_used(ring), lib.mgpu_shm_free(ring)) == (64, 0, 64)
    lib.mgpu_shm_destroy(ring)


def test_wraparound_and_full_state(lib):
    name = _name()
    ring = _create(lib, name, 100)
    other = C.c_void_p()
    assert lib.mgpu_shm_connect(name, 100, C.byref(other)) == 0       # second handle on the same objects
    assert lib.mgpu_shm_connect(name, 99, C.byref(C.c_void_p())) == 1  # size mismatch (the reference asserts)
    rng = np.random.default_rng(0)
    sent, got = bytearray(), bytearray()
    buf = C.create_string_buffer(100)
    for _ in range(200):
        n = int(rng.integers(1, lib.mgpu_shm_free(ring) + 1)) if lib.mgpu_shm_free(ring) else 0
        if n:
            chunk = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            assert lib.mgpu_shm_write(ring, chunk, n) == 0
            sent += chunk
        assert lib.mgpu_shm_used(other) == len(sent) - len(got)
        m = int(rng.integers(0, lib.mgpu_shm_used(other) + 1))
        if m:
            assert lib.mgpu_shm_read(other, buf, m) == 0
            got += buf.raw[:m]
    assert lib.mgpu_shm_write(ring, b"a" * lib.mgpu_shm_free(ring), lib.mgpu_shm_free(ring)) == 0
    assert lib.mgpu_shm_free(ring) == 0 and lib.mgpu_shm_used(ring) == 100   # head == tail with full set
    n = lib.mgpu_shm_read_all(other, buf)
    got += buf.raw[:n]
    assert bytes(got[: len(sent)]) == bytes(sent) and lib.mgpu_shm_used(ring) == 0
    lib.mgpu_shm_close(other)
    lib.mgpu_shm_destroy(ring)


def test_clear_empties_a_wrapped_or_full_ring_for_every_handle(lib):
    """mgpu_shm_clear = clear_buffer / circular_buf_reset (ring_buffer_posix.h:75,79; ring_buffer_posix.cc:284-330): after it the ring is empty for every handle on the
    objects, whatever state it was in (wrapped, full), and carries traffic again from position 0."""
    name = _name()
    ring = _create(lib, name, 48)
    other = C.c_void_p()
    assert lib.mgpu_shm_connect(name, 48, C.byref(other)) == 0
    buf = C.create_string_buffer(48)
    for fill in (10, 48, 31):
        assert lib.mgpu_shm_write(ring, b"z" * 30, 30) == 0 and lib.mgpu_shm_read(other, buf, 30) == 0      # head and tail away from 0
        assert lib.mgpu_shm_write(ring, bytes(range(fill)), fill) == 0                                          # wraps; 48 = full
        assert lib.mgpu_shm_used(other) == fill
        lib.mgpu_shm_clear(other)
        assert (lib.mgpu_shm_used(ring), lib.mgpu_shm_free(ring), lib.mgpu_shm_used(other)) == (0, 48, 0)
        assert lib.mgpu_shm_write(ring, b"abcdefg", 7) == 0 and lib.mgpu_shm_read_all(other, buf) == 7 and buf.raw[:7] == b"abcdefg"
    lib.mgpu_shm_close(other)
    lib.mgpu_shm_destroy(ring)


def test_blocking_reader_and_writer(lib):
    ring = _create(lib, _name(), 32)
    out = []

    def reader():
        buf = C.create_string_buffer(32)
        n = lib.mgpu_shm_read_all(ring, buf)           # blocks: nothing written yet
        out.append(buf.raw[:n])

    t = threading.Thread(target=reader)
    t.start()
    time.sleep(0.2)
    assert t.is_alive()
    lib.mgpu_shm_write(ring, b"hello", 5)
    t.join(5)
    assert out == [b"hello"]
    lib.mgpu_shm_write(ring, b"x" * 30, 30)
    done = []

    def writer():
        lib.mgpu_shm_write(ring, b"y" * 10, 10)        # blocks: only 2 bytes free
        done.append(1)

    t = threading.Thread(target=writer)
    t.start()
    time.sleep(0.2)
    assert t.is_alive() and not done
    buf = C.create_string_buffer(32)
    lib.mgpu_shm_read(ring, buf, 20)
    t.join(5)
    assert done and lib.mgpu_shm_used(ring) == 20
    lib.mgpu_shm_destroy(ring)


@needs_ref
def test_reference_reader_on_our_ring_and_our_reader_on_the_reference_ring(lib, ref):
    rng = np.random.default_rng(1)
    buf = C.create_string_buffer(4096)
    # (a) ring created by this library, the reference's code connects, reads what we write and writes back
    name = _name()
    mine = _create(lib, name, 1000)
    theirs = ref.circular_buf_connect_shm(1000, name)
    assert theirs
    for _ in range(50):                                 # 50 x 73 bytes through a 1000-byte ring: wraps several times
        frame = rng.integers(0, 256, 73, dtype=np.uint8).tobytes()
        assert lib.mgpu_shm_write(mine, frame, 73) == 0
        assert ref.size_buffer(theirs) == 73 and ref.circular_buf_free_size(theirs) == 927
        assert ref.read_buffer_all(theirs, buf) == 73 and buf.raw[:73] == frame
        assert ref.write_buffer(theirs, frame[::-1], 73) == 0
        assert lib.mgpu_shm_used(mine) == 73
        assert lib.mgpu_shm_read(mine, buf, 73) == 0 and buf.raw[:73] == frame[::-1]
    lib.mgpu_shm_destroy(mine)
    # (b) ring created by the reference (as Mercury does at start-up), this library connects
    name = _name()
    theirs = ref.circular_buf_init_shm(512, name)
    mine = C.c_void_p()
    assert lib.mgpu_shm_connect(name, 512, C.byref(mine)) == 0
    assert lib.mgpu_shm_capacity(mine) == ref.circular_buf_capacity(theirs) == 512
    sent = b""
    for n in (300, 200, 12):
        chunk = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert lib.mgpu_shm_write(mine, chunk, n) == 0
        sent += chunk
    assert ref.size_buffer(theirs) == 512 and ref.circular_buf_free_size(theirs) == 0       # full: head == tail
    assert ref.read_buffer(theirs, buf, 400) == 0 and buf.raw[:400] == sent[:400]
    chunk = rng.integers(0, 256, 350, dtype=np.uint8).tobytes()
    assert ref.write_buffer(theirs, chunk, 350) == 0                                         # wraps
    n = lib.mgpu_shm_read_all(mine, buf)
    assert n == 462 and buf.raw[:n] == sent[400:] + chunk
    lib.mgpu_shm_close(mine)
    ref.circular_buf_destroy_shm(theirs, 512, name)


def test_publish_decoded_frames_only_and_counts_losses(lib):
    from mercury_amd import STATS_DTYPE
    F, stride, nbytes = 12, 75, 73                      # mode 8: 73 payload bytes in a 75-byte record
    rng = np.random.default_rng(2)
    payload = rng.integers(0, 256, (F, stride), dtype=np.uint8)
    stats = np.zeros(F, STATS_DTYPE)
    stats["message_decoded"] = [1, 0, 1, 1, 0, 1, 1, 1, 1, 0, 1, 1]
    ring = _create(lib, _name(), 5 * nbytes + 10)       # room for five frames
    pub, lost = C.c_int(), C.c_int()
    assert lib.mgpu_shm_publish_decoded(ring, payload.ctypes.data, stats.ctypes.data, F, stride, nbytes, C.byref(pub), C.byref(lost)) == 0
    assert (pub.value, lost.value) == (5, 4)            # 9 decoded, the ring takes 5 ("Decoded frame lost because of full buffer!")
    buf = C.create_string_buffer(1024)
    n = lib.mgpu_shm_read_all(ring, buf)
    want = b"".join(payload[f, :nbytes].tobytes() for f in (0, 2, 3, 5, 6))
    assert buf.raw[:n] == want
    lib.mgpu_shm_destroy(ring)


@needs_ref
@pytest.mark.skipif(not os.path.exists(RECEIVER), reason="reference example client not built")
def test_unmodified_reference_client_receives_published_frames(lib, tmp_path):
    """examples/receiver.c (compiled as it is) connects to /mercury-comm and appends whatever arrives to a file."""
    from mercury_amd import STATS_DTYPE
    ring = _create(lib, b"/mercury-comm", 131072)       # SHM_PAYLOAD_NAME / SHM_PAYLOAD_BUFFER_SIZE
    outfile = tmp_path / "rx.bin"
    proc = subprocess.Popen([RECEIVER, str(outfile)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        rng = np.random.default_rng(3)
        F, stride, nbytes = 64, 75, 73
        payload = rng.integers(0, 256, (F, stride), dtype=np.uint8)
        stats = np.zeros(F, STATS_DTYPE)
        stats["message_decoded"] = rng.integers(0, 2, F)
        want = b"".join(payload[f, :nbytes].tobytes() for f in range(F) if stats["message_decoded"][f])
        assert lib.mgpu_shm_publish_decoded(ring, payload.ctypes.data, stats.ctypes.data, F, stride, nbytes, None, None) == 0
        deadline = time.time() + 10
        while time.time() < deadline and (not outfile.exists() or outfile.stat().st_size < len(want)):
            time.sleep(0.05)
        assert outfile.read_bytes() == want
    finally:
        proc.kill()
        proc.wait()
        lib.mgpu_shm_destroy(ring)


@pytest.mark.gpu
def test_gpu_decoded_batch_reaches_a_reader_through_the_ring():
    """End of the path: frames decoded on the GPU are published exactly as RX_SHM_process_main publishes them."""
    from conftest import OPERATING_ESN0, SEED
    from mercury_amd import RxPhy, ShmRing
    cfg, F = 8, 48
    orc = oraclelib.Oracle(cfg)
    snrs = [OPERATING_ESN0[cfg] + 1.5] * (F - 8) + [-15.0] * 8          # the last eight cannot decode
    frames = [orc.gen_frame(SEED, 7000 + i, oraclelib.noise_amp_for(s)) for i, s in enumerate(snrs)]
    rx = RxPhy(cfg, max_batch=F)
    out = rx.receive(np.stack([f[0] for f in frames]))
    ring = ShmRing("/mgpu-test-gpu-%d" % os.getpid(), 8192)
    reader = ShmRing("/mgpu-test-gpu-%d" % os.getpid(), 8192, create=False)
    pub, lost = ring.publish_decoded(out["payload"], out["stats"], rx.payload_bytes)
    ok = out["stats"]["message_decoded"] == 1
    assert pub == int(ok.sum()) == F - 8 and lost == 0
    assert reader.read_all() == b"".join(frames[f][1].astype(np.uint8).tobytes() for f in range(F) if ok[f])
    reader.close()
    ring.close()
    rx.close()


@pytest.mark.gpu
@needs_ref
@pytest.mark.skipif(not os.path.exists(RECEIVER), reason="reference example client not built")
def test_batched_rx_shm_loop_feeds_the_unmodified_reference_client(tmp_path):
    """examples/rx_shm_batch.cpp = RX_SHM_process_main batched, C-ABI only: passband capture windows in, decoded payloads out
    through /mercury-comm, where the reference's own examples/receiver.c (unmodified) picks them up."""
    from test_receive_byte import make_windows
    from mercury_amd import ShmRing
    exe = tmp_path / "rx_shm_batch"
    lib = os.path.join(ROOT, "mercury_amd")
    subprocess.run(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "rx_shm_batch.cpp"),
                    "-o", str(exe), "-L", lib, "-lmercury_gpu", "-Wl,-rpath," + lib, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    cfg = 8
    orc = oraclelib.Oracle(cfg)
    specs = [("frame", 7 * 1088 + 333, 0.01, 1), ("silence", 0, 1e-9, 2), ("frame", 20 * 1088 + 17, 0.02, 3), ("noise", 0, 0.3, 4),
             ("frame", 12 * 1088 + 5, 0.01, 5), ("frame", 30 * 1088, 0.02, 6), ("frame", 9 * 1088 + 100, 0.01, 7)]
    wins, pls = make_windows(orc, specs, seed=123)
    (tmp_path / "windows.f64").write_bytes(wins.tobytes())
    want = b"".join(pls[i].astype(np.uint8).tobytes() for i, s in enumerate(specs) if s[0] == "frame")
    ring = ShmRing("/mercury-comm", 131072)                # Mercury's ring; the program attaches to it
    outfile = tmp_path / "rx.bin"
    client = subprocess.Popen([RECEIVER, str(outfile)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        r = subprocess.run([str(exe), str(cfg), str(tmp_path / "windows.f64"), "3"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert "7 windows, 5 decoded, 0 lost" in r.stdout, r.stdout
        deadline = time.time() + 10
        while time.time() < deadline and (not outfile.exists() or outfile.stat().st_size < len(want)):
            time.sleep(0.05)
        assert outfile.read_bytes() == want
        # the same capture as the audio device's INT32 samples (what the reference's capture thread receives, audioio.c:744): the loop
        # decodes the same windows and prints the same status lines as for the doubles these samples widen to
        i32 = np.rint(np.clip(wins, -1.0, 1.0) * 2147483647.0).astype(np.int32)
        (tmp_path / "windows.i32").write_bytes(i32.tobytes())
        (tmp_path / "widened.f64").write_bytes((i32.astype(np.float64) / 2147483647.0).tobytes())
        ra = subprocess.run([str(exe), str(cfg), str(tmp_path / "windows.i32"), "3"], capture_output=True, text=True, timeout=120)
        rb = subprocess.run([str(exe), str(cfg), str(tmp_path / "widened.f64"), "3"], capture_output=True, text=True, timeout=120)
        assert ra.returncode == 0 and rb.returncode == 0, (ra.stderr, rb.stderr)
        assert ra.stdout == rb.stdout and "7 windows, 5 decoded" in ra.stdout, (ra.stdout, rb.stdout)
    finally:
        client.kill()
        client.wait()
        ring.close()
