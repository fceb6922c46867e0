"""ctypes binding of include/mercury_shm.h: the shared-memory byte ring Mercury's RX_SHM mode publishes decoded
payloads through (source/common/ring_buffer_posix.cc; default object name "/mercury-comm", 131072 bytes)."""
import ctypes as C

import numpy as np

from .physical_layer import MgpuError, load_library

PAYLOAD_NAME = "/mercury-comm"       # SHM_PAYLOAD_NAME, include/common/common_defines.h:209
PAYLOAD_BUFFER_SIZE = 131072         # SHM_PAYLOAD_BUFFER_SIZE, :208


class ShmRing:
    def __init__(self, base_name=PAYLOAD_NAME, size=PAYLOAD_BUFFER_SIZE, create=True):
        self.lib = load_library()
        l = self.lib
        l.mgpu_shm_create.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        l.mgpu_shm_connect.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
        for f in ("close", "destroy", "clear"):
            getattr(l, "mgpu_shm_" + f).argtypes = [C.c_void_p]
            getattr(l, "mgpu_shm_" + f).restype = None
        for f in ("used", "free", "capacity"):
            getattr(l, "mgpu_shm_" + f).argtypes = [C.c_void_p]
            getattr(l, "mgpu_shm_" + f).restype = C.c_size_t
        l.mgpu_shm_write.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        l.mgpu_shm_read_all.argtypes = [C.c_void_p, C.c_char_p]
        l.mgpu_shm_read_all.restype = C.c_long
        l.mgpu_shm_publish_decoded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self.h = C.c_void_p()
        self.size = size
        self.created = create
        rc = (l.mgpu_shm_create if create else l.mgpu_shm_connect)(base_name.encode(), size, C.byref(self.h))
        if rc != 0:
            raise MgpuError("cannot %s shared-memory ring %s (%d)" % ("create" if create else "connect to", base_name, rc))

    def used(self):
        return int(self.lib.mgpu_shm_used(self.h))

    def free(self):
        return int(self.lib.mgpu_shm_free(self.h))

    def write(self, data):
        b = bytes(data)
        if self.lib.mgpu_shm_write(self.h, b, len(b)) != 0:
            raise MgpuError("mgpu_shm_write failed")

    def read_all(self):
        """Blocks until something is in the ring (read_buffer_all)."""
        buf = C.create_string_buffer(self.size)
        n = self.lib.mgpu_shm_read_all(self.h, buf)
        return buf.raw[:n]

    def publish_decoded(self, payload, stats, payload_bytes):
        """RX_SHM_process_main's publishing step for a batch: returns (published, lost)."""
        p = np.ascontiguousarray(payload, np.uint8)
        s = np.ascontiguousarray(stats)
        pub, lost = C.c_int(), C.c_int()
        rc = self.lib.mgpu_shm_publish_decoded(self.h, p.ctypes.data, s.ctypes.data, p.shape[0], p.shape[1], payload_bytes,
                                               C.byref(pub), C.byref(lost))
        if rc != 0:
            raise MgpuError("mgpu_shm_publish_decoded failed (%d)" % rc)
        return pub.value, lost.value

    def close(self):
        if self.h and self.h.value:
            (self.lib.mgpu_shm_destroy if self.created else self.lib.mgpu_shm_close)(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
