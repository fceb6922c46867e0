"""mercury_amd — MI355X-native implementation of Mercury's physical-layer RX hot path (and its transmit mirror).

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels for gfx950 + the C-ABI
declared in ``include/mercury_gpu.h``), ``data/`` (the LDPC graphs as compact derived data) and
``physical_layer.py`` (a ctypes loader mirroring the reference's physical_layer surface), ``shm.py`` (the
shared-memory ring decoded payloads are published through, ``include/mercury_shm.h``).
"""
from .physical_layer import (DEC_GBF, DEC_MINSUM, DEC_SPA, DEC_SPA_FAST, EXPORTED_SYMBOLS, LIB_PATH, MgpuError, RxPhy,
                             RxPool, STATS_DTYPE, device_props, load_library, pool_shard)

from .shm import ShmRing  # noqa: E402

__all__ = ["ShmRing", "RxPhy", "RxPool", "pool_shard", "MgpuError", "DEC_GBF", "DEC_SPA", "DEC_MINSUM", "DEC_SPA_FAST", "STATS_DTYPE", "load_library",
           "LIB_PATH", "EXPORTED_SYMBOLS", "device_props"]
