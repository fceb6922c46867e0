"""mercury_amd — MI355X-native implementation of Mercury's physical-layer RX hot path.

Only what the path needs lives here: ``csrc/`` (hand-written HIP kernels for gfx950 + the C-ABI
declared in ``include/mercury_gpu.h``), ``data/`` (the LDPC graphs as compact derived data) and
``physical_layer.py`` (a ctypes loader mirroring the reference's physical_layer surface).
"""
from .physical_layer import (DEC_GBF, DEC_MINSUM, DEC_SPA, EXPORTED_SYMBOLS, LIB_PATH, MgpuError, RxPhy,
                             STATS_DTYPE, load_library)

__all__ = ["RxPhy", "MgpuError", "DEC_GBF", "DEC_SPA", "DEC_MINSUM", "STATS_DTYPE", "load_library",
           "LIB_PATH", "EXPORTED_SYMBOLS"]
