"""ctypes binding of the C-ABI in include/mercury_gpu.h.

This is only a thin loader: every computation happens in the HIP library
(``mercury_amd/libmercury_gpu.so``). There is NO CPU fallback — if the library is missing or no
GPU is visible, constructing :class:`RxPhy` raises.

Naming follows the reference's physical layer (source/physical_layer/telecom_system.cc): a
*frame* is one LDPC codeword worth of OFDM symbols for one ``CONFIG_n`` (0..16) or ``ROBUST_n``
(100..102, MFSK) mode, ``receive`` runs
the span of ``receive_byte`` after synchronisation (telecom_system.cc:1132-1345) on a batch of
frames, ``ldpc_decode`` is ``cl_ldpc::decode`` (ldpc.h:90) on a batch of LLR vectors.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MERCURY_GPU_LIB: another build of the same HIP library (kernel experiments: tools/build_variants.sh); never a CPU path
LIB_PATH = os.environ.get("MERCURY_GPU_LIB") or os.path.join(HERE, "libmercury_gpu.so")

DEC_GBF, DEC_SPA, DEC_MINSUM, DEC_SPA_FAST = 0, 1, 2, 3
EST_ZF, EST_LS = 0, 1


class Config(C.Structure):
    _fields_ = [("cfg", C.c_int), ("max_iters", C.c_int), ("decoder", C.c_int), ("agc", C.c_int),
                ("variance_source", C.c_int), ("device", C.c_int), ("max_batch", C.c_int),
                ("minsum_alpha", C.c_float), ("mfsk_ctrl_mode", C.c_int), ("test_puncture_nBits", C.c_int)]


INFO_FIELDS = ("cfg M bits_per_symbol K P N Nsymb Nc Nfft Ngi Nofdm nData nBits nPilots nVirtual nReal "
               "bit_blk tf_blk preamble_nsymb estimator amp_restore ls_window Cwidth Vwidth E "
               "payload_bytes payload_stride frame_samples mfsk_M mfsk_nStreams active_nsymb active_nbits").split()


class ExplicitParams(C.Structure):    # mgpu_explicit_params
    _fields_ = [("pilot_boost", C.c_float), ("ls_window", C.c_int), ("seeds_set", C.c_int), ("pilot_seed", C.c_uint),
                ("scrambler_seed", C.c_uint), ("preamble_seed", C.c_uint), ("Nc", C.c_int), ("Nfft", C.c_int), ("Dx", C.c_int), ("Dy", C.c_int), ("Nsymb", C.c_int)]


class Info(C.Structure):
    _fields_ = [(n, C.c_int) for n in INFO_FIELDS]


class Taps(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in "grid H eq syms llr_demod llr_ldpc variance agc_gain cycles".split()]


class ReceiveConfig(C.Structure):
    _fields_ = [("carrier_hz", C.c_double), ("time_sync_trials_max", C.c_int), ("use_last_good_time_sync", C.c_int),
                ("use_last_good_freq_offset", C.c_int), ("coarse_freq_sync_enabled", C.c_int)]


class TransmitConfig(C.Structure):     # include/mercury_tx.h
    _fields_ = [("carrier_hz", C.c_double), ("carrier_amplitude", C.c_double), ("output_power_watt", C.c_double),
                ("preamble_papr_cut", C.c_double), ("data_papr_cut", C.c_double), ("start_sample", C.c_uint64),
                ("message_location", C.c_int), ("phase_continuous", C.c_int)]


FIRST_MESSAGE, MIDDLE_MESSAGE, FLUSH_MESSAGE = 0, 1, 2
SINGLE_MESSAGE, NO_FILTER_MESSAGE, BATCH_MESSAGE = 3, 4, 16

LINK_STATE_DTYPE = np.dtype([("delay_of_last_decoded_message", "<i4"), ("freq_offset_of_last_decoded_message", "<f8"),
                             ("mfsk_search_start", "<i4"), ("fixed_delay_plus_one", "<i4")], align=True)
RECEIVE_STATS_DTYPE = np.dtype([("iterations_done", "<i4"), ("crc", "<i4"), ("all_zeros", "<i4"), ("message_decoded", "<i4"),
                                ("snr_db", "<f8"), ("delay", "<i4"), ("sync_trials", "<i4"), ("freq_offset", "<f8"),
                                ("coarse_metric", "<f8"), ("frame_overflow_symbols", "<i4"), ("mean_H", "<f8"),
                                ("signal_strength_dbm", "<f8")], align=True)

STATS_DTYPE = np.dtype([("iterations_done", "<i4"), ("crc", "<i4"), ("all_zeros", "<i4"),
                        ("message_decoded", "<i4"), ("variance", "<f4"), ("snr_db", "<f4")])

_lib = None


def load_library():
    """dlopen the HIP library; raise (never fall back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the Mercury RX path has no CPU fallback)" % LIB_PATH)
        # torch (used by bench/tests for device buffers, streams and torch.distributed) bundles its own
        # HIP runtime under the same SONAME; it has to be the first one in the process or the process ends
        # up with two HSA runtimes and the second one sees no GPU.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing, not a dependency of the library itself
            pass
        lib = C.CDLL(LIB_PATH)
        lib.mgpu_last_error.restype = C.c_char_p
        lib.mgpu_last_error.argtypes = [C.c_void_p]
        lib.mgpu_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        lib.mgpu_txgen_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mgpu_rx_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mgpu_frontend_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mgpu_ldpc_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]
        lib.mgpu_destroy.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


EXPORTED_SYMBOLS = [
    "mgpu_alloc_host", "mgpu_free_host", "mgpu_create", "mgpu_destroy", "mgpu_last_error", "mgpu_get_info", "mgpu_rx_batch", "mgpu_rx_batch_taps",
    "mgpu_host_libm_selfcheck", "mgpu_host_select_peak", "mgpu_host_fir_taps", "mgpu_host_preamble_carriers", "mgpu_host_mode_info", "mgpu_host_layout_stats",
    "mgpu_device_props_get", "mgpu_alloc_host_near", "mgpu_host_numa_node_of_pci", "mgpu_host_numa_cpus", "mgpu_pool_device_numa_node",
    "mgpu_ldpc_batch", "mgpu_ldpc_encode_batch", "mgpu_rx_batch_dev", "mgpu_frontend_dev", "mgpu_ldpc_batch_dev", "mgpu_txgen_dev",
    "mgpu_host_path_last", "mgpu_device_malloc", "mgpu_device_free", "mgpu_context_stream", "mgpu_synchronize", "mgpu_copy_to_host", "mgpu_copy_to_device",
    "mgpu_pool_rx_batch_dev", "mgpu_pool_ldpc_batch_dev", "mgpu_pool_txgen_dev",
    "mgpu_last_kernel_ms", "mgpu_kernel_ms_avg", "mgpu_decoder_hard_frames", "mgpu_enable_timing", "mgpu_debug_spa_math", "mgpu_debug_glibc_trig", "mgpu_debug_tsync_metric", "mgpu_debug_occupancy", "mgpu_debug_select_peak", "mgpu_debug_span_energy", "mgpu_debug_p2b_variant", "mgpu_debug_mfsk_sync", "mgpu_baseband_test_esn0", "mgpu_passband_test_esn0", "mgpu_passband_to_baseband", "mgpu_time_sync_preamble", "mgpu_freq_sync", "mgpu_last_sync_kernel_ms",
    "mgpu_time_sync_mfsk", "mgpu_detect_ack_pattern", "mgpu_detect_ack_pattern_from_passband",
    "mgpu_receive_buffer_nsymb", "mgpu_receive_byte_batch", "mgpu_receive_byte_batch_samples", "mgpu_measure_signal_only",
    "mgpu_host_pre_equalization_channel", "mgpu_context_pre_equalization_channel", "mgpu_set_pre_equalization_channel", "mgpu_transmit_bit_batch", "mgpu_transmit_frame_samples", "mgpu_transmit_byte_batch", "mgpu_transmit_byte_batch_dev", "mgpu_transmit_buffer", "mgpu_symbol_mod", "mgpu_generate_ack_pattern_passband",
    "mgpu_symbol_demod", "mgpu_automatic_gain_control", "mgpu_channel_estimator", "mgpu_restore_channel_amplitude", "mgpu_channel_equalizer",
    "mgpu_measure_variance", "mgpu_deframer", "mgpu_deinterleaver_c128", "mgpu_deinterleaver_f32", "mgpu_psk_demod",
    "mgpu_bit_energy_dispersal", "mgpu_bit_to_byte", "mgpu_crc16_modbus_rtu",
    "mgpu_shm_create", "mgpu_shm_connect", "mgpu_shm_close", "mgpu_shm_destroy", "mgpu_shm_used", "mgpu_shm_free", "mgpu_shm_capacity",
    "mgpu_shm_clear", "mgpu_shm_write", "mgpu_shm_read", "mgpu_shm_read_all", "mgpu_shm_publish_decoded",
    "mgpu_create_explicit", "mgpu_pool_create", "mgpu_pool_destroy", "mgpu_pool_size", "mgpu_pool_context", "mgpu_pool_last_error", "mgpu_pool_last_counters",
    "mgpu_pool_shard", "mgpu_pool_rx_batch", "mgpu_pool_ldpc_batch", "mgpu_pool_receive_byte_batch",
]


def cfg_explicit(M, rate16, preamble_nsymb, estimator):
    """MGPU_CFG_EXPLICIT (include/mercury_gpu.h): cfg id of an explicit (constellation, LDPC rate, preamble, estimator) combination."""
    mods, rates = {2: 0, 4: 1, 8: 2, 16: 3, 32: 4}, {1: 0, 2: 1, 3: 2, 4: 3, 5: 4, 6: 5, 8: 6, 14: 7}
    if M not in mods or rate16 not in rates or not 1 <= preamble_nsymb <= 8 or estimator not in (0, 1):
        return -1
    return 1000 + (((mods[M] * 8 + rates[rate16]) * 8 + (preamble_nsymb - 1)) * 2 + estimator)


class MgpuError(RuntimeError):
    pass


class DeviceProps(C.Structure):    # mgpu_device_props
    _fields_ = [("compute_units", C.c_int), ("clock_khz", C.c_int), ("memory_clock_khz", C.c_int), ("lds_bytes_per_cu", C.c_int),
                ("wavefront_size", C.c_int), ("numa_node", C.c_int), ("hbm_bytes", C.c_ulonglong),
                ("name", C.c_char * 64), ("gcn_arch", C.c_char * 32), ("pci_bus_id", C.c_char * 32)]


def device_props(device=0):
    """mgpu_device_props_get: compute units, clocks, LDS per compute unit, PCI address and NUMA node of a device, as a dict."""
    lib = load_library()
    p = DeviceProps()
    rc = lib.mgpu_device_props_get(C.c_int(device), C.byref(p))
    if rc != 0:
        raise MgpuError("mgpu_device_props_get(%d) failed (%d): no such device" % (device, rc))
    return {n: (getattr(p, n).decode() if isinstance(getattr(p, n), bytes) else getattr(p, n)) for n, _ in DeviceProps._fields_}


def pinned_empty(shape, dtype, device=None):
    """numpy array over page-locked host memory from mgpu_alloc_host — or, with ``device``, from mgpu_alloc_host_near (pages on that
    GPU's NUMA node). Keep the array alive while it is in use; the memory is released when the returned array's base object is collected."""
    lib = load_library()
    lib.mgpu_alloc_host.restype = C.c_void_p
    lib.mgpu_alloc_host.argtypes = [C.c_size_t]
    lib.mgpu_alloc_host_near.restype = C.c_void_p
    lib.mgpu_alloc_host_near.argtypes = [C.c_int, C.c_size_t]
    lib.mgpu_free_host.argtypes = [C.c_void_p]
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    ptr = lib.mgpu_alloc_host(n * dt.itemsize) if device is None else lib.mgpu_alloc_host_near(int(device), n * dt.itemsize)
    if not ptr:
        raise MgpuError("mgpu_alloc_host failed")

    class _Owner:
        def __del__(self):
            lib.mgpu_free_host(ptr)

    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    buf._owner = _Owner()
    return np.frombuffer(buf, dtype=dt).reshape(shape)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class RxPhy:
    """One GPU receive context for one Mercury mode (``load_configuration(cfg)`` equivalent)."""

    def __init__(self, cfg, max_iters=50, decoder=DEC_SPA, agc=1, variance_source=1, device=0,
                 max_batch=4096, minsum_alpha=0.0, mfsk_ctrl_mode=False, test_puncture_nbits=0, explicit=None):
        """explicit: dict with any of pilot_boost, ls_window, pilot_seed, scrambler_seed, preamble_seed, Dy, Nsymb (and Nc, Nfft, Dx, which
        must be the reference's) -> mgpu_create_explicit (include/mercury_gpu.h); the seeds override the reference's 0 / 0 / 1 together."""
        self.lib = load_library()
        self.h = C.c_void_p()
        c = Config(cfg, max_iters, decoder, agc, variance_source, device, max_batch, minsum_alpha, 1 if mfsk_ctrl_mode else 0, test_puncture_nbits)
        if explicit:
            seeds = any(k in explicit for k in ("pilot_seed", "scrambler_seed", "preamble_seed"))
            xp = ExplicitParams(float(explicit.get("pilot_boost", 0.0)), int(explicit.get("ls_window", 0)), 1 if seeds else 0,
                                int(explicit.get("pilot_seed", 0)), int(explicit.get("scrambler_seed", 0)), int(explicit.get("preamble_seed", 1)),
                                int(explicit.get("Nc", 0)), int(explicit.get("Nfft", 0)), int(explicit.get("Dx", 0)), int(explicit.get("Dy", 0)),
                                int(explicit.get("Nsymb", 0)))
            rc = self.lib.mgpu_create_explicit(C.byref(c), C.byref(xp), C.byref(self.h))
        else:
            rc = self.lib.mgpu_create(C.byref(c), C.byref(self.h))
        if rc != 0:
            raise MgpuError("mgpu_create failed (%d): %s" % (rc, self.lib.mgpu_last_error(None).decode()))
        self.config = c
        self.max_iters = max_iters
        self.max_batch = max_batch
        i = Info()
        self._ck(self.lib.mgpu_get_info(self.h, C.byref(i)))
        self.info = i
        for n in INFO_FIELDS:
            setattr(self, n, getattr(i, n))

    def _ck(self, rc):
        if rc != 0:
            raise MgpuError("mgpu error %d: %s" % (rc, self.lib.mgpu_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.mgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-buffer entry points -------------------------------------------------------------
    def receive(self, baseband, taps=False, want_llr=False):
        """baseband: complex128 [F, Nsymb*Nofdm]. Returns dict(payload, stats[, taps...])."""
        bb = np.ascontiguousarray(baseband, np.complex128).reshape(-1, self.frame_samples)
        F = bb.shape[0]
        payload = np.zeros((F, self.payload_stride), np.uint8)
        stats = np.zeros(F, STATS_DTYPE)
        out = {"payload": payload, "stats": stats}
        if taps:
            G = self.Nsymb * self.Nc
            t = dict(grid=np.zeros((F, G), np.complex128), H=np.zeros((F, G), np.complex128),
                     eq=np.zeros((F, G), np.complex128), syms=np.zeros((F, self.nData), np.complex128),
                     llr_demod=np.zeros((F, self.nBits), np.float32), llr_ldpc=np.zeros((F, 1600), np.float32),
                     variance=np.zeros(F, np.float64), agc_gain=np.zeros(F, np.float64), cycles=np.zeros(16, np.int64))
            if self.mfsk_M:      # no channel estimate / equalised grid on the MFSK path
                for k in ("H", "eq", "syms"):
                    del t[k]
            ts = Taps(**{k: v.ctypes.data for k, v in t.items()})
            self._ck(self.lib.mgpu_rx_batch_taps(self.h, _ptr(bb), C.c_int(F), _ptr(payload), _ptr(stats), C.byref(ts)))
            out.update(t)
        else:
            llr = np.zeros((F, 1600), np.float32) if want_llr else None
            self._ck(self.lib.mgpu_rx_batch(self.h, _ptr(bb), C.c_int(F), _ptr(payload), _ptr(stats), _ptr(llr)))
            if want_llr:
                out["llr_ldpc"] = llr
        return out

    def ldpc_decode(self, llr):
        """llr: float32 [F,1600] -> (bits uint8 [F,K], iterations int32 [F])  (cl_ldpc::decode)."""
        l = np.ascontiguousarray(llr, np.float32).reshape(-1, 1600)
        F = l.shape[0]
        bits = np.zeros((F, self.K), np.uint8)
        iters = np.zeros(F, np.int32)
        self._ck(self.lib.mgpu_ldpc_batch(self.h, _ptr(l), C.c_int(F), _ptr(bits), _ptr(iters)))
        return bits, iters

    # ---- device-buffer entry points (raw device pointers, e.g. torch tensor.data_ptr()) --------
    def ldpc_encode(self, bits):
        """cl_ldpc::encode: uint8 [F, K] (one byte per bit) -> uint8 [F, N]."""
        b = np.ascontiguousarray(bits, np.uint8)
        b = b.reshape(1, -1) if b.ndim == 1 else b
        if b.shape[1] != self.K:
            raise MgpuError("a data word is K = %d bits" % self.K)
        out = np.zeros((b.shape[0], self.N), np.uint8)
        self._ck(self.lib.mgpu_ldpc_encode_batch(self.h, _ptr(b), C.c_int(b.shape[0]), _ptr(out)))
        return out

    def receive_dev(self, d_baseband, F, d_payload, d_stats, d_llr=None, stream=None):
        self._ck(self.lib.mgpu_rx_batch_dev(self.h, d_baseband, F, d_payload, d_stats, d_llr, stream))

    def frontend_dev(self, d_baseband, F, d_llr, d_variance=None, stream=None):
        self._ck(self.lib.mgpu_frontend_dev(self.h, d_baseband, F, d_llr, d_variance, stream))

    def ldpc_decode_dev(self, d_llr, F, d_bits=None, d_iters=None, d_payload=None, d_stats=None, d_variance=None, stream=None):
        self._ck(self.lib.mgpu_ldpc_batch_dev(self.h, d_llr, F, d_bits, d_iters, d_payload, d_stats, d_variance, stream))

    def txgen_dev(self, seed, frame0, F, noise_amp, d_baseband, d_payload=None, channel=0, stream=None):
        self._ck(self.lib.mgpu_txgen_dev(self.h, seed, frame0, F, noise_amp, channel, d_baseband, d_payload, stream))

    def baseband_test_esn0(self, esn0_db, frames_per_point, seed=1, frame0=0, channel=0):
        """cl_telecom_system::baseband_test_EsN0 per Es/N0 point (BER_PLOT_baseband): list of dicts with cl_error_rate's counters."""
        pts = np.ascontiguousarray(np.atleast_1d(esn0_db), np.float64)
        out = (ErrorRate * pts.size)()
        self._ck(self.lib.mgpu_baseband_test_esn0(self.h, _ptr(pts), C.c_int(pts.size), C.c_longlong(frames_per_point), C.c_uint64(seed),
                                                  C.c_uint64(frame0), C.c_int(channel), out))
        return [{n: getattr(r, n) for n, _ in ErrorRate._fields_} for r in out]

    def debug_mfsk_sync(self, energy, size, search_start, variant):
        """Test hook: cl_ofdm::time_sync_mfsk's search on slot energies [W, nslots, Nc]; variant 0 host, 1 device kernel -> delay [W]."""
        e = np.ascontiguousarray(energy, np.float64)
        W, nslots = e.shape[0], e.shape[1]
        ss = None if search_start is None else np.ascontiguousarray(search_start, np.int32)
        d = np.zeros(W, np.int32)
        self._ck(self.lib.mgpu_debug_mfsk_sync(self.h, _ptr(e), C.c_int(W), C.c_int(nslots), C.c_int(size), _ptr(ss), C.c_int(variant), _ptr(d)))
        return d

    def debug_p2b_variant(self, variant):
        """Test hook (process-wide): -1 = sliding-tap passband_to_baseband kernels where they apply, 0 = generic kernel. Returns the old value."""
        return int(self.lib.mgpu_debug_p2b_variant(C.c_int(variant)))

    def debug_span_energy(self, z, wv, off, length, variant):
        """Test hook: (sum, count) of |z|^2 over `length` samples from off[j] in window wv[j] (clipped at the window end), sample order."""
        z = np.ascontiguousarray(z, np.complex128)
        wv = np.ascontiguousarray(wv, np.int32)
        off = np.ascontiguousarray(off, np.int32)
        s = np.zeros(wv.size, np.float64)
        c = np.zeros(wv.size, np.int32)
        self._ck(self.lib.mgpu_debug_span_energy(self.h, _ptr(z), C.c_int(z.shape[0]), C.c_int(z.shape[1]), _ptr(wv), _ptr(off), C.c_int(wv.size),
                                                 C.c_int(length), C.c_int(variant), _ptr(s), _ptr(c)))
        return s, c

    def passband_test_esn0(self, esn0_db, frames_per_point, carrier_hz, seed=1, frame0=0, want_windows=False, output_power_watt=0.1):
        """cl_telecom_system::passband_test_EsN0 per Es/N0 point (PLOT_PASSBAND): list of cl_error_rate dicts [, windows, sent]."""
        pts = np.ascontiguousarray(np.atleast_1d(esn0_db), np.float64)
        out = (ErrorRate * pts.size)()
        win = sent = None
        if want_windows:
            win = np.zeros((pts.size * frames_per_point, self.receive_buffer_samples()), np.float64)
            sent = np.zeros((pts.size * frames_per_point, self.payload_stride), np.uint8)
        self._ck(self.lib.mgpu_passband_test_esn0(self.h, _ptr(pts), C.c_int(pts.size), C.c_longlong(frames_per_point), C.c_uint64(seed),
                                                  C.c_uint64(frame0), C.c_double(carrier_hz), C.c_double(output_power_watt), out, _ptr(win) if want_windows else None,
                                                  _ptr(sent) if want_windows else None))
        res = [{n: getattr(r, n) for n, _ in ErrorRate._fields_} for r in out]
        return (res, win, sent) if want_windows else res

    # ---- synchroniser building blocks -------------------------------------------
    def passband_to_baseband(self, passband, carrier_hz, which=0, start=None, count=None, decimation=1):
        """passband: float64 [W, in_size]; carrier_hz scalar or [W]. -> complex128 [W, count]."""
        x = np.ascontiguousarray(passband, np.float64)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        W, n = x.shape
        fc = np.ascontiguousarray(np.broadcast_to(np.asarray(carrier_hz, np.float64), (W,)))
        st = None if start is None else np.ascontiguousarray(np.broadcast_to(np.asarray(start, np.int32), (W,)))
        if count is None:
            count = (n + decimation - 1) // decimation
        out = np.zeros((W, count), np.complex128)
        self._ck(self.lib.mgpu_passband_to_baseband(self.h, _ptr(x), C.c_int(W), C.c_int(n), _ptr(fc), C.c_int(which), _ptr(st),
                                                    C.c_int(count), C.c_int(decimation), _ptr(out)))
        return out

    def time_sync_preamble(self, baseband_interp, step, location_to_return=0, nTrials_max=1):
        z = np.ascontiguousarray(baseband_interp, np.complex128)
        z = z.reshape(1, -1) if z.ndim == 1 else z
        W, size = z.shape
        delay = np.zeros(W, np.int32)
        corr = np.zeros(W, np.float64)
        self._ck(self.lib.mgpu_time_sync_preamble(self.h, _ptr(z), C.c_int(W), C.c_int(size), C.c_int(step), C.c_int(location_to_return),
                                                  C.c_int(nTrials_max), _ptr(delay), _ptr(corr)))
        return delay, corr

    def debug_tsync_metric(self, baseband_interp, step, variant=-1, start=None, sub_size=None):
        """Test hook: the Schmidl-Cox metric of every candidate, [W][ncand]; variant 0 = staged kernel, 1 = streaming kernel;
        start / sub_size: per-window sub-range to search."""
        z = np.ascontiguousarray(baseband_interp, np.complex128)
        z = z.reshape(1, -1) if z.ndim == 1 else z
        W, size = z.shape
        L = self.preamble_nsymb * self.Nofdm * 4
        ncand = (size - L + step - 1) // step
        vals = np.zeros((W, ncand), np.float64)
        st = None if start is None else np.ascontiguousarray(start, np.int32)
        sz = None if sub_size is None else np.ascontiguousarray(sub_size, np.int32)
        self._ck(self.lib.mgpu_debug_tsync_metric(self.h, _ptr(z), C.c_int(W), C.c_int(size), C.c_int(step), C.c_int(variant),
                                                  _ptr(st) if st is not None else None, _ptr(sz) if sz is not None else None, _ptr(vals)))
        return vals

    def freq_sync(self, baseband):
        z = np.ascontiguousarray(baseband, np.complex128)
        z = z.reshape(1, -1) if z.ndim == 1 else z
        W, stride = z.shape
        out = np.zeros(W, np.float64)
        self._ck(self.lib.mgpu_freq_sync(self.h, _ptr(z), C.c_int(W), C.c_int(stride), _ptr(out)))
        return out

    def time_sync_mfsk(self, baseband_interp, search_start_symb=0):
        """cl_ofdm::time_sync_mfsk on W windows of interpolated baseband -> delay [W] (MFSK modes only)."""
        z = np.ascontiguousarray(baseband_interp, np.complex128)
        z = z.reshape(1, -1) if z.ndim == 1 else z
        W, size = z.shape
        delay = np.zeros(W, np.int32)
        self._ck(self.lib.mgpu_time_sync_mfsk(self.h, _ptr(z), C.c_int(W), C.c_int(size), C.c_int(search_start_symb), _ptr(delay)))
        return delay

    def detect_ack_pattern(self, baseband_interp, pattern=1):
        """cl_ofdm::detect_ack_pattern (pattern 1 = ACK, 2 = BREAK) -> (metric [W], matched [W])."""
        z = np.ascontiguousarray(baseband_interp, np.complex128)
        z = z.reshape(1, -1) if z.ndim == 1 else z
        W, size = z.shape
        metric = np.zeros(W, np.float64)
        matched = np.zeros(W, np.int32)
        self._ck(self.lib.mgpu_detect_ack_pattern(self.h, _ptr(z), C.c_int(W), C.c_int(size), C.c_int(pattern), _ptr(metric), _ptr(matched)))
        return metric, matched

    def detect_ack_pattern_from_passband(self, passband, carrier_hz, pattern=1):
        """detect_ack_pattern_from_passband / detect_break_pattern_from_passband (telecom_system.cc:1628-1710)."""
        x = np.ascontiguousarray(passband, np.float64)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        W, size = x.shape
        metric = np.zeros(W, np.float64)
        matched = np.zeros(W, np.int32)
        self._ck(self.lib.mgpu_detect_ack_pattern_from_passband(self.h, _ptr(x), C.c_int(W), C.c_int(size), C.c_double(carrier_hz),
                                                                C.c_int(pattern), _ptr(metric), _ptr(matched)))
        return metric, matched

    # ---- the whole of receive_byte on capture windows (include/mercury_rxloop.h) -------------
    def receive_buffer_samples(self):
        return int(self.lib.mgpu_receive_buffer_nsymb(self.h)) * self.Nofdm * 4

    def receive_byte(self, passband, carrier_hz, trials_max=2, use_last_good_time_sync=1, use_last_good_freq_offset=1, state=None,
                     coarse_freq_sync=0):
        """passband: [W, buffer samples] float64 - or the audio device's own samples, int32 (x / INT_MAX), int16 (x / 32768) or float32, which
        are widened on the device as the reference's capture thread widens them (audioio.c:893-936): same results, half / a quarter of the bytes
        over PCIe. Returns dict(payload [W, stride], stats [W] (RECEIVE_STATS_DTYPE), state)."""
        fmt = {np.dtype(np.int32): 1, np.dtype(np.int16): 2, np.dtype(np.float32): 3}.get(np.asarray(passband).dtype, 0)
        x = np.ascontiguousarray(passband, np.asarray(passband).dtype if fmt else np.float64)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        W, n = x.shape
        if n != self.receive_buffer_samples():
            raise MgpuError("a capture window is %d samples" % self.receive_buffer_samples())
        cfg = ReceiveConfig(carrier_hz, trials_max, use_last_good_time_sync, use_last_good_freq_offset, coarse_freq_sync)
        st = np.zeros(W, LINK_STATE_DTYPE) if state is None else np.ascontiguousarray(state, LINK_STATE_DTYPE)
        if state is None:
            st["delay_of_last_decoded_message"] = -1
        payload = np.zeros((W, self.payload_stride), np.uint8)
        stats = np.zeros(W, RECEIVE_STATS_DTYPE)
        if fmt:
            self._ck(self.lib.mgpu_receive_byte_batch_samples(self.h, _ptr(x), C.c_int(fmt), C.c_int(W), C.byref(cfg), _ptr(st), _ptr(payload), _ptr(stats)))
        else:
            self._ck(self.lib.mgpu_receive_byte_batch(self.h, _ptr(x), C.c_int(W), C.byref(cfg), _ptr(st), _ptr(payload), _ptr(stats)))
        return {"payload": payload, "stats": stats, "state": st}

    def transmit_frame_samples(self):
        return int(self.lib.mgpu_transmit_frame_samples(self.h))

    def transmit_config(self, carrier_hz, message_location=SINGLE_MESSAGE, start_sample=0, phase_continuous=0, carrier_amplitude=None,
                        output_power_watt=0.1, preamble_papr_cut=7.0, data_papr_cut=10.0):
        """The reference's defaults (telecom_system.cc:69, physical_config.cc:88,115-116) around the given carrier."""
        amp = float(np.sqrt(2.0)) if carrier_amplitude is None else carrier_amplitude
        return TransmitConfig(carrier_hz, amp, output_power_watt, preamble_papr_cut, data_papr_cut, start_sample, message_location,
                              phase_continuous)

    def transmit_buffer(self, buffer=None):
        """passband_data_tx_buffer of the FIRST / MIDDLE / FLUSH_MESSAGE calls: read it (buffer=None) or replace it."""
        n = 3 * self.transmit_frame_samples()
        if buffer is None:
            out = np.zeros(n)
            self._ck(self.lib.mgpu_transmit_buffer(self.h, _ptr(out), C.c_int(0)))
            return out
        b = np.ascontiguousarray(buffer, np.float64)
        if b.size != n:
            raise MgpuError("the transmit buffer is 3 frames = %d samples" % n)
        self._ck(self.lib.mgpu_transmit_buffer(self.h, _ptr(b), C.c_int(1)))
        return b

    def pre_equalization_channel(self, carrier_hz):
        """cl_telecom_system::get_pre_equalization_channel for this mode and carrier (host computation): complex128 [Nc]."""
        out = np.zeros(self.Nc, np.complex128)
        self._ck(self.lib.mgpu_context_pre_equalization_channel(self.h, C.c_double(carrier_hz), _ptr(out)))
        return out

    def set_pre_equalization_channel(self, channel):
        """Install (complex128 [Nc]) or remove (None) the table transmit_bit multiplies the carrier grids with."""
        if channel is None:
            self._ck(self.lib.mgpu_set_pre_equalization_channel(self.h, None))
        else:
            ch = np.ascontiguousarray(channel, np.complex128)
            assert ch.size == self.Nc
            self._ck(self.lib.mgpu_set_pre_equalization_channel(self.h, _ptr(ch)))

    def transmit_byte(self, payload, carrier_hz, nbytes=None, **kw):
        """cl_telecom_system::transmit_byte for F messages: payload uint8 [F, >= payload_bytes] -> float64 [F, total_frame_size]."""
        pl = np.ascontiguousarray(payload, np.uint8)
        pl = pl.reshape(1, -1) if pl.ndim == 1 else pl
        F, stride = pl.shape
        cfg = self.transmit_config(carrier_hz, **kw)
        nb = None if nbytes is None else np.ascontiguousarray(nbytes, np.int32)
        out = np.zeros((F, self.transmit_frame_samples()), np.float64)
        self._ck(self.lib.mgpu_transmit_byte_batch(self.h, _ptr(pl), C.c_int(stride), None if nb is None else _ptr(nb), C.c_int(F), C.byref(cfg),
                                                   _ptr(out)))
        return out

    def transmit_bit(self, bits, carrier_hz, **kw):
        """cl_telecom_system::transmit_bit for F frames: uint8 [F, nReal] data bits -> float64 [F, total_frame_size]."""
        b = np.ascontiguousarray(bits, np.uint8).reshape(-1, self.nReal)
        cfg = self.transmit_config(carrier_hz, **kw)
        out = np.zeros((b.shape[0], self.transmit_frame_samples()), np.float64)
        self._ck(self.lib.mgpu_transmit_bit_batch(self.h, _ptr(b), C.c_int(b.shape[0]), C.byref(cfg), _ptr(out)))
        return out

    def transmit_byte_dev(self, d_payload, payload_stride, F, d_passband, carrier_hz, d_nbytes=None, stream=None, **kw):
        cfg = self.transmit_config(carrier_hz, **kw)
        self._ck(self.lib.mgpu_transmit_byte_batch_dev(self.h, C.c_void_p(d_payload), C.c_int(payload_stride), C.c_void_p(d_nbytes), C.c_int(F),
                                                       C.byref(cfg), C.c_void_p(d_passband), C.c_void_p(stream)))

    def generate_ack_pattern_passband(self, pattern=1, carrier_hz=None, **kw):
        """cl_telecom_system::generate_ack_pattern_passband (pattern 1) / generate_break_pattern_passband (2) -> float64 [16*Nofdm*4]."""
        cfg = self.transmit_config(carrier_hz, **kw)
        out = np.zeros(16 * self.Nofdm * 4, np.float64)
        self._ck(self.lib.mgpu_generate_ack_pattern_passband(self.h, C.c_int(pattern), C.byref(cfg), _ptr(out)))
        return out

    def symbol_mod(self, carriers):
        """cl_ofdm::symbol_mod: complex128 [n, Nc] -> [n, Nofdm]."""
        x = np.ascontiguousarray(carriers, np.complex128).reshape(-1, self.Nc)
        out = np.zeros((x.shape[0], self.Nofdm), np.complex128)
        self._ck(self.lib.mgpu_symbol_mod(self.h, _ptr(x), C.c_int(x.shape[0]), _ptr(out)))
        return out

    def measure_signal_only(self, passband, carrier_hz):
        """cl_telecom_system::measure_signal_only: float64 [W, buffer samples] -> signal strength in dBm per window."""
        x = np.ascontiguousarray(passband, np.float64)
        x = x.reshape(1, -1) if x.ndim == 1 else x
        if x.shape[1] != self.receive_buffer_samples():
            raise MgpuError("a capture window is %d samples" % self.receive_buffer_samples())
        out = np.zeros(x.shape[0], np.float64)
        self._ck(self.lib.mgpu_measure_signal_only(self.h, _ptr(x), C.c_int(x.shape[0]), C.c_double(carrier_hz), _ptr(out)))
        return out

    def receive_byte_dev(self, d_passband, W, carrier_hz, trials_max=2, use_last_good_time_sync=1, use_last_good_freq_offset=1, state=None,
                         coarse_freq_sync=0):
        """receive_byte on W capture windows that already lie in device memory (d_passband: raw pointer); host results as above."""
        cfg = ReceiveConfig(carrier_hz, trials_max, use_last_good_time_sync, use_last_good_freq_offset, coarse_freq_sync)
        st = np.zeros(W, LINK_STATE_DTYPE) if state is None else np.ascontiguousarray(state, LINK_STATE_DTYPE)
        if state is None:
            st["delay_of_last_decoded_message"] = -1
        payload = np.zeros((W, self.payload_stride), np.uint8)
        stats = np.zeros(W, RECEIVE_STATS_DTYPE)
        self._ck(self.lib.mgpu_receive_byte_batch(self.h, C.c_void_p(d_passband), C.c_int(W), C.byref(cfg), _ptr(st), _ptr(payload), _ptr(stats)))
        return {"payload": payload, "stats": stats, "state": st}

    def receive_byte_samples_dev(self, d_capture, sample_format, W, carrier_hz, trials_max=2, use_last_good_time_sync=1, use_last_good_freq_offset=1,
                                 state=None, coarse_freq_sync=0):
        """receive_byte on W capture windows of INT32 (1) / INT16 (2) / FLOAT32 (3) samples in device memory (raw pointer)."""
        cfg = ReceiveConfig(carrier_hz, trials_max, use_last_good_time_sync, use_last_good_freq_offset, coarse_freq_sync)
        st = np.zeros(W, LINK_STATE_DTYPE) if state is None else np.ascontiguousarray(state, LINK_STATE_DTYPE)
        if state is None:
            st["delay_of_last_decoded_message"] = -1
        payload = np.zeros((W, self.payload_stride), np.uint8)
        stats = np.zeros(W, RECEIVE_STATS_DTYPE)
        self._ck(self.lib.mgpu_receive_byte_batch_samples(self.h, C.c_void_p(d_capture), C.c_int(sample_format), C.c_int(W), C.byref(cfg), _ptr(st),
                                                          _ptr(payload), _ptr(stats)))
        return {"payload": payload, "stats": stats, "state": st}

    def last_sync_kernel_ms(self):
        ms = C.c_float(0)
        self._ck(self.lib.mgpu_last_sync_kernel_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def debug_glibc_trig(self, x):
        """Device atan / sin / cos (csrc/glibc_trig.h) of a float64 array."""
        xin = np.ascontiguousarray(x, np.float64)
        a, s, c = np.zeros_like(xin), np.zeros_like(xin), np.zeros_like(xin)
        self._ck(self.lib.mgpu_debug_glibc_trig(self.h, _ptr(xin), C.c_int(xin.size), _ptr(a), _ptr(s), _ptr(c)))
        return a, s, c

    def debug_spa_math(self, x):
        """Device tanh / atanh (csrc/spa_math.h) of a float64 array -> (tanh, atanh[0 where |x|>=1])."""
        xin = np.ascontiguousarray(x, np.float64).ravel()
        t = np.zeros_like(xin)
        a = np.zeros_like(xin)
        self._ck(self.lib.mgpu_debug_spa_math(self.h, _ptr(xin), C.c_int(xin.size), _ptr(t), _ptr(a)))
        return t, a

    def enable_timing(self, on=True):
        self._ck(self.lib.mgpu_enable_timing(self.h, C.c_int(1 if on else 0)))

    def host_path_last(self):
        """Profile of the last chunked mgpu_rx_batch call: dict(chunk_frames, n_chunks, fill_ms, drain_ms, total_ms)."""
        a, b = C.c_int(), C.c_int()
        f, d, t = C.c_float(), C.c_float(), C.c_float()
        self._ck(self.lib.mgpu_host_path_last(self.h, C.byref(a), C.byref(b), C.byref(f), C.byref(d), C.byref(t)))
        return {"chunk_frames": a.value, "n_chunks": b.value, "fill_ms": f.value, "drain_ms": d.value, "total_ms": t.value}

    def decoder_hard_frames(self):
        """Frames the fp64 decoder has decided without iterating since this context was created (every |LLR| >= 200 and an odd parity check)."""
        n = C.c_longlong(0)
        self._ck(self.lib.mgpu_decoder_hard_frames(self.h, C.byref(n)))
        return int(n.value)

    def kernel_ms_avg(self):
        """(front-end ms, decoder ms, launches) averaged over the launches since enable_timing()."""
        ms = (C.c_float * 2)()
        n = C.c_int(0)
        self._ck(self.lib.mgpu_kernel_ms_avg(self.h, ms, C.byref(n)))
        return float(ms[0]), float(ms[1]), int(n.value)


# ---- multi-GPU pool (include/mercury_pool.h) ------------------------------------------------------------------------------
POOL_MAX_DEVICES = 16


class ErrorRate(C.Structure):
    """mgpu_error_rate: cl_error_rate's counters for one Es/N0 point."""
    _fields_ = [("esn0_db", C.c_double), ("Frames_total", C.c_longlong), ("Error_frames_total", C.c_longlong), ("Bits_total", C.c_longlong),
                ("Error_bits_total", C.c_longlong), ("BER", C.c_double), ("FER", C.c_double), ("avg_iterations", C.c_double),
                ("crc_ok_frames", C.c_longlong)]


class PoolCounters(C.Structure):
    _fields_ = [("n_devices", C.c_int), ("frames", C.c_longlong), ("decoded", C.c_longlong), ("ldpc_iterations", C.c_longlong),
                ("wall_ms", C.c_double), ("device_frames", C.c_int * POOL_MAX_DEVICES), ("device_ms", C.c_double * POOL_MAX_DEVICES)]


def pool_shard(F, G, g):
    """Frames of device g of G: (first, count) — mgpu_pool_shard, callable without a GPU."""
    lib = load_library()
    a, n = C.c_int(), C.c_int()
    lib.mgpu_pool_shard(C.c_int(F), C.c_int(G), C.c_int(g), C.byref(a), C.byref(n))
    return a.value, n.value


class RxPool:
    """One context + one host worker thread per device; a call is split into contiguous frame ranges (mgpu_pool_*)."""

    def __init__(self, cfg, devices, max_iters=50, decoder=DEC_SPA, agc=1, variance_source=1, max_batch=4096, minsum_alpha=0.0,
                 mfsk_ctrl_mode=False):
        self.lib = load_library()
        self.lib.mgpu_pool_context.restype = C.c_void_p
        self.lib.mgpu_pool_last_error.restype = C.c_char_p
        self.h = C.c_void_p()
        c = Config(cfg, max_iters, decoder, agc, variance_source, 0, max_batch, minsum_alpha, 1 if mfsk_ctrl_mode else 0)
        devs = (C.c_int * len(devices))(*devices)
        rc = self.lib.mgpu_pool_create(C.byref(c), devs, C.c_int(len(devices)), C.byref(self.h))
        if rc != 0:
            raise MgpuError("mgpu_pool_create failed (%d): %s" % (rc, self.lib.mgpu_pool_last_error(None).decode()))
        self.n_devices = len(devices)
        i = Info()
        ctx0 = C.c_void_p(self.lib.mgpu_pool_context(self.h, 0))
        if self.lib.mgpu_get_info(ctx0, C.byref(i)) != 0:
            raise MgpuError("mgpu_get_info failed")
        for n in INFO_FIELDS:
            setattr(self, n, getattr(i, n))
        self._ctx0 = ctx0

    def numa_nodes(self):
        """NUMA node of every context's device (mgpu_pool_device_numa_node; -1: the platform names none)."""
        return [int(self.lib.mgpu_pool_device_numa_node(self.h, C.c_int(g))) for g in range(self.n_devices)]

    def _ck(self, rc):
        if rc != 0:
            raise MgpuError("mgpu pool error %d: %s" % (rc, self.lib.mgpu_pool_last_error(self.h).decode()))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.mgpu_pool_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counters(self):
        c = PoolCounters()
        self._ck(self.lib.mgpu_pool_last_counters(self.h, C.byref(c)))
        return {"n_devices": c.n_devices, "frames": c.frames, "decoded": c.decoded, "ldpc_iterations": c.ldpc_iterations, "wall_ms": c.wall_ms,
                "device_frames": list(c.device_frames)[: c.n_devices], "device_ms": list(c.device_ms)[: c.n_devices]}

    def receive(self, baseband):
        bb = np.ascontiguousarray(baseband, np.complex128).reshape(-1, self.frame_samples)
        F = bb.shape[0]
        payload = np.zeros((F, self.payload_stride), np.uint8)
        stats = np.zeros(F, STATS_DTYPE)
        self._ck(self.lib.mgpu_pool_rx_batch(self.h, _ptr(bb), C.c_int(F), _ptr(payload), _ptr(stats)))
        return {"payload": payload, "stats": stats}

    def ldpc_decode(self, llr):
        l = np.ascontiguousarray(llr, np.float32).reshape(-1, 1600)
        F = l.shape[0]
        bits = np.zeros((F, self.K), np.uint8)
        iters = np.zeros(F, np.int32)
        self._ck(self.lib.mgpu_pool_ldpc_batch(self.h, _ptr(l), C.c_int(F), _ptr(bits), _ptr(iters)))
        return bits, iters

    # ---- device-resident shards: one pointer and one count per pool device (mercury_pool.h) ----
    @staticmethod
    def _ptrs(ptrs, n):
        if ptrs is None:
            return None
        assert len(ptrs) == n
        return (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ptrs])

    def shard(self, F):
        """The split the host-buffer calls use: [(first, count)] per device."""
        out = []
        for g in range(self.n_devices):
            a, b = C.c_int(), C.c_int()
            self.lib.mgpu_pool_shard(C.c_int(F), C.c_int(self.n_devices), C.c_int(g), C.byref(a), C.byref(b))
            out.append((a.value, b.value))
        return out

    def device_malloc(self, g, nbytes):
        self.lib.mgpu_device_malloc.restype = C.c_void_p
        ctx = C.c_void_p(self.lib.mgpu_pool_context(self.h, g))
        p = self.lib.mgpu_device_malloc(ctx, C.c_size_t(nbytes))
        if not p:
            raise MgpuError("mgpu_device_malloc(%d bytes) failed on pool device %d" % (nbytes, g))
        return p

    def device_free(self, g, ptr):
        self.lib.mgpu_device_free(C.c_void_p(self.lib.mgpu_pool_context(self.h, g)), C.c_void_p(ptr))

    def copy_to_host(self, g, dst, d_src):
        """dst: a numpy array; blocking copy from pool device g."""
        ctx = C.c_void_p(self.lib.mgpu_pool_context(self.h, g))
        rc = self.lib.mgpu_copy_to_host(ctx, _ptr(dst), C.c_void_p(d_src), C.c_size_t(dst.nbytes), None)
        if rc != 0:
            raise MgpuError("mgpu_copy_to_host failed (%d)" % rc)
        return dst

    def copy_to_device(self, g, d_dst, src):
        src = np.ascontiguousarray(src)
        ctx = C.c_void_p(self.lib.mgpu_pool_context(self.h, g))
        rc = self.lib.mgpu_copy_to_device(ctx, C.c_void_p(d_dst), _ptr(src), C.c_size_t(src.nbytes), None)
        if rc != 0:
            raise MgpuError("mgpu_copy_to_device failed (%d)" % rc)

    def txgen_dev(self, seed, frame0, counts, noise_amp, d_bb, d_payload=None, channel=0):
        n = self.n_devices
        cnt = (C.c_int * n)(*counts)
        self._ck(self.lib.mgpu_pool_txgen_dev(self.h, C.c_uint64(seed), C.c_uint64(frame0), cnt, C.c_double(noise_amp), C.c_int(channel),
                                              self._ptrs(d_bb, n), self._ptrs(d_payload, n)))

    def receive_dev(self, d_bb, counts, d_payload, d_stats):
        n = self.n_devices
        cnt = (C.c_int * n)(*counts)
        self._ck(self.lib.mgpu_pool_rx_batch_dev(self.h, self._ptrs(d_bb, n), cnt, self._ptrs(d_payload, n), self._ptrs(d_stats, n)))

    def ldpc_decode_dev(self, d_llr, counts, d_bits, d_iters):
        n = self.n_devices
        cnt = (C.c_int * n)(*counts)
        self._ck(self.lib.mgpu_pool_ldpc_batch_dev(self.h, self._ptrs(d_llr, n), cnt, self._ptrs(d_bits, n), self._ptrs(d_iters, n)))

    def enable_timing(self, on=True):
        for g in range(self.n_devices):
            self.lib.mgpu_enable_timing(C.c_void_p(self.lib.mgpu_pool_context(self.h, g)), C.c_int(1 if on else 0))

    def decoder_hard_frames(self):
        """Frames the pool's fp64 decoders have decided without iterating since the pool was created (sum over its devices)."""
        total = 0
        for g in range(self.n_devices):
            n = C.c_longlong(0)
            self.lib.mgpu_decoder_hard_frames(C.c_void_p(self.lib.mgpu_pool_context(self.h, g)), C.byref(n))
            total += int(n.value)
        return total

    def kernel_ms(self, g):
        """(front-end ms, decoder ms, launches) averaged since enable_timing on pool device g."""
        ms = (C.c_float * 2)()
        n = C.c_int()
        self.lib.mgpu_kernel_ms_avg(C.c_void_p(self.lib.mgpu_pool_context(self.h, g)), ms, C.byref(n))
        return float(ms[0]), float(ms[1]), n.value

    def receive_buffer_samples(self):
        return int(self.lib.mgpu_receive_buffer_nsymb(self._ctx0)) * self.Nofdm * 4

    def receive_byte(self, passband, carrier_hz, trials_max=2, use_last_good_time_sync=1, use_last_good_freq_offset=1, state=None,
                     coarse_freq_sync=0, W=None):
        """passband: float64 [W, buffer samples] in host memory, or (with W given) a raw device pointer when every context of the pool
        sits on the device that holds the windows (contexts time-sharing one GPU)."""
        if W is None:
            x = np.ascontiguousarray(passband, np.float64)
            x = x.reshape(1, -1) if x.ndim == 1 else x
            W, src = x.shape[0], _ptr(x)
        else:
            src = C.c_void_p(passband)
        cfg = ReceiveConfig(carrier_hz, trials_max, use_last_good_time_sync, use_last_good_freq_offset, coarse_freq_sync)
        st = np.zeros(W, LINK_STATE_DTYPE) if state is None else np.ascontiguousarray(state, LINK_STATE_DTYPE)
        if state is None:
            st["delay_of_last_decoded_message"] = -1
        payload = np.zeros((W, self.payload_stride), np.uint8)
        stats = np.zeros(W, RECEIVE_STATS_DTYPE)
        self._ck(self.lib.mgpu_pool_receive_byte_batch(self.h, src, C.c_int(W), C.byref(cfg), _ptr(st), _ptr(payload), _ptr(stats)))
        return {"payload": payload, "stats": stats, "state": st}
