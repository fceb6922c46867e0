"""ctypes bindings for the two CHECKERS (test infrastructure only):

* ``Oracle``  -> oracle/libmercury_oracle.so  (plain-C restatement, builds anywhere)
* ``RefLib``  -> oracle/_ref/libmercury_ref.so (the reference's own DSP objects; only built where
                 /root/reference exists, travels to the GPU box as a prebuilt .so)

Both expose the same Python surface so one test body can drive either.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLES = os.path.join(ROOT, "mercury_amd", "data", "mercury_ldpc_tables.bin")
ORACLE_SO = os.path.join(ROOT, "oracle", "libmercury_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libmercury_ref.so")

FLAG_AGC, FLAG_VAR_EQ, FLAG_NO_LDPC = 1, 2, 4
FLAGS_BASEBAND_TEST = 0                    # telecom_system.cc:155-198
FLAGS_RECEIVE_BYTE = FLAG_AGC | FLAG_VAR_EQ  # telecom_system.cc:1132-1345

INFO_FIELDS = ("cfg M bits_per_symbol K P N Nsymb Nc Nfft Ngi Nofdm nData nBits nPilots nVirtual nReal "
               "bit_blk tf_blk preamble_nsymb estimator amp_restore ls_window Cwidth Vwidth dwidth payload_bytes "
               "mfsk_M mfsk_nStreams active_nsymb active_nbits").split()


def ldpc_graph(K):
    """The Tanner graph of the rate-K/1600 code from the committed table blob (mercury_amd/data/mercury_ldpc_tables.bin, written from the
    compiled reference by oracle/gen_ldpc_tables.py): (checks, variables) = the variables of every check in row order, the checks of every variable."""
    import struct
    blob = open(TABLES, "rb").read()
    _, _, n = struct.unpack_from("<4sII", blob, 0)
    off = 12
    for _ in range(n):
        k, P, N, E, _, _ = struct.unpack_from("<6I", blob, off)
        off += 24
        cdeg = np.frombuffer(blob, "u1", P, off); off += P
        Cf = np.frombuffer(blob, "<u2", E, off); off += 2 * E
        vdeg = np.frombuffer(blob, "u1", N, off); off += N
        Vf = np.frombuffer(blob, "<u2", E, off); off += 2 * E
        if k == K:
            return np.split(Cf.astype(int), np.cumsum(cdeg)[:-1]), np.split(Vf.astype(int), np.cumsum(vdeg)[:-1])
    raise KeyError(K)


class Info(C.Structure):
    _fields_ = [(n, C.c_int) for n in INFO_FIELDS]


class RxOut(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("H", C.c_void_p), ("H_noamp", C.c_void_p), ("eq", C.c_void_p),
                ("syms", C.c_void_p), ("llr_demod", C.c_void_p), ("llr_ldpc", C.c_void_p),
                ("bits", C.c_void_p), ("bytes", C.c_void_p),
                ("variance", C.c_double), ("variance_f", C.c_float), ("agc_gain", C.c_double),
                ("mean_H", C.c_double), ("iterations", C.c_int), ("crc", C.c_int), ("all_zeros", C.c_int),
                ("snr_db", C.c_double)]


def build_oracle():
    """Compile the checkers (never the product). Safe to call repeatedly."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"], check=True)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class _Base:
    prefix = None

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def _init_info(self):
        i = Info()
        self._fn("get_info")(self.h, C.byref(i))
        self.info = i
        for n in INFO_FIELDS:
            setattr(self, n, getattr(i, n))
        self.frame_samples = self.active_nsymb * self.Nofdm

    def set_ctrl_mode(self, enable):
        """cl_telecom_system::set_mfsk_ctrl_mode (telecom_system.cc:1572-1585): short MFSK control frames."""
        self._fn("set_ctrl_mode")(self.h, C.c_int(1 if enable else 0))
        self._init_info()

    def set_test_puncture(self, nbits):
        """cl_telecom_system::test_puncture_nBits (telecom_system.cc:1186-1192): MFSK LLRs from this position on are erasures."""
        self._fn("set_test_puncture")(self.h, C.c_int(nbits))

    # ---- tables
    def frame_types(self):
        t = np.zeros(self.Nsymb * self.Nc, np.int32)
        self._fn("get_frame_types")(self.h, _p(t))
        return t

    def pilot_seq(self):
        s = np.zeros(self.nPilots, np.complex128)
        self._fn("get_pilot_seq")(self.h, _p(s))
        return s

    def scrambler(self):
        s = np.zeros(1600, np.int32)
        self._fn("get_scrambler")(self.h, _p(s))
        return s

    def constellation(self):
        c = np.zeros(self.M, np.complex128)
        self._fn("get_constellation")(self.h, _p(c))
        return c

    def prng(self, seed, n):
        o = np.zeros(n, np.int32)
        self._fn("prng")(C.c_uint(seed), C.c_int(n), _p(o))
        return o

    def crc16(self, data):
        d = np.ascontiguousarray(data, np.int32)
        f = self._fn("crc16")
        f.restype = C.c_uint
        return int(f(_p(d), C.c_int(len(d))))

    # ---- TX
    def payload_to_bits(self, payload):
        pl = np.ascontiguousarray(payload, np.int32)
        bits = np.zeros(1600, np.int32)
        self._fn("payload_to_bits")(self.h, _p(pl), C.c_int(len(pl)), _p(bits))
        return bits[: self.nReal].copy()

    def ldpc_encode(self, data):
        """cl_ldpc::encode: K data bits -> N code bits."""
        d = np.ascontiguousarray(data, np.int32)
        assert d.size == self.K
        out = np.zeros(self.N, np.int32)
        self._fn("ldpc_encode")(self.h, _p(d), _p(out))
        return out

    def tx(self, bits, scramble=1):
        b = np.zeros(1600, np.int32)
        b[: self.nReal] = bits[: self.nReal]
        out = np.zeros(self.frame_samples, np.complex128)
        self._fn("tx")(self.h, _p(b), C.c_int(scramble), _p(out))
        return out

    # ---- RX
    def rx(self, baseband, flags=FLAGS_RECEIVE_BYTE):
        bb = np.ascontiguousarray(baseband, np.complex128)
        assert bb.size == self.frame_samples
        G = self.Nsymb * self.Nc
        bufs = dict(grid=np.zeros(G, np.complex128), H=np.zeros(G, np.complex128),
                    H_noamp=np.zeros(G, np.complex128), eq=np.zeros(G, np.complex128),
                    syms=np.zeros(self.nData, np.complex128), llr_demod=np.zeros(self.nBits, np.float32),
                    llr_ldpc=np.zeros(1600, np.float32), bits=np.zeros(self.K, np.int32),
                    bytes=np.zeros((self.nReal + 7) // 8, np.int32))
        o = RxOut()
        for k, v in bufs.items():
            setattr(o, k, v.ctypes.data)
        self._fn("rx")(self.h, _p(bb), C.c_int(flags), C.byref(o))
        res = dict(bufs)
        for k in ("variance", "variance_f", "agc_gain", "mean_H", "iterations", "crc", "all_zeros", "snr_db"):
            res[k] = getattr(o, k)
        return res


# physical_config.cc:35-65: what every mode gets unless a context overrides it (mgpu_create_explicit / morc_create_explicit)
EXPLICIT_DEFAULTS = dict(pilot_boost=1.33, ls_window=20, pilot_seed=0, scrambler_seed=0, preamble_seed=1, Nsymb=0, Dy=0)


class Oracle(_Base):
    prefix = "morc_"

    def __init__(self, cfg, max_iters=50, explicit=None):
        """explicit: dict(pilot_boost, ls_window, pilot_seed, scrambler_seed, preamble_seed, Nsymb, Dy) overriding physical_config.cc:35-65
        (Nsymb / Dy 0 = what init() selects, telecom_system.cc:1810-1869)"""
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.morc_create_geometry.restype = C.c_void_p
        x = dict(EXPLICIT_DEFAULTS)
        x.update(explicit or {})
        h = self.lib.morc_create_geometry(C.c_int(cfg), C.c_int(max_iters), TABLES.encode(), C.c_float(x["pilot_boost"]), C.c_int(x["ls_window"]),
                                          C.c_uint(x["pilot_seed"]), C.c_uint(x["scrambler_seed"]), C.c_uint(x["preamble_seed"]),
                                          C.c_int(x["Nsymb"]), C.c_int(x["Dy"]))
        if not h:
            raise RuntimeError("morc_create failed")
        self.h = C.c_void_p(h)
        self.max_iters = max_iters
        self._init_info()

    def ldpc_decode(self, llr, alg=1):
        l = np.ascontiguousarray(llr, np.float32)
        bits = np.zeros(self.K, np.int32)
        it = self.lib.morc_ldpc_decode(self.h, _p(l), _p(bits), C.c_int(alg))
        return bits, int(it)

    def philox(self, seed, c0, c1, c2, c3):
        out = np.zeros(4, np.uint32)
        self.lib.morc_philox(C.c_uint64(seed), C.c_uint32(c0), C.c_uint32(c1), C.c_uint32(c2), C.c_uint32(c3), _p(out))
        return out

    def gen_payload(self, seed, frame):
        pl = np.zeros(1600, np.int32)
        self.lib.morc_gen_payload(self.h, C.c_uint64(seed), C.c_uint64(frame), _p(pl))
        return pl[: self.payload_bytes].copy()

    def gen_frame(self, seed, frame, noise_amp, channel=0):
        bb = np.zeros(self.frame_samples, np.complex128)
        pl = np.zeros(1600, np.int32)
        self.lib.morc_gen_frame(self.h, C.c_uint64(seed), C.c_uint64(frame), C.c_double(noise_amp),
                                C.c_int(channel), _p(bb), _p(pl))
        return bb, pl[: self.payload_bytes].copy()

    def channel(self, frame_c128, seed, frame, noise_amp, channel=0):
        x = np.array(frame_c128, np.complex128, copy=True)
        self.lib.morc_channel(self.h, C.c_uint64(seed), C.c_uint64(frame), C.c_double(noise_amp), C.c_int(channel), _p(x))
        return x

    def libm_tanh_atanh(self, x):
        xin = np.ascontiguousarray(x, np.float64).ravel()
        t = np.zeros_like(xin)
        a = np.zeros_like(xin)
        self.lib.morc_libm_tanh_atanh(_p(xin), C.c_int(xin.size), _p(t), _p(a))
        return t, a

    def libm_atan_sincos(self, x):
        xin = np.ascontiguousarray(x, np.float64).ravel()
        a, s, c = np.zeros_like(xin), np.zeros_like(xin), np.zeros_like(xin)
        self.lib.morc_libm_atan_sincos(_p(xin), C.c_int(xin.size), _p(a), _p(s), _p(c))
        return a, s, c

    def rx_many(self, baseband, flags=FLAGS_RECEIVE_BYTE):
        bb = np.ascontiguousarray(baseband, np.complex128).reshape(-1, self.frame_samples)
        n = bb.shape[0]
        iters = np.zeros(n, np.int32)
        crc = np.zeros(n, np.int32)
        pl = np.zeros((n, self.payload_bytes), np.uint8)
        f = self.lib.morc_rx_many
        f.restype = C.c_long
        tot = f(self.h, _p(bb), C.c_int(n), C.c_int(flags), _p(iters), _p(crc), _p(pl))
        return int(tot), iters, crc, pl


class LinkState(C.Structure):
    _fields_ = [("delay_of_last_decoded_message", C.c_int), ("freq_offset_of_last_decoded_message", C.c_double),
                ("mfsk_search_start", C.c_int), ("fixed_delay_plus_one", C.c_int)]


class ReceiveStats(C.Structure):
    _fields_ = [("iterations_done", C.c_int), ("crc", C.c_int), ("all_zeros", C.c_int), ("message_decoded", C.c_int),
                ("snr_db", C.c_double), ("delay", C.c_int), ("sync_trials", C.c_int), ("freq_offset", C.c_double),
                ("coarse_metric", C.c_double), ("frame_overflow_symbols", C.c_int), ("mean_H", C.c_double),
                ("signal_strength_dbm", C.c_double)]


def _receive_byte(self, passband, carrier=None, trials_max=2, use_last_time=1, use_last_freq=1, state=None, coarse_freq_sync=0):
    """The whole cl_telecom_system::receive_byte on one capture window (pinned against the reference's own: RefTelecomSystem.receive_byte below)."""
    x = np.ascontiguousarray(passband, np.float64)
    assert x.size == self.buffer_samples()
    out = np.zeros(1600, np.int32)
    rs = ReceiveStats()
    st = state if state is not None else LinkState(-1, 0.0, 0)
    self.lib.morc_receive_byte(self.h, _p(x), C.c_double(CARRIER if carrier is None else carrier), C.c_int(trials_max), C.c_int(use_last_time),
                               C.c_int(use_last_freq), C.c_int(coarse_freq_sync), C.byref(st), _p(out), C.byref(rs))
    res = {k: getattr(rs, k) for k, _ in ReceiveStats._fields_}
    res["payload"] = out[: self.payload_bytes].astype(np.uint8)
    res["state"] = st
    return res


def _buffer_samples(self):
    return int(self.lib.morc_buffer_nsymb(self.h)) * self.Nofdm * 4


Oracle.receive_byte = _receive_byte
Oracle.buffer_samples = _buffer_samples


class RefLib(_Base):
    prefix = "mref_"

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, cfg, max_iters=50, explicit=None):
        self.lib = C.CDLL(REF_SO)
        self.lib.mref_create_geometry.restype = C.c_void_p
        x = dict(EXPLICIT_DEFAULTS)
        x.update(explicit or {})
        self.h = C.c_void_p(self.lib.mref_create_geometry(C.c_int(cfg), C.c_int(max_iters), C.c_float(x["pilot_boost"]), C.c_int(x["ls_window"]),
                                                          C.c_uint(x["pilot_seed"]), C.c_uint(x["scrambler_seed"]), C.c_uint(x["preamble_seed"]),
                                                          C.c_int(x["Nsymb"]), C.c_int(x["Dy"])))
        if not self.h:
            raise RuntimeError("mref_create failed")
        self.max_iters = max_iters
        self._init_info()

    def ldpc_decode(self, llr, alg=1):
        l = np.ascontiguousarray(llr, np.float32)
        bits = np.zeros(self.K, np.int32)
        it = self.lib.mref_ldpc_decode(self.h, _p(l), _p(bits))
        return bits, int(it)


def noise_amp_for(esn0_db):
    """Per-component AWGN amplitude at the reference's 1/sqrt(Nfft) scale (telecom_system.cc:100,147)."""
    return float(10.0 ** (-esn0_db / 20.0) / np.sqrt(2.0))


# ---- synchroniser building blocks: same calls on both checkers ----------------
FS = 48000.0
BANDWIDTH = 48000.0 * 50.0 / 256 / 4          # physical_config.cc:80
CARRIER = BANDWIDTH / 2 + 300                 # physical_config.cc:84 with carrier_frequency_offset = 0
AMPLITUDE = float(np.sqrt(2.0))               # telecom_system.cc:69


FIRST_MESSAGE, MIDDLE_MESSAGE, FLUSH_MESSAGE = 0, 1, 2     # include/common/common_defines.h:197-199
SINGLE_MESSAGE, NO_FILTER_MESSAGE = 3, 4        # include/common/common_defines.h:200-201


class TxConfig(C.Structure):
    _fields_ = [("carrier_hz", C.c_double), ("carrier_amplitude", C.c_double), ("output_power_watt", C.c_double),
                ("preamble_papr_cut", C.c_double), ("data_papr_cut", C.c_double), ("start_sample", C.c_ulonglong),
                ("message_location", C.c_int), ("reserved", C.c_int)]


def _sync_methods(cls):
    def preamble(self):
        out = np.zeros(self.preamble_nsymb * self.Nc, np.complex128)
        self._fn("get_preamble")(self.h, _p(out))
        return out

    def fir_taps(self, which):
        t = np.zeros(64)
        f = self._fn("fir_taps")
        f.restype = C.c_int
        n = f(self.h, C.c_int(which), _p(t))
        return t[:n].copy()

    def passband_to_baseband(self, x, carrier=CARRIER, decimation=1, which=0, fs=FS, amplitude=AMPLITUDE):
        xin = np.ascontiguousarray(x, np.float64)
        out = np.zeros((xin.size + decimation - 1) // decimation, np.complex128)
        self._fn("passband_to_baseband")(self.h, _p(xin), C.c_int(xin.size), C.c_double(fs), C.c_double(carrier),
                                         C.c_double(amplitude), C.c_int(decimation), C.c_int(which), _p(out))
        return out

    def time_sync_preamble(self, bb, step, location_to_return=0, nTrials_max=1, interp=4):
        z = np.ascontiguousarray(bb, np.complex128)
        corr = C.c_double(0)
        f = self._fn("time_sync_preamble")
        f.restype = C.c_int
        d = f(self.h, _p(z), C.c_int(z.size), C.c_int(interp), C.c_int(location_to_return), C.c_int(step), C.c_int(nTrials_max), C.byref(corr))
        return int(d), float(corr.value)

    def freq_sync(self, bb, fs=FS):
        z = np.ascontiguousarray(bb, np.complex128)
        f = self._fn("freq_sync")
        f.restype = C.c_double
        return float(f(self.h, _p(z), C.c_double(BANDWIDTH / self.Nc), C.c_int(self.preamble_nsymb), C.c_double(fs)))

    def tx_passband(self, bits, carrier=CARRIER, fs=FS, amplitude=AMPLITUDE):
        b = np.zeros(1600, np.int32)
        b[: self.nReal] = bits[: self.nReal]
        out = np.zeros((self.preamble_nsymb + self.active_nsymb) * self.Nofdm * 4)
        f = self._fn("tx_passband")
        f.restype = C.c_int
        n = f(self.h, _p(b), C.c_double(fs), C.c_double(carrier), C.c_double(amplitude), _p(out))
        assert n == out.size
        return out

    def mfsk_pattern(self, which):
        """Known tone pattern as unscaled time-domain symbols: 0 = the mode's MFSK preamble, 1 = ACK, 2 = BREAK."""
        out = np.zeros(16 * self.Nofdm, np.complex128)
        f = self._fn("mfsk_pattern")
        f.restype = C.c_int
        n = f(self.h, C.c_int(which), _p(out))
        return out[: n * self.Nofdm].copy()

    def time_sync_mfsk(self, bb, search_start_symb=0, interp=4):
        z = np.ascontiguousarray(bb, np.complex128)
        f = self._fn("time_sync_mfsk")
        f.restype = C.c_int
        return int(f(self.h, _p(z), C.c_int(z.size), C.c_int(interp), C.c_int(search_start_symb)))

    def detect_ack_pattern(self, bb, which=1, interp=4):
        z = np.ascontiguousarray(bb, np.complex128)
        matched = C.c_int(0)
        f = self._fn("detect_ack_pattern")
        f.restype = C.c_double
        m = f(self.h, _p(z), C.c_int(z.size), C.c_int(interp), C.c_int(which), C.byref(matched))
        return float(m), int(matched.value)

    def get_pre_equalization_channel(self, carrier=CARRIER):
        """cl_telecom_system::get_pre_equalization_channel (telecom_system.cc:3108-3145) for a process that loaded this configuration:
        complex128 [Nc]."""
        out = np.zeros(self.Nc, np.complex128)
        f = self._fn("get_pre_equalization_channel")
        f.restype = C.c_int
        assert f(self.h, C.c_double(carrier), _p(out)) == self.Nc
        return out

    def transmit_byte(self, payload, carrier=CARRIER, message_location=SINGLE_MESSAGE, start_sample=0, amplitude=AMPLITUDE,
                      output_power_watt=0.1, preamble_papr_cut=7.0, data_papr_cut=10.0, pre_equalize=False):
        """cl_telecom_system::transmit_byte: payload bytes -> total_frame_size passband samples (None = message too long).
        pre_equalize: with the pre-equalisation of transmit_bit (telecom_system.cc:474-494; the table init() computes for `carrier`)."""
        pl = np.ascontiguousarray(payload, np.int32)
        c = TxConfig(carrier, amplitude, output_power_watt, preamble_papr_cut, data_papr_cut, start_sample, message_location, 1 if pre_equalize else 0)
        out = np.zeros((self.preamble_nsymb + self.Nsymb) * self.Nofdm * 4)
        f = self._fn("transmit_byte")
        f.restype = C.c_int
        n = f(self.h, _p(pl), C.c_int(pl.size), C.byref(c), _p(out))
        if n == -1:
            return None
        assert n == out.size, n
        return out

    def transmit_batch(self, payloads, nbytes=None, carrier=CARRIER, start_sample=0, amplitude=AMPLITUDE, output_power_watt=0.1,
                       preamble_papr_cut=7.0, data_papr_cut=10.0):
        """The signal path of cl_arq_controller::send_batch: [F, stride] message bytes -> [F, total_frame_size] filtered audio."""
        pl = np.ascontiguousarray(payloads, np.int32)
        F, stride = pl.shape
        nb = None if nbytes is None else np.ascontiguousarray(nbytes, np.int32)
        c = TxConfig(carrier, amplitude, output_power_watt, preamble_papr_cut, data_papr_cut, start_sample, NO_FILTER_MESSAGE, 0)
        out = np.zeros((F, (self.preamble_nsymb + self.Nsymb) * self.Nofdm * 4))
        f = self._fn("transmit_batch")
        f.restype = C.c_int
        n = f(self.h, _p(pl), C.c_int(stride), None if nb is None else _p(nb), C.c_int(F), C.byref(c), _p(out))
        assert n == out.size, n
        return out

    def transmit_stream(self, payloads, message_location, buffer=None, nbytes=None, carrier=CARRIER, start_sample=0, amplitude=AMPLITUDE,
                        output_power_watt=0.1, preamble_papr_cut=7.0, data_papr_cut=10.0):
        """transmit_byte with FIRST_MESSAGE (0) / MIDDLE_MESSAGE (1) / FLUSH_MESSAGE (2), F consecutive calls on one 3-frame
        passband_data_tx_buffer (`buffer`, float64 [3*total], updated in place; None = a fresh zeroed one).
        Returns ([F, total_frame_size] audio, buffer)."""
        pl = np.ascontiguousarray(payloads, np.int32)
        F, stride = pl.shape
        nb = None if nbytes is None else np.ascontiguousarray(nbytes, np.int32)
        total = (self.preamble_nsymb + self.Nsymb) * self.Nofdm * 4
        buf = np.zeros(3 * total) if buffer is None else buffer
        assert buf.dtype == np.float64 and buf.size == 3 * total and buf.flags.c_contiguous
        c = TxConfig(carrier, amplitude, output_power_watt, preamble_papr_cut, data_papr_cut, start_sample, message_location, 0)
        out = np.zeros((F, total))
        f = self._fn("transmit_stream")
        f.restype = C.c_int
        n = f(self.h, _p(pl), C.c_int(stride), None if nb is None else _p(nb), C.c_int(F), C.byref(c), _p(buf), _p(out))
        assert n == out.size, n
        return out, buf

    def generate_ack_pattern_passband(self, which=1, carrier=CARRIER, start_sample=0, amplitude=AMPLITUDE, output_power_watt=0.1,
                                      data_papr_cut=10.0):
        """cl_telecom_system::generate_ack_pattern_passband (which=1) / generate_break_pattern_passband (which=2)."""
        c = TxConfig(carrier, amplitude, output_power_watt, 7.0, data_papr_cut, start_sample, SINGLE_MESSAGE, 0)
        out = np.zeros(16 * self.Nofdm * 4)
        f = self._fn("generate_ack_pattern_passband")
        f.restype = C.c_int
        assert f(self.h, C.c_int(which), C.byref(c), _p(out)) == out.size
        return out

    for fn in (get_pre_equalization_channel, preamble, fir_taps, passband_to_baseband, time_sync_preamble, freq_sync, tx_passband, mfsk_pattern, time_sync_mfsk,
               detect_ack_pattern, transmit_byte, generate_ack_pattern_passband, transmit_batch, transmit_stream):
        setattr(cls, fn.__name__, fn)


_sync_methods(_Base)


# ---- the reference's own cl_telecom_system (oracle/ref_ts_harness.cc; oracle/_ref/libmercury_ref_ts.so) -----------------------------
REF_TS_SO = os.path.join(ROOT, "oracle", "_ref", "libmercury_ref_ts.so")
TS_INFO_FIELDS = ("M K P N Nsymb Nc Nfft Ngi Nofdm nData nBits nPilots nVirtual nReal bit_blk tf_blk preamble_nsymb estimator amp_restore "
                  "ls_window buffer_nsymb payload_bytes").split()


class RefTelecomSystem:
    """cl_telecom_system as the reference compiles it, driven as main.cc drives it for RX_SHM: load_configuration(cfg), then receive_byte
    per capture window (telecom_system.cc:646-1503). Only where /root/reference was present at build time (the .so travels)."""

    @staticmethod
    def available():
        return os.path.exists(REF_TS_SO)

    SO = REF_TS_SO

    def _create(self, cfg):
        if self.geometry:
            self.lib.mrefts_create_geometry.restype = C.c_void_p
            return self.lib.mrefts_create_geometry(C.c_int(cfg), C.c_int(self.geometry.get("Nsymb", 0)), C.c_int(self.geometry.get("Dy", 0)))
        self.lib.mrefts_create.restype = C.c_void_p
        return self.lib.mrefts_create(C.c_int(cfg))

    def __init__(self, cfg, geometry=None):
        """geometry: dict(Nsymb, Dy) written into default_configurations_telecom_system.ofdm_Nsymb / ofdm_pilot_configurator_Dy in front of
        load_configuration (physical_config.cc:38-40, telecom_system.cc:2775-2778)"""
        self.lib = C.CDLL(self.SO, mode=1)            # RTLD_LAZY: the GUI / ARQ / audio-driver functions the units mention stay unbound
        self.cfg = cfg
        self.geometry = geometry
        self.h = C.c_void_p(self._create(cfg))
        assert self.h.value, "could not create the reference object for cfg %d" % cfg
        o = (C.c_int * 32)()
        n = self.lib.mrefts_info(self.h, o)
        assert n == len(TS_INFO_FIELDS)
        self.info = dict(zip(TS_INFO_FIELDS, list(o)[:n]))
        self.payload_bytes = self.info["payload_bytes"]

    def buffer_samples(self):
        return int(self.lib.mrefts_buffer_samples(self.h))

    def receive_byte(self, passband, carrier=None, trials_max=2, use_last_time=1, use_last_freq=1, state=None, coarse_freq_sync=0):
        x = np.ascontiguousarray(passband, np.float64)
        assert x.size == self.buffer_samples()
        out = np.zeros(1600, np.int32)
        rs = ReceiveStats()
        st = state if state is not None else LinkState(-1, 0.0, 0)
        self.lib.mrefts_receive_byte(self.h, _p(x), C.c_double(CARRIER if carrier is None else carrier), C.c_int(trials_max), C.c_int(use_last_time),
                                     C.c_int(use_last_freq), C.c_int(coarse_freq_sync), C.byref(st), _p(out), C.byref(rs))
        res = {k: getattr(rs, k) for k, _ in ReceiveStats._fields_}
        res["payload"] = out[: self.payload_bytes].astype(np.uint8)
        res["state"] = st
        return res

    def carrier(self):
        self.lib.mrefts_carrier.restype = C.c_double
        return float(self.lib.mrefts_carrier(self.h))

    def transmit_byte(self, payload, message_location=SINGLE_MESSAGE, start_sample=0):
        """cl_telecom_system::transmit_byte with what load_configuration left in the object (carrier, 0.1 W, PAPR cuts, pre-equalisation)."""
        pl = np.ascontiguousarray(payload, np.int32)
        out = np.zeros((self.info["preamble_nsymb"] + self.info["Nsymb"]) * self.info["Nofdm"] * 4)
        n = self.lib.mrefts_transmit_byte(self.h, _p(pl), C.c_int(pl.size), C.c_int(message_location), C.c_ulong(start_sample), _p(out))
        assert n == out.size, (n, out.size)
        return out

    def pre_equalization_channel(self):
        out = np.zeros(self.info["Nc"], np.complex128)
        assert self.lib.mrefts_pre_equalization_channel(self.h, _p(out)) == self.info["Nc"]
        return out

    def generate_pattern(self, which=1, start_sample=0):
        out = np.zeros(16 * self.info["Nofdm"] * 4)
        n = self.lib.mrefts_generate_pattern(self.h, C.c_int(which), C.c_ulong(start_sample), _p(out))
        return out[:n]

    def detect_pattern(self, passband, which=1):
        x = np.ascontiguousarray(passband, np.float64)
        m = C.c_int(0)
        self.lib.mrefts_detect_pattern.restype = C.c_double
        v = self.lib.mrefts_detect_pattern(self.h, C.c_int(which), _p(x), C.c_int(x.size), C.byref(m))
        return float(v), int(m.value)

    def measure_signal_only(self, passband):
        x = np.ascontiguousarray(passband, np.float64)
        assert x.size == self.buffer_samples()
        self.lib.mrefts_measure_signal_only.restype = C.c_double
        return float(self.lib.mrefts_measure_signal_only(self.h, _p(x)))

    def transmit_buffer(self, buffer=None):
        """passband_data_tx_buffer of the FIRST / MIDDLE / FLUSH_MESSAGE calls: read it (buffer=None) or replace it."""
        n = 3 * (self.info["preamble_nsymb"] + self.info["Nsymb"]) * self.info["Nofdm"] * 4
        if buffer is None:
            out = np.zeros(n)
            assert self.lib.mrefts_transmit_buffer(self.h, _p(out), C.c_int(0)) == n
            return out
        b = np.ascontiguousarray(buffer, np.float64)
        assert b.size == n and self.lib.mrefts_transmit_buffer(self.h, _p(b), C.c_int(1)) == n

    def baseband_test_one_frame(self, esn0):
        """baseband_test_EsN0(esn0, 1) and what it left in data_container -> dict(baseband, grid, eq, syms, llr_demod, llr_ldpc, data_bits,
        decoded_bits, err = [Bits_total, Error_bits_total, Frames_total, Error_frames_total])."""
        i = self.info
        G, nReal = i["Nsymb"] * i["Nc"], i["nReal"]
        r = dict(baseband=np.zeros(i["Nofdm"] * i["Nsymb"], np.complex128), grid=np.zeros(G, np.complex128), eq=np.zeros(G, np.complex128),
                 syms=np.zeros(i["nData"], np.complex128), llr_demod=np.zeros(i["nBits"], np.float32), llr_ldpc=np.zeros(i["N"], np.float32),
                 data_bits=np.zeros(nReal, np.int32), decoded_bits=np.zeros(nReal, np.int32), err=np.zeros(4))
        self.lib.mrefts_baseband_test_one_frame(self.h, C.c_float(esn0), *[_p(r[k]) for k in ("baseband", "grid", "eq", "syms", "llr_demod", "llr_ldpc",
                                                                                           "data_bits", "decoded_bits", "err")])
        return r

    def set_ctrl_mode(self, enable):
        return int(self.lib.mrefts_set_mfsk_ctrl_mode(self.h, C.c_int(1 if enable else 0)))

    def load_configuration(self, cfg):
        """load_configuration(cfg), or return_to_last_configuration() for cfg = -1 -> (current_configuration, last_configuration, Nsymb, nReal)"""
        o = (C.c_int * 4)()
        self.lib.mrefts_load_configuration(self.h, C.c_int(cfg), o)
        return tuple(o)

    def seed(self, libc_seed, reference_seed=None):
        """srand() (cl_awgn's noise) and __srandom() (the self-simulations' data bits) of this library's copy of the reference"""
        self.lib.mrefts_seed(C.c_uint(libc_seed), C.c_uint(libc_seed if reference_seed is None else reference_seed))

    RAW_INTS = "iterations_done delay delay_of_last_decoded_message sync_trials message_decoded crc all_zeros mfsk_search_raw frame_overflow_symbols".split()
    RAW_DOUBLES = "freq_offset freq_offset_of_last_decoded_message SNR signal_stregth_dbm coarse_metric".split()

    def receive_byte_raw(self, passband, carrier=None, trials_max=2, use_last_time=1, use_last_freq=1, coarse_freq_sync=0):
        """receive_byte on the object AS IT IS (no member is reset): -> (returned, held, out) where returned / held are dicts of the
        st_receive_stats the call returned / the object holds afterwards and out the int array receive_byte wrote into (a caller's
        buffer that is NOT cleared between calls: kept on the wrapper)."""
        x = np.ascontiguousarray(passband, np.float64)
        assert x.size == self.buffer_samples()
        if not hasattr(self, "_raw_out"):
            self._raw_out = np.full(1600, -1, np.int32)
        ri, hi = (C.c_int * 9)(), (C.c_int * 9)()
        rd, hd = (C.c_double * 5)(), (C.c_double * 5)()
        self.lib.mrefts_receive_byte_raw(self.h, _p(x), C.c_double(CARRIER if carrier is None else carrier), C.c_int(trials_max), C.c_int(use_last_time),
                                         C.c_int(use_last_freq), C.c_int(coarse_freq_sync), _p(self._raw_out), ri, rd, hi, hd)
        ret = dict(zip(self.RAW_INTS, list(ri)), **dict(zip(self.RAW_DOUBLES, list(rd))))
        held = dict(zip(self.RAW_INTS, list(hi)), **dict(zip(self.RAW_DOUBLES, list(hd))))
        return ret, held, self._raw_out.copy()

    KEEP = -2147483648

    def set_loop_members(self, n_under=KEEP, fixed_delay=KEEP, search_raw=KEEP):
        self.lib.mrefts_set_loop_members(self.h, C.c_int(n_under), C.c_int(fixed_delay), C.c_int(search_raw))

    def close(self):
        if self.h:
            self.lib.mrefts_destroy(self.h)
            self.h = None


# ---- the same reference objects with the section-8b methods re-bound to libmercury_gpu.so (oracle/ref_ts_gpu_harness.cc) -------------
REF_TS_GPU_SO = os.path.join(ROOT, "oracle", "_ref", "libmercury_ref_ts_gpu.so")
GPU_METHODS = ("symbol_demod automatic_gain_control channel_estimator restore_channel_amplitude channel_equalizer "
               "channel_equalizer_without_amplitude_restoration measure_variance deframer deinterleaver_c128 deinterleaver_f32 psk_demod ldpc_decode "
               "bit_energy_dispersal bit_to_byte crc16 receive_byte").split()
MODE_REFERENCE, MODE_STAGES, MODE_SHADOW, MODE_WHOLE = 0, 1, 2, 4


class RefTelecomSystemGpu(RefTelecomSystem):
    """The reference's cl_telecom_system (unmodified object code) whose cl_ofdm / cl_psk / cl_ldpc methods, free functions and - with
    MODE_WHOLE - receive_byte are served by libmercury_gpu.so: the drop-in as the reference's own callers see it. mode 0 = everything falls
    through to the original machine code (works without a GPU)."""
    SO = REF_TS_GPU_SO

    @staticmethod
    def available():
        return os.path.exists(REF_TS_GPU_SO)

    def __init__(self, cfg, mode, max_iters=50):
        self.mode, self.max_iters = mode, max_iters
        super().__init__(cfg)

    def _create(self, cfg):
        self.lib.mreftsgpu_create.restype = C.c_void_p
        return self.lib.mreftsgpu_create(C.c_int(cfg), C.c_int(self.mode), C.c_int(self.max_iters))

    def set_mode(self, mode):
        self.mode = mode
        return int(self.lib.mreftsgpu_set_mode(self.h, C.c_int(mode)))

    def counters(self, reset=False):
        """{method: (calls, served by the GPU, differed from the original machine code under MODE_SHADOW)}"""
        o = (C.c_long * (3 * 32))()
        n = self.lib.mreftsgpu_counters(self.h, o, C.c_int(1 if reset else 0))
        assert n == len(GPU_METHODS), n
        return {m: (int(o[3 * i]), int(o[3 * i + 1]), int(o[3 * i + 2])) for i, m in enumerate(GPU_METHODS)}

    def error(self):
        self.lib.mreftsgpu_error.restype = C.c_char_p
        return self.lib.mreftsgpu_error(self.h).decode()

    def rx_rand_process_main(self, passband, frames_to_read=0):
        """cl_telecom_system::RX_RAND_process_main on one capture window -> (what it printed, frames_to_read afterwards)"""
        x = np.ascontiguousarray(passband, np.float64)
        assert x.size == self.buffer_samples()
        buf = C.create_string_buffer(1 << 16)
        ftr = C.c_int(frames_to_read)
        n = self.lib.mreftsgpu_rx_rand_process_main(self.h, _p(x), C.byref(ftr), buf, C.c_int(len(buf)))
        return buf.raw[:n].decode(errors="replace"), int(ftr.value)

    def held_receive_stats(self):
        i, d = (C.c_int * 9)(), (C.c_double * 5)()
        self.lib.mreftsgpu_receive_stats(self.h, i, d)
        return dict(zip(self.RAW_INTS, list(i)), **dict(zip(self.RAW_DOUBLES, list(d))))

    def close(self):
        if self.h:
            self.lib.mreftsgpu_destroy(self.h)
            self.h = None
