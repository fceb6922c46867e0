// One kernel per cl_ofdm / cl_psk method of the RX path, for callers that keep the reference's method-by-method
// call sequence (SURVEY.md §8b "signatures to keep"). These are the plain loops of the reference, one frame per
// workgroup, FP64 in the reference's operation order with no FMA contraction; they exist for drop-in fidelity
// and per-stage testing, not for speed — the production path is the fused front-end (frontend.hip), which the
// parity tests require to give the same values.
//   stage_symbol_demod ............. cl_ofdm::symbol_demod              ofdm.cc:862-867
//   stage_agc ...................... cl_ofdm::automatic_gain_control    ofdm.cc:1467-1498
//   stage_estimate ................. LS_/ZF_channel_estimator + interpolate_linear_col   ofdm.cc:1315-1451 / :1266-1313, interpolator.cc:163-254
//   stage_restore_amplitude ........ cl_ofdm::restore_channel_amplitude ofdm.cc:1453-1466
//   stage_equalize ................. cl_ofdm::channel_equalizer         ofdm.cc:1637-1647
//   stage_variance ................. cl_ofdm::measure_variance          ofdm.cc:1500-1521
//   stage_deframe .................. cl_ofdm::deframer                  ofdm.cc:837-852
//   stage_deinterleave_{c128,f32} .. deinterleaver                      interleaver.cc:77-109
//   stage_psk_demod ................ cl_psk::demod                      psk.cc:278-326
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"
#include "fe_math.h"

#define ST_THREADS 256

extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_symbol_demod_kernel(
    const double* __restrict__ in, int n, const double* __restrict__ twiddle, double* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) c2 tw[128];      // 16-byte aligned: fft256_twiddle reads it as ds_read_b128
    __shared__ __attribute__((aligned(16))) c2 fftb[(ST_THREADS / 64) * FFT256_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128; i += ST_THREADS) tw[fft256_tw_slot(i)] = {twiddle[2 * i], twiddle[2 * i + 1]};
    __syncthreads();
    const int s = blockIdx.x * (ST_THREADS / 64) + wave;
    if (s >= n) return;
    const c2* x = reinterpret_cast<const c2*>(in) + size_t(s) * 272 + 16;                  // gi_remover
    c2 r0 = x[lane], r1 = x[lane + 64], r2 = x[lane + 128], r3 = x[lane + 192];
    const Fft256CarrierLane fcl = fft256_carrier_lane(lane);
    const c2 v = wave_fft256_carriers(r0, r1, r2, r3, fftb + wave * FFT256_STRIDE, tw, lane, fcl);
    c2* y = reinterpret_cast<c2*>(out) + size_t(s) * 50;
    if (fcl.col >= 0) y[fcl.col] = {v.re / 256.0, v.im / 256.0};
}

// in place: gain = boost / mean_{pilots} |Y| (sum in pilot order), Y *= gain
extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_agc_kernel(MgpuDev T, double* __restrict__ grid) {
    __shared__ double gain;
    c2* g = reinterpret_cast<c2*>(grid) + size_t(blockIdx.x) * T.G;
    if (threadIdx.x == 0) {
        double amp = 0;
        for (int p = 0; p < T.nPilots; ++p) { const c2 y = g[T.pilot_cell[p]]; amp += sqrt(y.re * y.re + y.im * y.im); }
        amp /= T.nPilots;
        gain = T.pilot_boost / amp;
    }
    __syncthreads();
    const double a = gain;
    for (int c = threadIdx.x; c < T.G; c += ST_THREADS) g[c] = {g[c].re * a, g[c].im * a};
}

// estimate at the pilots (LS over the clipped window in row-major order, or ZF), then per-column linear
// inter/extrapolation for the data cells
extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_estimate_kernel(MgpuDev T, const double* __restrict__ grid, double* __restrict__ Hout) {
    const int Nc = 50, Ns = T.Nsymb, hw = T.lsw / 2;
    const c2* g = reinterpret_cast<const c2*>(grid) + size_t(blockIdx.x) * T.G;
    c2* H = reinterpret_cast<c2*>(Hout) + size_t(blockIdx.x) * T.G;
    for (int p = threadIdx.x; p < T.nPilots; p += ST_THREADS) {
        const int c = T.pilot_cell[p], i = c / Nc, j = c - i * Nc;
        if (T.estimator == 0) {
            const double x = T.pilot_val[c];
            H[c] = {g[c].re / x, g[c].im / x};
        } else {
            const int k0 = max(i - hw, 0), k1 = min(i + hw, Ns - 1), l0 = max(j - hw, 0), l1 = min(j + hw, Nc - 1);
            int n = 0;
            for (int k = k0; k <= k1; ++k) for (int l = l0; l <= l1; ++l) n += T.cell_type[k * Nc + l] != 0;
            const double w = T.ls_weight[n];
            double hr = 0, hi = 0;
            for (int k = k0; k <= k1; ++k)
                for (int l = l0; l <= l1; ++l) {
                    const int q = k * Nc + l;
                    if (!T.cell_type[q]) continue;
                    const double xw = T.pilot_val[q] < 0 ? -w : w;
                    hr += xw * g[q].re;
                    hi += xw * g[q].im;
                }
            H[c] = {hr, hi};
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < T.G; c += ST_THREADS) {
        if (T.cell_type[c]) continue;
        const int i = c / Nc, j = c - i * Nc;
        int prev = -1, next = -1;
        for (int r = i - 1; r >= 0; --r) if (T.cell_type[r * Nc + j]) { prev = r; break; }
        for (int r = i + 1; r < Ns; ++r) if (T.cell_type[r * Nc + j]) { next = r; break; }
        int a, b;
        if (prev >= 0 && next >= 0) { a = prev; b = next; }
        else if (prev < 0) { a = next; b = -1; for (int r = next + 1; r < Ns; ++r) if (T.cell_type[r * Nc + j]) { b = r; break; } }
        else { b = prev; a = -1; for (int r = prev - 1; r >= 0; --r) if (T.cell_type[r * Nc + j]) { a = r; break; } }
        H[c] = lerp(H[a * Nc + j], double(a), H[b * Nc + j], double(b), double(i));
    }
}

extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_restore_amplitude_kernel(MgpuDev T, double* __restrict__ Hio) {
    c2* H = reinterpret_cast<c2*>(Hio) + size_t(blockIdx.x) * T.G;
    for (int c = threadIdx.x; c < T.G; c += ST_THREADS) { const double th = get_angle(H[c]); double sn, cs; gl_sincos(th, &sn, &cs); H[c] = {1 * cs, 1 * sn}; }
}

extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_equalize_kernel(MgpuDev T, const double* __restrict__ grid, const double* __restrict__ Hin,
                                                                                  double* __restrict__ out) {
    const size_t base = size_t(blockIdx.x) * T.G;
    const c2* g = reinterpret_cast<const c2*>(grid) + base;
    const c2* H = reinterpret_cast<const c2*>(Hin) + base;
    c2* o = reinterpret_cast<c2*>(out) + base;
    for (int c = threadIdx.x; c < T.G; c += ST_THREADS) o[c] = cdiv(g[c], H[c]);
}

extern "C" __global__ __launch_bounds__(64) void mgpu_stage_variance_kernel(MgpuDev T, const double* __restrict__ grid, int F, double* __restrict__ var) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    const c2* g = reinterpret_cast<const c2*>(grid) + size_t(f) * T.G;
    double v = 0;
    for (int p = 0; p < T.nPilots; ++p) {
        const int c = T.pilot_cell[p];
        const double dr = g[c].re - T.pilot_val[c], di = g[c].im - 0.0;
        v += dr * dr + di * di;
    }
    var[f] = v / double(T.nPilots);
}

extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_deframe_kernel(MgpuDev T, const double* __restrict__ grid, double* __restrict__ out) {
    const c2* g = reinterpret_cast<const c2*>(grid) + size_t(blockIdx.x) * T.G;
    c2* o = reinterpret_cast<c2*>(out) + size_t(blockIdx.x) * T.nData;
    for (int i = threadIdx.x; i < T.nData; i += ST_THREADS) o[i] = g[T.data_cell[i]];
}

// out[i*bs + j] = in[j*nb + i] for the nb = n / bs full blocks, the tail copied through; `elem` bytes per item
extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_deinterleave_kernel(const unsigned char* __restrict__ in, int n, int bs, int elem,
                                                                                      unsigned char* __restrict__ out) {
    const int nb = n / bs;
    const unsigned char* x = in + size_t(blockIdx.x) * n * elem;
    unsigned char* y = out + size_t(blockIdx.x) * n * elem;
    for (int d = threadIdx.x; d < n; d += ST_THREADS) {
        int src = d;
        if (d < nb * bs) { const int i = d / bs, j = d - i * bs; src = j * nb + i; }
        for (int b = 0; b < elem; ++b) y[size_t(d) * elem + b] = x[size_t(src) * elem + b];
    }
}

extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_psk_demod_kernel(MgpuDev T, const double* __restrict__ syms, const float* __restrict__ variance,
                                                                                   float* __restrict__ llr) {
    const c2* s = reinterpret_cast<const c2*>(syms) + size_t(blockIdx.x) * T.nData;
    float* o = llr + size_t(blockIdx.x) * T.nBits;
    const float inv_var = 1 / variance[blockIdx.x];
    const int M = T.M, bps = T.bps;
    for (int k = threadIdx.x; k < T.nData; k += ST_THREADS) {
        float d0[5], d1[5];
        for (int b = 0; b < 5; ++b) { d0[b] = __builtin_inff(); d1[b] = __builtin_inff(); }
        for (int j = 0; j < M; ++j) {
            const double dr = s[k].re - T.constellation[2 * j], di = s[k].im - T.constellation[2 * j + 1];
            const float D = float(dr * dr + di * di);
            for (int b = 0; b < bps; ++b) {
                if ((j >> b) & 1) { if (D < d1[b]) d1[b] = D; }
                else { if (D < d0[b]) d0[b] = D; }
            }
        }
        for (int b = 0; b < bps; ++b) o[k * bps + (bps - 1 - b)] = inv_var * (d1[b] - d0[b]);
    }
}

// ---- the integer tail as separate stages: bit_energy_dispersal (interleaver.cc:111-117), bit_to_byte (misc.cc:107-130),
// CRC16_MODBUS_RTU_calc (crc16_modbus_rtu.cc:25-45). One byte per bit / per byte value, F rows.
extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_dispersal_kernel(const uint8_t* __restrict__ in, const uint8_t* __restrict__ scrambler,
                                                                                   int n, uint8_t* __restrict__ out) {
    const size_t base = size_t(blockIdx.x) * n;
    for (int i = threadIdx.x; i < n; i += ST_THREADS) out[base + i] = in[base + i] ^ scrambler[i];
}

extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_stage_bit_to_byte_kernel(const uint8_t* __restrict__ bits, int nbits, uint8_t* __restrict__ bytes) {
    const int nbytes = (nbits + 7) / 8;
    const uint8_t* b = bits + size_t(blockIdx.x) * nbits;
    uint8_t* o = bytes + size_t(blockIdx.x) * nbytes;
    for (int j = threadIdx.x; j < nbytes; j += ST_THREADS) {
        unsigned v = 0;
        for (int q = 0; q < 8 && j * 8 + q < nbits; ++q) v |= unsigned(b[j * 8 + q] & 1) << q;      // LSB first, trailing partial byte
        o[j] = uint8_t(v);
    }
}

extern "C" __global__ __launch_bounds__(64) void mgpu_stage_crc16_kernel(const uint8_t* __restrict__ bytes, int F, int n, uint16_t* __restrict__ crc_out) {
    const int f = blockIdx.x * 64 + threadIdx.x;
    if (f >= F) return;
    const uint8_t* b = bytes + size_t(f) * n;
    unsigned crc = 0xffff;
    for (int j = 0; j < n; ++j) {
        crc ^= b[j];
        for (int i = 0; i < 8; ++i) crc = (crc & 1) ? ((crc >> 1) ^ 0xA001) : (crc >> 1);
    }
    crc_out[f] = uint16_t(crc);
}
