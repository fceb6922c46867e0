// Synchroniser building blocks in front of the RX hot path, batched over capture
// windows. Each kernel reproduces one cl_ofdm method:
//   mgpu_p2b_kernel        passband_to_baseband  ofdm.cc:2316-2339  (mixer, cl_FIR::apply fir_filter.cc:164-187,
//                                                 rational_resampler DECIMATION ofdm.cc:2267-2278)
//   mgpu_tsync_metric_kernel  the Schmidl-Cox metric of time_sync_preamble_with_metric  ofdm.cc:1893-1941
//   mgpu_fsync_kernel      carrier_sampling_frequency_sync (Moose)  ofdm.cc:540-595
// FP64, reference operation order, no FMA contraction. The only non-bit-exact ingredients are the
// device cos/sin of the mixer phase and atan in get_angle (<= 1 ulp from glibc).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
struct c2 { double re, im; };
__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
}  // namespace

#define P2B_THREADS 256
#define P2B_MAXTAPS 64

// One block = 256 consecutive outputs of one window. out[k] = sum_j l[n+h-j]*c[j], n = start + k*decim,
// l[i] = in[i]*amp*(cos, sin)(2*pi*fc*i*Ts). The (255*decim + ntaps) mixed samples a block needs are formed once in LDS.
extern "C" __global__ __launch_bounds__(P2B_THREADS) void mgpu_p2b_kernel(
    const double* __restrict__ passband, int in_size, const double* __restrict__ carrier_hz, const int* __restrict__ start_opt,
    int start_all, int count, int decim, const double* __restrict__ taps, int ntaps, double fs, double amplitude,
    double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c2* l = reinterpret_cast<c2*>(smem);
    __shared__ double c[P2B_MAXTAPS];
    const int w = blockIdx.y, tid = threadIdx.x;
    const int start = start_opt ? start_opt[w] : start_all;
    const int k0 = blockIdx.x * P2B_THREADS;
    const int h = (ntaps - 1) / 2;
    const int span = (P2B_THREADS - 1) * decim + ntaps;          // inputs [base, base+span)
    const int base = start + k0 * decim + h - (ntaps - 1);
    const double* in = passband + size_t(w) * in_size;
    const double fc = carrier_hz[w];
    const double Ts = 1.0 / fs;
    if (tid < ntaps) c[tid] = taps[tid];
    for (int t = tid; t < span; t += P2B_THREADS) {
        const int i = base + t;
        c2 v = {0.0, 0.0};
        if (i >= 0 && i < in_size) {
            const double ph = 2 * M_PI * fc * double(i) * Ts;
            const double a = in[i] * amplitude;
            v = {a * cos(ph), a * sin(ph)};
        }
        l[t] = v;
    }
    __syncthreads();
    const int k = k0 + tid;
    if (k >= count) return;
    const int n = start + k * decim;
    double ar = 0, ai = 0;
    for (int j = 0; j < ntaps; ++j) {
        const int i = n + h - j;
        if (i >= 0 && i < in_size) {                            // cl_FIR::apply skips taps that fall outside the input
            const c2 v = l[i - base];
            ar += v.re * c[j];
            ai += v.im * c[j];
        }
    }
    out[(size_t(w) * count + k) * 2] = ar;
    out[(size_t(w) * count + k) * 2 + 1] = ai;
}

// Schmidl-Cox metric. One lane per candidate offset i = cand*step; the three accumulators run in the
// reference's order (a dependent chain of 2*(Ngi+Nfft/2)*Nsymb additions each), so the parallelism is across
// candidates. Lanes of a wave are `step` samples apart, which would make every global load touch 64 different
// cache lines; instead the wave walks the preamble in chunks of TS_CH samples and stages, per chunk, the two
// sample rows (a and b) of each of its 64 candidates in LDS with coalesced row loads (row r = 64 consecutive
// samples = 1 KiB = one wave-wide load), padded by one element per row so the per-lane reads are conflict-free.
#define TS_CH 64

extern "C" __global__ __launch_bounds__(64) void mgpu_tsync_metric_kernel(
    const double* __restrict__ bb, int size, int ncand, int step, int pre_nsymb, int ngi_i, int nfft_i, double* __restrict__ vals) {
    __shared__ c2 ra[64][TS_CH + 1];
    __shared__ c2 rb[64][TS_CH + 1];
    const int w = blockIdx.y, lane = threadIdx.x;
    const int cand0 = blockIdx.x * 64;
    const int cand = cand0 + lane;
    const c2* win = reinterpret_cast<const c2*>(bb) + size_t(w) * size;
    const int sym = ngi_i + nfft_i;
    double cc = 0, na = 0, nb = 0;
    // segments of the reference's loops: per preamble symbol l, (a = l*sym, b = a + Nfft, len Ngi) then
    // (a = l*sym + Ngi, b = a + Nfft/2, len Nfft/2)
    for (int l = 0; l < pre_nsymb; ++l) {
        for (int part = 0; part < 2; ++part) {
            const int a_off = l * sym + (part ? ngi_i : 0);
            const int b_off = a_off + (part ? nfft_i / 2 : nfft_i);
            const int len = part ? nfft_i / 2 : ngi_i;
            for (int m0 = 0; m0 < len; m0 += TS_CH) {
                const int n = min(TS_CH, len - m0);
                // stage rows: row r belongs to candidate cand0 + r
                for (int r = 0; r < 64; ++r) {
                    const long base = long(cand0 + r) * step;
                    if (cand0 + r < ncand && lane < n) {
                        ra[r][lane] = win[base + a_off + m0 + lane];
                        rb[r][lane] = win[base + b_off + m0 + lane];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (cand < ncand) {
                    for (int m = 0; m < n; ++m) {
                        const c2 x = ra[lane][m], y = rb[lane][m];
                        cc += x.re * y.re; na += x.re * x.re; nb += y.re * y.re;
                        cc += x.im * y.im; na += x.im * x.im; nb += y.im * y.im;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    if (cand >= ncand) return;
    if (na < 0.001 || nb < 0.001) cc = 0.0;
    else cc = cc / sqrt(na * nb);
    vals[size_t(w) * ncand + cand] = cc;
}

// Same metric for small steps (fine search, step 1): neighbouring lanes read neighbouring samples, so plain
// global loads are already coalesced and L1 serves the 64-fold overlap between candidates.
extern "C" __global__ __launch_bounds__(64) void mgpu_tsync_metric_dense_kernel(
    const double* __restrict__ bb, int size, int ncand, int step, int pre_nsymb, int ngi_i, int nfft_i, double* __restrict__ vals) {
    const int w = blockIdx.y;
    const int cand = blockIdx.x * 64 + threadIdx.x;
    if (cand >= ncand) return;
    const c2* data = reinterpret_cast<const c2*>(bb) + size_t(w) * size + size_t(cand) * step;
    const int sym = ngi_i + nfft_i;
    double cc = 0, na = 0, nb = 0;
    for (int l = 0; l < pre_nsymb; ++l) {
        const c2* a = data + l * sym;
        const c2* b = a + nfft_i;
        for (int m = 0; m < ngi_i; ++m) {
            const c2 x = a[m], y = b[m];
            cc += x.re * y.re; na += x.re * x.re; nb += y.re * y.re;
            cc += x.im * y.im; na += x.im * x.im; nb += y.im * y.im;
        }
        a = data + l * sym + ngi_i;
        b = data + l * sym + ngi_i + nfft_i / 2;
        for (int m = 0; m < nfft_i / 2; ++m) {
            const c2 x = a[m], y = b[m];
            cc += x.re * y.re; na += x.re * x.re; nb += y.re * y.re;
            cc += x.im * y.im; na += x.im * x.im; nb += y.im * y.im;
        }
    }
    if (na < 0.001 || nb < 0.001) cc = 0.0;
    else cc = cc / sqrt(na * nb);
    vals[size_t(w) * ncand + cand] = cc;
}

// Moose: up to two preamble symbols, each as two 256-point FFTs of a half symbol repeated twice.
extern "C" __global__ __launch_bounds__(256) void mgpu_fsync_kernel(
    const double* __restrict__ bb, int stride, int pre_half, const double* __restrict__ twiddle, double carrier_freq_width,
    double* __restrict__ freq_out) {
    __shared__ c2 v[4][256];
    __shared__ c2 tw[128];
    __shared__ c2 dep[4][50];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const c2* in = reinterpret_cast<const c2*>(bb) + size_t(w) * stride;
    if (tid < 128) tw[tid] = {twiddle[2 * tid], twiddle[2 * tid + 1]};
    __syncthreads();
    const int j = wave >> 1, halfsel = wave & 1;                 // wave -> (symbol j, first/second half)
    const bool act = j < pre_half;
    if (act) {
        for (int i = lane; i < 256; i += 64) v[wave][__brev(unsigned(i)) >> 24] = in[j * 272 + (i & 127) + halfsel * 128];
        __builtin_amdgcn_wave_barrier();
        for (int size = 2; size <= 256; size <<= 1) {
            const int half = size >> 1, step = 256 / size;
            for (int b = lane; b < 128; b += 64) {
                const int q = b & (half - 1);
                const int i0 = ((b - q) << 1) + q, i1 = i0 + half;
                const c2 t = cmul(tw[q * step], v[wave][i1]);
                const c2 u = v[wave][i0];
                v[wave][i1] = {u.re - t.re, u.im - t.im};
                v[wave][i0] = {u.re + t.re, u.im + t.im};
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < 50) {
            const int bin = lane < 25 ? lane + 256 - 25 : lane - 25 + 1;
            dep[wave][lane] = {v[wave][bin].re / 256.0, v[wave][bin].im / 256.0};
        }
    }
    __syncthreads();
    if (tid == 0) {
        c2 mul = {0.0, 0.0};
        for (int s = 0; s < pre_half; ++s)
            for (int i = 0; i < 50; ++i) {
                const c2 d1 = dep[2 * s][i], d2 = dep[2 * s + 1][i];
                const c2 p = cmul({d2.re, -d2.im}, d1);          // conj(frame_depadded2[i]) * frame_depadded1[i]
                mul = {mul.re + p.re, mul.im + p.im};
            }
        double theta = 0;                                         // get_angle, misc.cc:34-56
        if (mul.re == 0) theta = M_PI / 2;
        else if (mul.re > 0) theta = atan(mul.im / mul.re);
        else if (mul.re < 0 && mul.im >= 0) theta = atan(mul.im / mul.re) + M_PI;
        else if (mul.re < 0 && mul.im < 0) theta = atan(mul.im / mul.re) - M_PI;
        freq_out[w] = (theta / M_PI) * carrier_freq_width;
    }
}
