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

#include "glibc_trig.h"

namespace {
struct c2 { double re, im; };
__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
}  // namespace

#define P2B_THREADS 256
#define P2B_MAXTAPS 64

// One block = 256 consecutive outputs of one window. out[k] = sum_j l[n+h-j]*c[j], n = start + k*decim,
// l[i] = in[i]*amp*(cos, sin)(2*pi*fc*i*Ts). The (255*decim + ntaps) mixed samples a block needs are formed once in LDS.
// start_opt (optional) is indexed by the window, like carrier_hz.
// cs (optional): cos / sin per sample index from the host, for launches whose windows all share one carrier — the reference's
// own libm values, so the output is then bit-identical to the reference's, and the device evaluates no trigonometry.
extern "C" __global__ __launch_bounds__(P2B_THREADS) void mgpu_p2b_kernel(
    const double* __restrict__ passband, int in_size, const double* __restrict__ carrier_hz, const int* __restrict__ start_opt,
    int start_all, int count, int decim, const double* __restrict__ taps, int ntaps, double fs, double amplitude,
    double* __restrict__ out, const int* __restrict__ widx, const double* __restrict__ cs, const int* __restrict__ out_row, int row_by_launch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c2* l = reinterpret_cast<c2*>(smem);
    __shared__ double c[P2B_MAXTAPS];
    const int w = widx ? widx[blockIdx.y] : blockIdx.y, tid = threadIdx.x;   // optional subset of the windows
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
            const double a = in[i] * amplitude;
            if (cs) {                // every window of this launch uses the carrier the table was made for (host libm, see api.hip)
                v = {a * cs[2 * i], a * cs[2 * i + 1]};
            } else {
                const double ph = 2 * M_PI * fc * double(i) * Ts;
                double sn, cs1;
                gl_sincos(ph, &sn, &cs1);            // the reference's sincos() call restated (glibc_trig.h): bit-identical mixer
                v = {a * cs1, a * sn};
            }
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
    // output row: the window's own index, or (frames cut out for the RX path) the launch index / an explicit slot
    const int row = out_row ? out_row[blockIdx.y] : (row_by_launch ? int(blockIdx.y) : w);
    out[(size_t(row) * count + k) * 2] = ar;
    out[(size_t(row) * count + k) * 2 + 1] = ai;
}

// The same filter for the reference's 33-tap receive filters with the taps sliding through registers: a lane produces R adjacent outputs
// (D input samples apart), so a mixed sample read from LDS once feeds every output whose window covers it — out[k + r] takes it with tap
// j = (k + r) D + h - i, and as the samples are walked downwards every output meets its taps in ascending order, the reference's order of
// additions. 33 -> (32 + (R - 1) D + 1) / R LDS reads per output (the kernel above spends its time on them: one 16-byte read per two
// multiply-adds); the taps are uniform and sit in scalar registers. Samples outside the input are staged as zeros: a skipped tap and an
// added +-0.0 leave the same sum (the accumulators start at +0.0 and can never become -0.0). Sample p of the block's span lies at
// [p % (R D)][p / (R D)], so the lanes of one read are contiguous and the staging writes conflict-free.
template <int NTAPS, int D, int R, int THREADS>
struct P2sGeom {
    static constexpr int M = R * D;                                  // lane stride in samples = LDS rows
    static constexpr int SPAN = (THREADS * R - 1) * D + NTAPS;       // mixed samples a block needs
    static constexpr int COLS = (SPAN + M - 1) / M + 1;
    static constexpr int ROW = M == 4 ? ((COLS + 13) / 16) * 16 + 2 : (COLS | 1);       // 4 rows: = 2 mod 16; 16 rows: odd
    static constexpr int STEPS = NTAPS + (R - 1) * D;                // samples a lane walks
    static_assert(ROW >= COLS, "row too short");
};
template <int NTAPS, int D, int R, int THREADS, bool TABLE>
__device__ __forceinline__ void p2b_slide(
    const double* __restrict__ passband, int in_size, const double* __restrict__ carrier_hz, const int* __restrict__ start_opt,
    int start_all, int count, const double* __restrict__ taps, double fs, double amplitude,
    double* __restrict__ out, const int* __restrict__ widx, const double* __restrict__ cs, const int* __restrict__ out_row, int row_by_launch, c2* l) {
    using G = P2sGeom<NTAPS, D, R, THREADS>;
    const int w = widx ? widx[blockIdx.y] : blockIdx.y, tid = threadIdx.x;
    const int start = start_opt ? start_opt[w] : start_all;
    const int k0 = blockIdx.x * THREADS * R;
    constexpr int h = (NTAPS - 1) / 2;
    const int base = start + k0 * D + h - (NTAPS - 1);               // input index of the span's sample 0
    const double* in = passband + size_t(w) * in_size;
    const double fc = carrier_hz[w];
    const double Ts = 1.0 / fs;
    for (int t = tid; t < G::SPAN; t += THREADS) {
        const int i = base + t;
        c2 v = {0.0, 0.0};
        if (i >= 0 && i < in_size) {
            const double a = in[i] * amplitude;
            if constexpr (TABLE) {       // the launch's windows share one carrier: cos / sin from the host's table (a kernel of its own: the
                v = {a * cs[2 * i], a * cs[2 * i + 1]};        // inlined sincos would cost this path a third of its wavefronts in registers)
            } else {
                const double ph = 2 * M_PI * fc * double(i) * Ts;
                double sn, cs1;
                gl_sincos(ph, &sn, &cs1);
                v = {a * cs1, a * sn};
            }
        }
        l[(t % G::M) * G::ROW + t / G::M] = v;
    }
    __syncthreads();
    const int k = k0 + tid * R;                                       // lanes past the last output filter staged zeros and store nothing
    double ar[R], ai[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { ar[r] = 0; ai[r] = 0; }
    // the lane's samples are span positions tid * M + q, q = 0 .. STEPS - 1; output r meets position q with tap j = r D + NTAPS - 1 - q
    // in groups of four samples, the next group's reads in flight during the current group's multiply-adds. Left to itself the compiler
    // hoists all 36 reads to the top (150 registers) and then runs the outputs one after another as two dependent chains; the empty asm
    // ties the lane's LDS position and the accumulators to a point in the program, which pins one group's reads and one group's
    // arithmetic between two such points.
    constexpr int GRP = 4, NG = (G::STEPS + GRP - 1) / GRP;
    c2 cur[GRP], nxt[GRP];
    int pos = tid;
    auto fetch = [&](int g, c2* dst) {
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int q = G::STEPS - 1 - (g * GRP + u);
            if (q >= 0) dst[u] = l[(q % G::M) * G::ROW + pos + q / G::M];
        }
    };
    fetch(0, cur);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if constexpr (R == 4)
            asm volatile("" : "+v"(pos), "+v"(ar[0]), "+v"(ai[0]), "+v"(ar[1]), "+v"(ai[1]), "+v"(ar[2]), "+v"(ai[2]), "+v"(ar[3]), "+v"(ai[3]));
        else if constexpr (R == 2)
            asm volatile("" : "+v"(pos), "+v"(ar[0]), "+v"(ai[0]), "+v"(ar[1]), "+v"(ai[1]));
        if (g + 1 < NG) fetch(g + 1, nxt);
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int q = G::STEPS - 1 - (g * GRP + u);
            if (q < 0) continue;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int j = r * D + NTAPS - 1 - q;
                if (j >= 0 && j < NTAPS) { ar[r] += cur[u].re * taps[j]; ai[r] += cur[u].im * taps[j]; }
            }
        }
#pragma unroll
        for (int u = 0; u < GRP; ++u) cur[u] = nxt[u];
    }
    const int row = out_row ? out_row[blockIdx.y] : (row_by_launch ? int(blockIdx.y) : w);
    c2* o = reinterpret_cast<c2*>(out) + size_t(row) * count;
    if constexpr (D == 1) {
        // Write-out through LDS: a lane holds R adjacent outputs, so a direct store instruction would touch a 16-byte piece of every
        // R-th element; in output order a wavefront's stores are contiguous kilobytes (0.21 -> 0.16 ms per 256 windows: two thirds of
        // the kernel's traffic are these stores). A wavefront's outputs are its own 64 R elements of the block, so once every wavefront
        // is done with the sample tile it only has to wait for itself. (At decimation 4, two outputs per lane, the round trip costs more
        // than it saves: 0.051 -> 0.060 ms.)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) l[tid * R + r] = {ar[r], ai[r]};
        __builtin_amdgcn_wave_barrier();
        const int wave0 = (tid & ~63) * R, lane = tid & 63;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int e = wave0 + j * 64 + lane;
            if (k0 + e < count) o[k0 + e] = l[e];
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (k + r < count) o[k + r] = {ar[r], ai[r]};
    }
}

#define P2S_ARGS                                                                                                                              \
    const double *__restrict__ passband, int in_size, const double *__restrict__ carrier_hz, const int *__restrict__ start_opt, int start_all, \
        int count, const double *__restrict__ taps, double fs, double amplitude, double *__restrict__ out, const int *__restrict__ widx,        \
        const double *__restrict__ cs, const int *__restrict__ out_row, int row_by_launch
// decimation 1 (the whole capture window at the interpolated rate): 256 lanes x 4 outputs; decimation 4 (frames cut out at their delay):
// 128 lanes x 2 outputs (mixing the 4 input samples per output is most of that kernel: the block keeps 1053 samples for 256 outputs)
#define P2S_KERNEL(name, D, R, THREADS, TABLE)                                                                                              \
    extern "C" __global__ __launch_bounds__(THREADS) void name(P2S_ARGS) {                                                                 \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                                                               \
        p2b_slide<33, D, R, THREADS, TABLE>(passband, in_size, carrier_hz, start_opt, start_all, count, taps, fs, amplitude, out, widx, cs, \
                                            out_row, row_by_launch, reinterpret_cast<c2*>(smem));                                          \
    }
P2S_KERNEL(mgpu_p2b_slide_d1_kernel, 1, 4, 256, true)
P2S_KERNEL(mgpu_p2b_slide_d4_kernel, 4, 2, 128, true)
P2S_KERNEL(mgpu_p2b_slide_d1_sincos_kernel, 1, 4, 256, false)
P2S_KERNEL(mgpu_p2b_slide_d4_sincos_kernel, 4, 2, 128, false)
extern "C" int mgpu_p2b_slide_geometry(int* o) {   // per kernel: outputs per block, threads, LDS bytes
    o[0] = 256 * 4; o[1] = 256; o[2] = P2sGeom<33, 1, 4, 256>::M * P2sGeom<33, 1, 4, 256>::ROW * 16;
    o[3] = 128 * 2; o[4] = 128; o[5] = P2sGeom<33, 4, 2, 128>::M * P2sGeom<33, 4, 2, 128>::ROW * 16;
    return 0;
}

// Schmidl-Cox metric (ofdm.cc:1893-1941). One lane per candidate offset i = cand*step; the three accumulators run
// in the reference's order (a dependent chain of 2*(Ngi+Nfft/2)*Nsymb additions each), so the parallelism is
// across candidates: a wavefront owns 64 consecutive candidates and walks the preamble in chunks of TS_CH pairs.
// Work per candidate is fixed (2304 sample pairs, 12 fp64 operations each); what has to be engineered is the
// operand delivery, because lanes of a wave are `step` samples apart.
//
//  * mgpu_tsync_metric_kernel (step >= 5, the coarse search uses 100): a direct load would touch 64 cache lines
//    per instruction. Instead TS_CH lanes fetch one candidate's TS_CH-sample row, so a load instruction
//    covers 64/TS_CH rows; the rows are transposed through a small wave-private LDS tile (row stride TS_CH+1
//    elements: conflict-free for the per-lane reads). The next chunk's rows are requested before the current
//    chunk is consumed. 10 KB of LDS per wave.
//  * mgpu_tsync_metric_dense_kernel (step <= 4, the fine search uses 1): the 64 candidates overlap almost
//    completely, so the wave stages the contiguous span 63*step + TS_DCH samples once (coalesced) and every lane
//    reads its own offset from LDS.
#define TS_CH 8
#define TS_RPL (64 / TS_CH)     // candidate rows covered by one wave-wide load
#define TS_NLD (64 / TS_RPL)    // loads per array and chunk
#define TS_WAVES 2           // coarse kernel: wavefronts per workgroup
#define TS_DWAVES 4          // dense kernel

namespace {
typedef double v2d __attribute__((ext_vector_type(2)));   // one complex sample as a native 16-byte vector (keeps prefetch arrays in registers)
struct TsSeg { int a_off, b_off, len; };
// segment q of the reference's loops: per preamble symbol l, (a = l*sym, b = a + Nfft, len Ngi) then
// (a = l*sym + Ngi, b = a + Nfft/2, len Nfft/2)
__device__ __forceinline__ TsSeg ts_segment(int q, int ngi_i, int nfft_i) {
    const int l = q >> 1, part = q & 1, sym = ngi_i + nfft_i;
    const int a_off = l * sym + (part ? ngi_i : 0);
    return {a_off, a_off + (part ? nfft_i / 2 : nfft_i), part ? nfft_i / 2 : ngi_i};
}
// the chunk after (q, m0) in the walk over the 2*pre_nsymb segments; the last chunk is its own successor, so the
// software pipeline can always prefetch without a branch
__device__ __forceinline__ void ts_next_chunk(int q, int m0, int ch, int nseg, int ngi_i, int nfft_i, int& qn, int& mn) {
    const int len = (q & 1) ? nfft_i / 2 : ngi_i;
    qn = q; mn = m0 + ch;
    if (mn >= len) { ++qn; mn = 0; }
    if (qn >= nseg) { qn = q; mn = m0; }
}
__device__ __forceinline__ void ts_accumulate(const c2& x, const c2& y, double& cc, double& na, double& nb) {
    cc += x.re * y.re; na += x.re * x.re; nb += y.re * y.re;
    cc += x.im * y.im; na += x.im * x.im; nb += y.im * y.im;
}
}  // namespace

extern "C" __global__ __launch_bounds__(64 * TS_WAVES) void mgpu_tsync_metric_kernel(
    const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
    const int* __restrict__ ncand_w, int ncand_max, int step, int pre_nsymb, int ngi_i, int nfft_i, double* __restrict__ vals) {
    // window k of the launch is window widx[k] (or k) of the buffer, searched from sample start[k] (or 0) with ncand_w[k]
    // (or ncand_max) candidates; vals is [gridDim.y][ncand_max]
    const int wsel = widx ? widx[blockIdx.x] : blockIdx.x;
    const int wstart = start ? start[blockIdx.x] : 0;
    const int ncand = ncand_w ? ncand_w[blockIdx.x] : ncand_max;
    const int size = stride - wstart;
    // requires ngi_i % TS_CH == 0 and (nfft_i / 2) % TS_CH == 0 (the host checks; 64 and 512 in the reference's calls)
    __shared__ __attribute__((aligned(16))) c2 tile[TS_WAVES][2][64][TS_CH + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cand0 = (blockIdx.y * TS_WAVES + wave) * 64;
    if (cand0 >= ncand) return;
    const int cand = cand0 + lane;
    const char* win = reinterpret_cast<const char*>(bb) + (size_t(wsel) * stride + wstart) * 16;
    c2 (*ra)[TS_CH + 1] = tile[wave][0];
    c2 (*rb)[TS_CH + 1] = tile[wave][1];
    const int rsub = lane / TS_CH, col = lane % TS_CH;           // staging role: row rr*TS_RPL + rsub, sample col
    // byte offset of this lane's element in each of its rows; rows past the last candidate re-read the last one
    // (their lanes' results are discarded), so no load needs a predicate
    unsigned rowbyte[TS_NLD];
#pragma unroll
    for (int rr = 0; rr < TS_NLD; ++rr)
        rowbyte[rr] = (unsigned(min(cand0 + rr * TS_RPL + rsub, ncand - 1)) * unsigned(step) + unsigned(col)) * 16u;
    const int nseg = 2 * pre_nsymb;
    v2d pa[TS_NLD], pb[TS_NLD];                                  // rows in flight for the next chunk
    double cc = 0, na = 0, nb = 0;
    int q = 0, m0 = 0;
    {
        const TsSeg sg = ts_segment(0, ngi_i, nfft_i);
#pragma unroll
        for (int rr = 0; rr < TS_NLD; ++rr) {
            pa[rr] = *reinterpret_cast<const v2d*>(win + size_t(sg.a_off) * 16 + rowbyte[rr]);
            pb[rr] = *reinterpret_cast<const v2d*>(win + size_t(sg.b_off) * 16 + rowbyte[rr]);
        }
    }
    const int nchunks = pre_nsymb * (ngi_i + nfft_i / 2) / TS_CH;
    for (int it = 0; it < nchunks; ++it) {
#pragma unroll
        for (int rr = 0; rr < TS_NLD; ++rr) {
            ra[rr * TS_RPL + rsub][col] = {pa[rr].x, pa[rr].y};
            rb[rr * TS_RPL + rsub][col] = {pb[rr].x, pb[rr].y};
        }
        int qn, mn;
        ts_next_chunk(q, m0, TS_CH, nseg, ngi_i, nfft_i, qn, mn);
        {
            const TsSeg sg = ts_segment(qn, ngi_i, nfft_i);
            const char* abase = win + size_t(sg.a_off + mn) * 16;    // wave-uniform
            const char* bbase = win + size_t(sg.b_off + mn) * 16;
#pragma unroll
            for (int rr = 0; rr < TS_NLD; ++rr) {
                pa[rr] = *reinterpret_cast<const v2d*>(abase + rowbyte[rr]);
                pb[rr] = *reinterpret_cast<const v2d*>(bbase + rowbyte[rr]);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int m = 0; m < TS_CH; ++m) ts_accumulate(ra[lane][m], rb[lane][m], cc, na, nb);
        __builtin_amdgcn_wave_barrier();
        q = qn; m0 = mn;
    }
    if (cand >= ncand) return;
    if (na < 0.001 || nb < 0.001) cc = 0.0;
    else cc = cc / sqrt(na * nb);
    vals[size_t(blockIdx.x) * ncand_max + cand] = cc;
}

// Fallback for segment lengths that are not multiples of TS_CH: one lane per candidate, direct loads.
extern "C" __global__ __launch_bounds__(64) void mgpu_tsync_metric_generic_kernel(
    const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
    const int* __restrict__ ncand_w, int ncand_max, int step, int pre_nsymb, int ngi_i, int nfft_i, double* __restrict__ vals) {
    // window k of the launch is window widx[k] (or k) of the buffer, searched from sample start[k] (or 0) with ncand_w[k]
    // (or ncand_max) candidates; vals is [gridDim.y][ncand_max]
    const int wsel = widx ? widx[blockIdx.y] : blockIdx.y;
    const int wstart = start ? start[blockIdx.y] : 0;
    const int ncand = ncand_w ? ncand_w[blockIdx.y] : ncand_max;
    const int size = stride - wstart;
    const int cand = blockIdx.x * 64 + threadIdx.x;
    if (cand >= ncand) return;
    const c2* data = reinterpret_cast<const c2*>(bb) + size_t(wsel) * stride + wstart + size_t(cand) * step;
    double cc = 0, na = 0, nb = 0;
    for (int q = 0; q < 2 * pre_nsymb; ++q) {
        const TsSeg sg = ts_segment(q, ngi_i, nfft_i);
        for (int m = 0; m < sg.len; ++m) ts_accumulate(data[sg.a_off + m], data[sg.b_off + m], cc, na, nb);
    }
    if (na < 0.001 || nb < 0.001) cc = 0.0;
    else cc = cc / sqrt(na * nb);
    vals[size_t(blockIdx.y) * ncand_max + cand] = cc;
}

// ---- streaming coarse search (many windows per launch) ----
// The staged kernel above fetches every sample once per candidate that overlaps it (~44 times: 70 GB of L2 traffic per 1024
// windows, which is what bounds it). Here one wavefront walks a window once. Lane j of the first J lanes holds candidate
// q (q mod J == j), and candidate q starts K pair-steps (one "period") after candidate q-1. K is chosen so that the samples a
// candidate consumes per period (~ K * sym / pairs-per-symbol) match the candidate spacing `step`: all candidates in flight then
// read within one sliding span of < 2048 samples around step*q, which lives in a wave-private LDS ring that is refilled with 1-2
// coalesced 64-sample blocks per period. Every sample comes from L2/HBM once per wave; each candidate's three sums still run in
// the reference's order. Rows of 16 samples are padded to 17 so that lanes 48 samples (3 rows) apart read distinct banks.
//
// With one wavefront per SIMD (a ring is 35 KB) the kernel runs at the speed of its instruction stream, so the bookkeeping is
// kept off the vector unit: the geometry is a template parameter, which makes the places where a candidate of age e (periods
// since its start) crosses from the guard-interval pairs to the half-symbol pairs or into the next preamble symbol compile-time
// constants per chunk; the one or two lanes concerned are found with scalar arithmetic and patched under the exec mask. Per chunk
// of 4 pairs a lane spends 48 fp64 operations, two position increments and two ring-slot computations.
#define TSS_K 52
#define TSS_CHUNKS (TSS_K / 4)
#define TSS_RING 2048
#define TSS_RING_BYTES ((TSS_RING / 16) * 17 * 16)

namespace {
__device__ __forceinline__ unsigned tss_slot(unsigned p) { const unsigned r = p & (TSS_RING - 1); return (r + (r >> 4)) << 4; }

template <int STEP, int NGI_, int NFFT_, int PRE_>
struct TssGeom {
    static constexpr int STEPV = STEP, NGI = NGI_, NFFT = NFFT_, PRE = PRE_, SYM = NGI_ + NFFT_, HALF = NFFT_ / 2, PS = NGI_ + HALF, NP = PRE_ * PS;
    static constexpr int J = (NP + TSS_K - 1) / TSS_K;                   // candidates in flight
    static constexpr int FZP = NP / TSS_K, FZC = (NP % TSS_K) / 4;       // a candidate is complete before chunk FZC of period (its start + FZP)
    static_assert(STEP % 4 == 0 && NGI % 4 == 0 && HALF % 4 == 0 && J <= 64 && STEP <= 128, "geometry");
    static_assert(TSS_K == 4 * ((STEP * PS) / (4 * SYM)), "the skew must track the candidate spacing");
};

// segment changes of the candidates in flight before the chunk with index ch of a period whose starting lane is js
template <class G>
__device__ __forceinline__ void tss_segment_events(int ch, int js, int lane, int& pa, int& pb) {
#pragma unroll
    for (int e = 0; e < G::J; ++e) {
        const int n = TSS_K * e + 4 * ch;                            // pair index of the candidate of age e at this chunk
        const int k = n % G::PS;
        if (n > 0 && n < G::NP && (k == 0 || k == G::NGI)) {
            const int le = js - e < 0 ? js - e + G::J : js - e;     // scalar
            if (k == 0) { if (lane == le) { pa += G::SYM - G::PS; pb += G::SYM - G::PS + G::NFFT - G::HALF; } }
            else if (lane == le) pb -= G::NFFT - G::HALF;
        }
    }
}
}  // namespace

// the geometry the kernel is built for: the reference's coarse search (step 100, ofdm.cc:1893) on the x4 interpolated baseband of its
// Nfft 256 / guard interval 1/16 / 4-symbol-preamble modes; anything else goes to the staged kernel
typedef TssGeom<100, 64, 1024, 4> TssCoarse;
extern "C" void mgpu_tsync_stream_geometry(int* g) { g[0] = TssCoarse::STEPV; g[1] = TssCoarse::NGI; g[2] = TssCoarse::NFFT; g[3] = TssCoarse::PRE; g[4] = TSS_K; g[5] = TSS_RING; }

template <class G>
__device__ __forceinline__ void tss_run(const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
                                        const int* __restrict__ ncand_w, int ncand_max, double* __restrict__ vals, int lo, int hi, int cpp, char* ring) {
    // lo / hi: the span [step*q + lo, step*q + hi) holds everything the candidates in flight read during the period that candidate
    // q starts in (the host enumerates it); cpp: candidates per workgroup (blockIdx.y = piece of the candidate range)
    const int wsel = widx ? widx[blockIdx.x] : blockIdx.x;
    const int wstart = start ? start[blockIdx.x] : 0;
    const int ncand = ncand_w ? ncand_w[blockIdx.x] : ncand_max;
    const int c0 = blockIdx.y * cpp;
    if (c0 >= ncand) return;
    const int nq = min(cpp, ncand - c0);
    const int lane = threadIdx.x;
    const int last = stride - wstart - 1;
    const v2d* winv = reinterpret_cast<const v2d*>(bb) + (size_t(wsel) * stride + wstart);
    constexpr int step = G::STEPV;
    int loaded = max(step * c0 + lo, 0) & ~63;
    for (const int need = step * c0 + hi; loaded < need; loaded += 64)
        *reinterpret_cast<v2d*>(ring + tss_slot(unsigned(loaded + lane))) = winv[min(loaded + lane, last)];
    const int nperiods = (nq + G::FZP + (G::FZC ? 1 : 0) + 1) & ~1;  // even: the two register sets below swap roles every chunk, 13 chunks per period
    int pa = step * c0, pb = pa + G::NFFT;                           // every lane starts somewhere inside the ring; lane 0 is candidate 0
    double cc = 0, na = 0, nb = 0;
    int js = 0, jf = 0;                                              // lanes of the candidate starting / finishing in this period
    // The reads of chunk i+1 are issued before chunk i is accumulated (one wavefront per SIMD: nothing else hides the LDS latency).
    c2 xa[2][4], xb[2][4];
    auto fetch = [&](c2* da, c2* db) {
        const c2* qa = reinterpret_cast<const c2*>(ring + tss_slot(unsigned(pa)));
        const c2* qb = reinterpret_cast<const c2*>(ring + tss_slot(unsigned(pb)));
#pragma unroll
        for (int m = 0; m < 4; ++m) { da[m] = qa[m]; db[m] = qb[m]; }
        pa += 4; pb += 4;
    };
    auto period = [&](int p, auto parity) {
        constexpr int par = decltype(parity)::value;
        // request what the next period needs beyond `loaded`; it goes into the ring once this period's last chunk has been read
        const int nblk = (step * (c0 + p + 1) + hi - loaded + 63) >> 6;            // 0..2
        v2d pf0 = {0, 0}, pf1 = {0, 0};
        if (nblk > 0) pf0 = winv[min(loaded + lane, last)];
        if (nblk > 1) pf1 = winv[min(loaded + 64 + lane, last)];
        const int jz = js;                                           // the lane whose new candidate begins with this period's chunk 0
#pragma unroll
        for (int ch = 0; ch < TSS_CHUNKS; ++ch) {
            const int cur = (ch + par) & 1, nxt = cur ^ 1;
            if (ch == TSS_CHUNKS - 1) {
                if (nblk > 0) *reinterpret_cast<v2d*>(ring + tss_slot(unsigned(loaded + lane))) = pf0;
                if (nblk > 1) *reinterpret_cast<v2d*>(ring + tss_slot(unsigned(loaded + 64 + lane))) = pf1;
                loaded += nblk * 64;
                js = js + 1 == G::J ? 0 : js + 1;
                if (lane == js) { pa = step * (c0 + p + 1); pb = pa + G::NFFT; }    // next period's starter
                tss_segment_events<G>(0, js, lane, pa, pb);
            } else {
                tss_segment_events<G>(ch + 1, js, lane, pa, pb);
            }
            // the reads of the next chunk go between the previous chunk's arithmetic and this chunk's: tying their addresses to the sums keeps
            // the optimiser from collecting a period's reads ahead of its arithmetic (13 chunks of operands do not fit the registers)
            asm volatile("" : "+v"(pa), "+v"(pb), "+v"(cc), "+v"(na), "+v"(nb));
            fetch(xa[nxt], xb[nxt]);
            __builtin_amdgcn_sched_barrier(0);
            if (ch == G::FZC && p >= G::FZP) {
                const int q = p - G::FZP;
                if (q < nq && lane == jf) {
                    double r = 0.0;
                    if (!(na < 0.001 || nb < 0.001)) r = cc / sqrt(na * nb);
                    vals[size_t(blockIdx.x) * ncand_max + c0 + q] = r;
                }
                jf = jf + 1 == G::J ? 0 : jf + 1;
            }
            if (ch == 0 && lane == jz) { cc = 0; na = 0; nb = 0; }
#pragma unroll
            for (int m = 0; m < 4; ++m) ts_accumulate(xa[cur][m], xb[cur][m], cc, na, nb);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    fetch(xa[0], xb[0]);
    for (int p = 0; p < nperiods; p += 2) {
        period(p, std::integral_constant<int, 0>());
        period(p + 1, std::integral_constant<int, 1>());
    }
}

extern "C" __global__ __launch_bounds__(64) void mgpu_tsync_metric_stream_kernel(
    const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
    const int* __restrict__ ncand_w, int ncand_max, double* __restrict__ vals, int lo, int hi, int cpp) {
    __shared__ __attribute__((aligned(16))) char ring[TSS_RING_BYTES];
    tss_run<TssCoarse>(bb, stride, start, widx, ncand_w, ncand_max, vals, lo, hi, cpp, ring);
}

#define TS_DCH 64
#define TS_DSPAN (63 * 4 + TS_DCH)      // step <= 4

extern "C" int mgpu_tsync_coarse_threads() { return 64 * TS_WAVES; }

extern "C" __global__ __launch_bounds__(64 * TS_DWAVES) void mgpu_tsync_metric_dense_kernel(
    const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
    const int* __restrict__ ncand_w, int ncand_max, int step, int pre_nsymb, int ngi_i, int nfft_i, double* __restrict__ vals) {
    // window k of the launch is window widx[k] (or k) of the buffer, searched from sample start[k] (or 0) with ncand_w[k]
    // (or ncand_max) candidates; vals is [gridDim.y][ncand_max]
    const int wsel = widx ? widx[blockIdx.y] : blockIdx.y;
    const int wstart = start ? start[blockIdx.y] : 0;
    const int ncand = ncand_w ? ncand_w[blockIdx.y] : ncand_max;
    const int size = stride - wstart;
    __shared__ __attribute__((aligned(16))) c2 span[TS_DWAVES][2][TS_DSPAN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cand0 = (blockIdx.x * TS_DWAVES + wave) * 64;
    if (cand0 >= ncand) return;
    const int cand = cand0 + lane;
    const c2* win = reinterpret_cast<const c2*>(bb) + size_t(wsel) * stride + wstart + size_t(cand0) * step;
    const long avail = long(size) - long(cand0) * step;         // samples from the wave's first candidate to the end
    c2* sa = span[wave][0];
    c2* sb = span[wave][1];
    const int nseg = 2 * pre_nsymb;
    const int width = 63 * step + TS_DCH;                        // <= TS_DSPAN
    constexpr int kLoads = (TS_DSPAN + 63) / 64;
    const int nld = (width + 63) / 64;                           // 2 for step 1
    const int last = int(avail) - 1;                             // clamp instead of predicating: clamped elements feed discarded lanes only
    v2d pa[kLoads], pb[kLoads];
    const v2d* winv = reinterpret_cast<const v2d*>(win);
    double cc = 0, na = 0, nb = 0;
    int q = 0, m0 = 0;
    {
        const TsSeg sg = ts_segment(0, ngi_i, nfft_i);
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int t = j * 64 + lane;
            pa[j] = winv[min(sg.a_off + t, last)];
            pb[j] = winv[min(sg.b_off + t, last)];
        }
    }
    const int nchunks = pre_nsymb * (ngi_i + nfft_i / 2) / TS_DCH;
    for (int it = 0; it < nchunks; ++it) {
#pragma unroll
        for (int j = 0; j < kLoads; ++j) {
            const int t = j * 64 + lane;
            if (j < nld && t < TS_DSPAN) { sa[t] = {pa[j].x, pa[j].y}; sb[t] = {pb[j].x, pb[j].y}; }
        }
        int qn, mn;
        ts_next_chunk(q, m0, TS_DCH, nseg, ngi_i, nfft_i, qn, mn);
        {
            const TsSeg sg = ts_segment(qn, ngi_i, nfft_i);
#pragma unroll
            for (int j = 0; j < kLoads; ++j) {
                if (j < nld) {
                    const int t = j * 64 + lane;
                    pa[j] = winv[min(sg.a_off + mn + t, last)];
                    pb[j] = winv[min(sg.b_off + mn + t, last)];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const c2* la = sa + lane * step;
        const c2* lb = sb + lane * step;
#pragma unroll 8
        for (int m = 0; m < TS_DCH; ++m) ts_accumulate(la[m], lb[m], cc, na, nb);
        __builtin_amdgcn_wave_barrier();
        q = qn; m0 = mn;
    }
    if (cand >= ncand) return;
    if (na < 0.001 || nb < 0.001) cc = 0.0;
    else cc = cc / sqrt(na * nb);
    vals[size_t(blockIdx.y) * ncand_max + cand] = cc;
}

// The fine search (step 1, ofdm.cc:1893-1941 called from telecom_system.cc:1014-1018 over (preamble + 4) symbols = 4352 candidates of
// 2304 sample pairs): candidates c and c + 1 read the same samples one pair apart — the pair (x, y) candidate c uses at position m of
// a segment is (win[a_off + c + m], win[b_off + c + m]) — so the six products a pair contributes (x.re*y.re, x.re*x.re, y.re*y.re and
// the .im ones) depend on c + m only. A lane therefore owns R *adjacent* candidates and slides a window of R samples' products along
// the segment: per step one new sample (two 16-byte LDS reads, six multiplications) feeds R candidates (6 additions each, every
// accumulator in the reference's order). 12 -> 6 + 6/R fp64 operations and 2 -> 2/R LDS reads per candidate pair; the dense kernel
// above was bound by its LDS reads (two ds_read_b128 per 12 operations: 64 LDS cycles per 48 issue cycles and compute unit).
// LDS layout: sample i of the wave's span (relative to its first candidate) lies at [i % R][i / R], so the lanes of one read are
// contiguous; the row length makes the coalesced staging writes (8 consecutive samples = 8 / R columns of every row) conflict-free.
template <int R>
struct TfGeom {
    static constexpr int LOADS = R + 1;                                   // 64 R + 63 samples per chunk of 64 pairs
    static constexpr int COLS = 64 + 64 / R + 1;
    static constexpr int ROW = R == 4 ? 82 : 73;                          // >= COLS; = 2 (R = 4) / odd (R = 8) mod 16 columns of 16 bytes
    static constexpr int WAVES = 1;                                       // nothing is shared between wavefronts: one per workgroup, so the 17 of a 4352-candidate search leave no slot idle
    static_assert(R == 4 || R == 8, "row length chosen for 4 or 8 candidates per lane");
    static_assert(ROW >= COLS, "row too short");
};

template <int R>
__device__ __forceinline__ void tfine_run(const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
                                          const int* __restrict__ ncand_w, int ncand_max, int pre_nsymb, int ngi_i, int nfft_i,
                                          double* __restrict__ vals, c2* lds) {
    using G = TfGeom<R>;
    const int wsel = widx ? widx[blockIdx.y] : blockIdx.y;
    const int wstart = start ? start[blockIdx.y] : 0;
    const int ncand = ncand_w ? ncand_w[blockIdx.y] : ncand_max;
    const int size = stride - wstart;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cand0 = (blockIdx.x * G::WAVES + wave) * 64 * R;
    if (cand0 >= ncand) return;
    const v2d* winv = reinterpret_cast<const v2d*>(bb) + size_t(wsel) * stride + wstart + cand0;
    const int last = size - cand0 - 1;                                    // clamp instead of predicating: clamped samples feed discarded candidates only
    c2* sa = lds + size_t(wave) * 2 * R * G::ROW;
    c2* sb = sa + R * G::ROW;
    const int wr = (lane % R) * G::ROW + lane / R;                        // staging: sample j*64 + lane -> row lane % R, column j*(64/R) + lane/R
    const int nseg = 2 * pre_nsymb;
    v2d pa[G::LOADS], pb[G::LOADS];
    double cc[R], na[R], nb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { cc[r] = 0; na[r] = 0; nb[r] = 0; }
    int q = 0, m0 = 0;
    {
        const TsSeg sg = ts_segment(0, ngi_i, nfft_i);
#pragma unroll
        for (int j = 0; j < G::LOADS; ++j) {
            const int t = j * 64 + lane;
            pa[j] = winv[min(sg.a_off + t, last)];
            pb[j] = winv[min(sg.b_off + t, last)];
        }
    }
    const int nchunks = pre_nsymb * (ngi_i + nfft_i / 2) / 64;
    for (int it = 0; it < nchunks; ++it) {
#pragma unroll
        for (int j = 0; j < G::LOADS; ++j) {
            sa[wr + j * (64 / R)] = {pa[j].x, pa[j].y};
            sb[wr + j * (64 / R)] = {pb[j].x, pb[j].y};
        }
        int qn, mn;
        ts_next_chunk(q, m0, 64, nseg, ngi_i, nfft_i, qn, mn);
        {
            const TsSeg sg = ts_segment(qn, ngi_i, nfft_i);
#pragma unroll
            for (int j = 0; j < G::LOADS; ++j) {
                const int t = j * 64 + lane;
                pa[j] = winv[min(sg.a_off + mn + t, last)];
                pb[j] = winv[min(sg.b_off + mn + t, last)];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // products of the window's samples: slot k % R holds sample R*lane + k
        double prr[R], pii[R], xrr[R], xii[R], yrr[R], yii[R];
        auto rd = [&](int row, int col, c2& x, c2& y) { x = sa[row * G::ROW + lane + col]; y = sb[row * G::ROW + lane + col]; };
        auto prod = [&](int slot, const c2& x, const c2& y) {
            prr[slot] = x.re * y.re; xrr[slot] = x.re * x.re; yrr[slot] = y.re * y.re;
            pii[slot] = x.im * y.im; xii[slot] = x.im * x.im; yii[slot] = y.im * y.im;
        };
        c2 nx, ny;                                                        // the sample one step ahead: its LDS read is in flight during the additions
#pragma unroll
        for (int k = 0; k < R - 1; ++k) { rd(k, 0, nx, ny); prod(k, nx, ny); }
        rd(R - 1, 0, nx, ny);
        for (int u = 0; u < 64 / R; ++u) {
#pragma unroll
            for (int t = 0; t < R; ++t) {
                // pair m = R u + t: the new sample is k = m + R - 1 (slot (t + R - 1) % R); the one after it lies at row t, column lane + u + 1
                // (the last step of a chunk reads one column past the span: inside the wave's rows, never used)
                prod((t + R - 1) % R, nx, ny);
                rd(t, u + 1, nx, ny);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int sl = (t + r) % R;
                    cc[r] += prr[sl]; na[r] += xrr[sl]; nb[r] += yrr[sl];
                    cc[r] += pii[sl]; na[r] += xii[sl]; nb[r] += yii[sl];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        q = qn; m0 = mn;
    }
    double* out = vals + size_t(blockIdx.y) * ncand_max;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int cand = cand0 + R * lane + r;
        if (cand < ncand) out[cand] = (na[r] < 0.001 || nb[r] < 0.001) ? 0.0 : cc[r] / sqrt(na[r] * nb[r]);
    }
}

extern "C" int mgpu_tsync_fine_geometry(int* out) {   // candidates per workgroup, threads, LDS bytes for R = 4 and R = 8
    out[0] = 64 * 4 * TfGeom<4>::WAVES; out[1] = 64 * TfGeom<4>::WAVES; out[2] = TfGeom<4>::WAVES * 2 * 4 * TfGeom<4>::ROW * 16;
    out[3] = 64 * 8 * TfGeom<8>::WAVES; out[4] = 64 * TfGeom<8>::WAVES; out[5] = TfGeom<8>::WAVES * 2 * 8 * TfGeom<8>::ROW * 16;
    return 0;
}

extern "C" __global__ __launch_bounds__(64 * TfGeom<4>::WAVES) void mgpu_tsync_metric_fine_kernel_r4(
    const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
    const int* __restrict__ ncand_w, int ncand_max, int pre_nsymb, int ngi_i, int nfft_i, double* __restrict__ vals) {
    extern __shared__ __attribute__((aligned(16))) char tf_lds[];
    tfine_run<4>(bb, stride, start, widx, ncand_w, ncand_max, pre_nsymb, ngi_i, nfft_i, vals, reinterpret_cast<c2*>(tf_lds));
}

extern "C" __global__ __launch_bounds__(64 * TfGeom<8>::WAVES) void mgpu_tsync_metric_fine_kernel_r8(
    const double* __restrict__ bb, int stride, const int* __restrict__ start, const int* __restrict__ widx,
    const int* __restrict__ ncand_w, int ncand_max, int pre_nsymb, int ngi_i, int nfft_i, double* __restrict__ vals) {
    extern __shared__ __attribute__((aligned(16))) char tf_lds[];
    tfine_run<8>(bb, stride, start, widx, ncand_w, ncand_max, pre_nsymb, ngi_i, nfft_i, vals, reinterpret_cast<c2*>(tf_lds));
}

// Moose: pre_half preamble symbols (up to 4: preamble_nSymb / 2 for preambles of up to 8 symbols), each as two 256-point FFTs of a
// half symbol repeated twice; the four wavefronts take two symbols per round.
extern "C" __global__ __launch_bounds__(256) void mgpu_fsync_kernel(
    const double* __restrict__ bb, int stride, int pre_half, const double* __restrict__ twiddle, double* __restrict__ freq_out) {
    __shared__ __attribute__((aligned(16))) c2 v[4][256];
    __shared__ __attribute__((aligned(16))) c2 tw[128];      // 16-byte aligned: fft256_twiddle reads it as ds_read_b128
    __shared__ __attribute__((aligned(16))) c2 dep[8][50];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const c2* in = reinterpret_cast<const c2*>(bb) + size_t(w) * stride;
    if (tid < 128) tw[tid] = {twiddle[2 * tid], twiddle[2 * tid + 1]};
    __syncthreads();
    for (int j0 = 0; j0 < pre_half; j0 += 2) {
    const int j = j0 + (wave >> 1), halfsel = wave & 1;          // wave -> (symbol j, first/second half)
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
            dep[2 * j + halfsel][lane] = {v[wave][bin].re / 256.0, v[wave][bin].im / 256.0};
        }
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
        // the closing get_angle(mul) / pi * width (misc.cc:34-56, ofdm.cc:594) is one atan per window: the host does it with the
        // reference's libm (moose_hz in ctx.hpp), so the offset is bit-identical; the sum leaves the device as it is
        freq_out[2 * w] = mul.re;
        freq_out[2 * w + 1] = mul.im;
    }
}

// Sums of |x|^2 over short spans of the interpolated baseband: the energy gates of receive_byte
// (telecom_system.cc:758-766, :826-834, :1044-1066, ...). Span j lies in window wv[j] at offset off[j]; the terms
// re*re + im*im are added in sample order for i < len while off + i < stride (the reference's loop bounds); the caller
// divides by the count or by len as the call site does. One wavefront per span: the terms are formed in parallel from
// coalesced loads into LDS, lane 0 adds them in order. len <= SE_MAXLEN.
#define SE_MAXLEN 1088
extern "C" __global__ __launch_bounds__(256) void mgpu_span_energy_kernel(
    const double* __restrict__ bb, int stride, const int* __restrict__ wv, const int* __restrict__ off, int n, int len,
    double* __restrict__ sum, int* __restrict__ cnt) {
    __shared__ double term[4][SE_MAXLEN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= n) return;
    const c2* x = reinterpret_cast<const c2*>(bb) + size_t(wv[j]) * stride;
    const int o = off[j];
    int m = stride - o;                                           // terms that exist: i < len && o + i < stride
    m = m < 0 ? 0 : (m > len ? len : m);
    double* t = term[wave];
    for (int i = lane; i < m; i += 64) { const c2 v = x[o + i]; t[i] = v.re * v.re + v.im * v.im; }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double e = 0.0;
        int p = 0;
        for (; p + 8 <= m; p += 8) {
            const double a0 = t[p], a1 = t[p + 1], a2 = t[p + 2], a3 = t[p + 3], a4 = t[p + 4], a5 = t[p + 5], a6 = t[p + 6], a7 = t[p + 7];
            e += a0; e += a1; e += a2; e += a3; e += a4; e += a5; e += a6; e += a7;
        }
        for (; p < m; ++p) e += t[p];
        sum[j] = e;
        cnt[j] = m;
    }
}

// The same sums for many spans at once (the forward scans of receive_byte's recoveries ask for up to ~80 per window): one *lane* per span.
// A wavefront owns 64 spans; per chunk of 16 samples it fetches 64 rows of 256 contiguous bytes (16 lanes per row, 4 rows per load), forms
// the terms and parks them in a wave-private LDS tile (row stride 17: the per-lane reads are conflict-free), then every lane adds its own
// row's 16 terms in sample order. 64 dependent chains per wavefront instead of one: the wave-per-span kernel above spends its time in a
// single lane's 1088 additions.
#define SEM_CH 16
extern "C" __global__ __launch_bounds__(256) void mgpu_span_energy_many_kernel(
    const double* __restrict__ bb, int stride, const int* __restrict__ wv, const int* __restrict__ off, int n, int len,
    double* __restrict__ sum, int* __restrict__ cnt) {
    __shared__ double tile[4][64][SEM_CH + 1];
    __shared__ size_t base_s[4][64];
    __shared__ int m_s[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j0 = (blockIdx.x * 4 + wave) * 64;
    if (j0 >= n) return;
    const int j = min(j0 + lane, n - 1);                          // lanes past the last span repeat it (results discarded)
    const int o = off[j];
    int m = stride - o;                                           // terms that exist: i < len && o + i < stride
    m = m < 0 ? 0 : (m > len ? len : m);
    base_s[wave][lane] = size_t(wv[j]) * size_t(stride) + size_t(o);              // complex-sample index of the span's first term
    m_s[wave][lane] = m;
    __builtin_amdgcn_wave_barrier();
    const c2* x = reinterpret_cast<const c2*>(bb);
    const int rsub = lane >> 4, col = lane & 15;                  // staging role: row 4 r + rsub, sample col of the chunk
    size_t rbase[16];
    int rm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { rbase[r] = base_s[wave][4 * r + rsub] + size_t(col); rm[r] = m_s[wave][4 * r + rsub] - col; }
    double (*t)[SEM_CH + 1] = tile[wave];
    int mmax = m;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mmax = max(mmax, __shfl_xor(mmax, d));
    double e = 0.0;
    c2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = rm[r] > 0 ? x[rbase[r]] : c2{0.0, 0.0};
    for (int i0 = 0; i0 < mmax; i0 += SEM_CH) {
#pragma unroll
        for (int r = 0; r < 16; ++r) t[4 * r + rsub][col] = v[r].re * v[r].re + v[r].im * v[r].im;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = rm[r] > i0 + SEM_CH ? x[rbase[r] + size_t(i0 + SEM_CH)] : c2{0.0, 0.0};      // next chunk in flight
        const int left = m - i0;
        if (left >= SEM_CH) {
#pragma unroll
            for (int q = 0; q < SEM_CH; ++q) e += t[lane][q];
        } else {
            for (int q = 0; q < left; ++q) e += t[lane][q];
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (j0 + lane < n) { sum[j0 + lane] = e; cnt[j0 + lane] = m; }
}

// rational_resampler(..., DECIMATION) (ofdm.cc:2267-2278) from a per-window offset:
// out[slot[k] or k][i] = bb[widx[k]][delay[k] + i*rate]
extern "C" __global__ __launch_bounds__(256) void mgpu_decimate_kernel(
    const double* __restrict__ bb, int stride, const int* __restrict__ widx, const int* __restrict__ delay, const int* __restrict__ slot,
    int rate, int count, double* __restrict__ out) {
    const int k = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const c2* x = reinterpret_cast<const c2*>(bb) + size_t(widx[k]) * stride + delay[k];
    reinterpret_cast<c2*>(out)[size_t(slot ? slot[k] : k) * count + i] = x[size_t(i) * rate];
}

// cl_ofdm::measure_signal_stregth (ofdm.cc:1523-1539): sum over a whole window of re^2 + im^2, added in sample order.
// One workgroup per window: the terms of a chunk are formed in parallel from coalesced loads, one lane adds them.
#define WE_CHUNK 512
#define WE_LOADS (WE_CHUNK / 64)
extern "C" __global__ __launch_bounds__(64) void mgpu_window_energy_kernel(
    const double* __restrict__ bb, int stride, int n, double* __restrict__ out) {
    // One wavefront and 4 KB of LDS per window, so that the 92 k-term chain can sit beside whatever else fills the compute units
    // (receive_byte launches it ahead of the coarse search). The next chunk's samples are in flight while the current chunk's terms
    // are added; every lane carries the same sum.
    __shared__ __attribute__((aligned(16))) double term[WE_CHUNK + 32];
    const c2* x = reinterpret_cast<const c2*>(bb) + size_t(blockIdx.x) * stride;
    const int lane = threadIdx.x;
    double acc = 0.0;
    c2 v[WE_LOADS];
#pragma unroll
    for (int r = 0; r < WE_LOADS; ++r) { const int i = r * 64 + lane; v[r] = i < n ? x[i] : c2{0.0, 0.0}; }
    for (int base = 0; base < n; base += WE_CHUNK) {
        const int m = min(WE_CHUNK, n - base);
#pragma unroll
        for (int r = 0; r < WE_LOADS; ++r) term[r * 64 + lane] = v[r].re * v[r].re + v[r].im * v[r].im;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < WE_LOADS; ++r) { const int i = base + WE_CHUNK + r * 64 + lane; v[r] = i < n ? x[i] : c2{0.0, 0.0}; }
        if (m == WE_CHUNK) {
            // The chain. A dependent v_add_f64 can issue every 8.25 cycles (tools/ubench/dep_chain.hip: 3.46 ns); a lone wavefront also
            // spends ~7 cycles of issue on every 16-byte LDS read (two terms), which is what is left to pay: 12 cycles per term when the 16
            // terms of the next group are already on their way while a group is added. Written as a loop the compiler waits for *all*
            // outstanding reads at the loop head (22 cycles per term); as straight-line code it counts them exactly. Every lane adds the
            // same terms (LDS broadcasts): with a single lane enabled a dependent addition takes 10 cycles instead of 8.25. Tried and
            // slower: one read after every two additions (15.8 cycles per term) and terms spread over the lanes with a v_mov_b64_dpp
            // row_newbcast per term (the 64-bit move costs two issue passes: 15.8 again; fp64 additions have no DPP form on gfx9).
            double a[16], b[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] = term[q];
#pragma unroll
            for (int g = 0; g < WE_CHUNK / 32; ++g) {
#pragma unroll
                for (int q = 0; q < 16; ++q) b[q] = term[32 * g + 16 + q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc += a[q];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = term[32 * g + 32 + q];   // the last group reads 16 terms past the chunk: inside the array, never added
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc += b[q];
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int p = 0; p < m; ++p) acc += term[p];                 // the window's last, partial chunk: every lane reads the same word
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) out[blockIdx.x] = acc;
}

// The reference's peak selection (ofdm.cc:1943-1964) on the candidate metrics where they lie in HBM, one wavefront per window:
// vals[k*step] = metric of candidate k, 0 elsewhere; pass j starts from entry j and takes every strictly larger later entry
// (nothing is swapped out); the entry of pass `loc` is returned. A running "strictly larger" maximum ends at the FIRST occurrence
// of the largest entry at or after j, so the scan parallelises: every lane keeps the first maximum of its own (position-ordered)
// share of the candidates, lanes are merged with "larger value, else smaller position", the implicit zeros between / behind the
// candidates are represented by the first of them, and the start entry is kept unless something is strictly larger (which also
// reproduces the reference when that entry is NaN). Same result as select_peak in api.hip (the host emulation the CPU tests pin
// against the reference's loop); one lane per window crawling through 881 or 4352 entries took 0.37 ms per launch.
extern "C" __global__ __launch_bounds__(64) void mgpu_select_peak_kernel(
    const double* __restrict__ vals, const int* __restrict__ ncand_w, int ncand_max, int step, const int* __restrict__ size_w,
    const int* __restrict__ loc_w, int ntrials, int n, int* __restrict__ delay, double* __restrict__ corr) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= n) return;
    const double* v = vals + size_t(k) * ncand_max;
    const int ncand = ncand_w[k], size = size_w[k];
    int j = loc_w[k];
    if (j >= ntrials) j = ntrials - 1;
    const double ninf = -__builtin_inf();
    double best = ninf;                                             // "nothing yet": no entry is strictly smaller
    int pos = 0x7fffffff;
    for (int c = (j + step) / step + lane; c < ncand; c += 64) {    // candidates behind entry j, in position order per lane
        const double x = v[c];
        if (x > best) { best = x; pos = c * step; }
    }
    if (lane == 0) {                                                // the first entry behind j that is not a candidate: an implicit 0.0
        int z = j + 1;
        if (z % step == 0 && z / step < ncand) z = step == 1 ? ncand : z + 1;
        if (z < size && (0.0 > best || (0.0 == best && z < pos))) { best = 0.0; pos = z; }
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double ob = __shfl_xor(best, d);
        const int op = __shfl_xor(pos, d);
        if (ob > best || (ob == best && op < pos)) { best = ob; pos = op; }
    }
    if (lane == 0) {
        double cur = (j < size && j % step == 0 && j / step < ncand) ? v[j / step] : 0.0;
        int loc = j;
        if (best > cur) { cur = best; loc = pos; }
        delay[k] = loc;
        corr[k] = cur;
    }
}
