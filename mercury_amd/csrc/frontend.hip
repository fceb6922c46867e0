// RX front-end for gfx950: one workgroup per OFDM frame, everything between the baseband
// samples and the LDPC input stays in LDS; HBM sees the samples once (16 B/sample, coalesced)
// and 1600 floats out.
//
// Stages and the reference code they reproduce (same operations, same order, FP64, no FMA
// contraction — this TU is built with -ffp-contract=off):
//   symbol_demod ............. ofdm.cc:862-867 (gi_remover :423-429, radix-2 DIT fft :310-340 with
//                              1/Nfft scaling :431-444, zero_depadder :401-411)
//   automatic_gain_control ... ofdm.cc:1467-1498
//   LS_/ZF_channel_estimator . ofdm.cc:1315-1451 / :1266-1313 ; matrix_multiplication misc.cc:73-91
//   interpolate_linear_col ... interpolator.cc:163-254
//   restore_channel_amplitude  ofdm.cc:1453-1466 ; get_angle/set_complex misc.cc:34-71
//   channel_equalizer ........ ofdm.cc:1637-1647 (complex divide = libgcc __divdc3, Smith)
//   measure_variance ......... ofdm.cc:1500-1521 (sequential sum, narrowed to float by the callers)
//   deframer + deinterleaver . ofdm.cc:837-852, interleaver.cc:94-109   (one gather table)
//   cl_psk::demod ............ psk.cc:278-326 (double distances narrowed to float, float LLR math)
//   deinterleaver + re-pack .. interleaver.cc:77-92, telecom_system.cc:1300-1308 (one gather table)
//
// Sequential reductions (AGC mean, variance) are kept sequential on purpose: the terms are
// produced in parallel, one lane adds them in the reference's order, so the float `variance`
// that scales every LLR is bit-identical to the CPU path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"
#include "fft256.h"
#include "fe_math.h"

namespace {

// restore_channel_amplitude for one cell (ofdm.cc:1453-1466, set_complex misc.cc:65-71), with the reference platform's atan and
// sincos restated (glibc_trig.h): the unit phasor is bit-identical to the CPU's. Kept out of line so that the constants of the
// three routines do not stay live (and spill) across the rest of the front-end kernel.
__device__ __attribute__((noinline)) c2 unit_phasor(c2 h) {
    const double th = get_angle(h);
    double s, c;
    gl_sincos(th, &s, &c);
    return {1 * c, 1 * s};
}

}  // namespace

// device probe of glibc_trig.h for tests: atan, sin and cos of n arguments
extern "C" __global__ void mgpu_glibc_trig_probe_kernel(const double* __restrict__ in, double* __restrict__ out_atan, double* __restrict__ out_sin,
                                                        double* __restrict__ out_cos, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_atan[i] = gl_atan(in[i]);
    double s, c;
    gl_sincos(in[i], &s, &c);
    out_sin[i] = s;
    out_cos[i] = c;
}

// Sum n doubles from LDS in index order with one lane: the additions are a dependent chain by
// construction (the reference's `+=` loop), so the only thing to hide is the LDS latency: fetch eight
// ahead while the previous eight are being added.
__device__ __forceinline__ double serial_sum(const double* red, int n) {
    double acc = 0;
    int p = 0;
    for (; p + 8 <= n; p += 8) {
        const double a0 = red[p], a1 = red[p + 1], a2 = red[p + 2], a3 = red[p + 3];
        const double a4 = red[p + 4], a5 = red[p + 5], a6 = red[p + 6], a7 = red[p + 7];
        acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
    }
    for (; p < n; ++p) acc += red[p];
    return acc;
}

// Workgroup size: 512 threads (three workgroups per compute unit in the QPSK and QAM modes, two in the long BPSK frames); a
// 1024-thread variant exists for configurations whose LDS carve lets only one workgroup onto a compute unit.

// LDS carve (bytes): grid 16G (the carrier grid, equalised in place at the end) | work = Hp 16 nPilots (channel estimate at the pilots,
//                    pilot order) + rsz (reduction terms / signed pilots / demapper LLRs), first the FFT work areas | aux = tw 2048
//                    during the FFTs, then type G + scal 64 (the cell types are first needed by the estimator)
// The channel estimate of a DATA cell is never stored: interpolation, amplitude restoration and the equaliser are applied to a cell
// in one go (the interpolated value is a function of two pilot estimates). That keeps the frame at one grid + one pilot-sized
// array: 40 KB in mode 8, 67 KB in the BPSK modes (92 KB with a second full grid: one workgroup per compute unit).
// The FFT runs on as many wavefronts as work areas fit next to the grid without costing a workgroup per compute unit.
struct FeCarve { int fft_waves; size_t rsz, work, total; };
__host__ __device__ inline FeCarve fe_carve(int G, int nPilots, int nBits, int FE_WAVES) {
    FeCarve c;
    c.rsz = size_t(16) * (nPilots + 8) > size_t(4) * nBits ? size_t(16) * (nPilots + 8) : size_t(4) * nBits;   // + 8: zero pad behind the signed pilots (LS row reads)
    c.rsz = (c.rsz + 15) & ~size_t(15);
    const size_t per_wave = size_t(FFT256_STRIDE) * 16, lds_cu = size_t(160) * 1024;
    const size_t aux0 = size_t((G + 15) & ~15) + 64, aux = aux0 > 2048 ? aux0 : 2048;
    const size_t fixed = size_t(16) * G + aux, need = size_t(16) * nPilots + c.rsz;
    const size_t minimal = fixed + (need > 4 * per_wave ? need : 4 * per_wave);
    // LDS is handed out in blocks of 1280 bytes (measured: three workgroups of 53,504 bytes run side by side on a compute unit, three of
    // 54,016 do not — the runtime's occupancy calculator says 3 for both), so a workgroup's share is a whole number of blocks
    const size_t block = 1280;
    // ... less 256 bytes: round 4 measured three workgroups of 53,504 bytes resident and three of 53,760 (42 blocks exactly) not. The same
    // budget (bytes + 256, rounded up to blocks) counts the workgroups AND sizes the FFT work areas.
    const size_t overhead = 256;
    size_t wgs = lds_cu / ((minimal + overhead + block - 1) / block * block);  // workgroups per compute unit by LDS with the fewest FFT work areas
    const size_t cap = size_t(24 / FE_WAVES) > 0 ? size_t(24 / FE_WAVES) : 1;    // 80 registers per lane: 24 wavefronts per compute unit
    if (wgs > cap) wgs = cap;
    if (wgs < 1) wgs = 1;
    const size_t share = lds_cu / wgs / block * block;
    size_t w = share > fixed + overhead ? (share - overhead - fixed) / per_wave : 0;
    c.fft_waves = int(w < 4 ? 4 : (w > size_t(FE_WAVES) ? size_t(FE_WAVES) : w));
    c.work = need > c.fft_waves * per_wave ? need : c.fft_waves * per_wave;
    c.total = fixed + c.work;
    return c;
}
extern "C" size_t mgpu_frontend_lds_bytes(int G, int nPilots, int nBits, int threads) { return fe_carve(G, nPilots, nBits, threads / 64).total; }

template <int FE_THREADS>
__device__ __forceinline__ void fe_frame(const MgpuDev& T, const double* __restrict__ baseband, int F, float* __restrict__ llr_out,
                                         float* __restrict__ variance_out, float* __restrict__ snr_variance_out, double* __restrict__ eqdata_out,
                                         const MgpuTapsDev& taps) {
    constexpr int FE_WAVES = FE_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int G = T.G, Nc = 50, Ns = T.Nsymb;
    const FeCarve carve = fe_carve(G, T.nPilots, T.nBits, FE_WAVES);
    c2* grid = reinterpret_cast<c2*>(smem);
    c2* Hp = grid + G;                                              // channel estimate at the pilots, pilot order; Hp and red are the FFT work area first
    c2* fftb = Hp;
    double* red = reinterpret_cast<double*>(__builtin_assume_aligned(Hp + T.nPilots, 16));        // nPilots doubles
    float* llr = reinterpret_cast<float*>(red);                     // demapper output reuses the reduction area
    c2* yp = reinterpret_cast<c2*>(red);                            // pilots in row-major pilot order, multiplied by their sign
    // (every piece of the carve is a multiple of 16 bytes; said so, a c2 is one ds_read_b128 - 4 LDS-pipeline cycles - instead of a ds_read2_b64 - 8: tools/ubench/lds_mask.hip)
    c2* tw = reinterpret_cast<c2*>(__builtin_assume_aligned(reinterpret_cast<unsigned char*>(Hp) + carve.work, 16));
    int8_t* type = reinterpret_cast<int8_t*>(tw);                   // 0 data, +1 / -1 pilot with that sign; takes the twiddles' place after the FFTs
    double* scal = reinterpret_cast<double*>(type + ((G + 15) & ~15));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.x;
    if (f >= F) return;
    const c2* bb = reinterpret_cast<const c2*>(baseband) + size_t(f) * T.frame_samples;
    const double boost = T.pilot_boost;
    int stamp_i = 0;
#define FE_STAMP() do { if (taps.cycles && f == (F >> 1) && tid == 0) taps.cycles[stamp_i] = __builtin_readcyclecounter(); ++stamp_i; } while (0)    /* a frame from the middle of the launch: the compute unit is in its steady mix of phases */
    FE_STAMP();

    for (int i = tid; i < 128; i += FE_THREADS) tw[fft256_tw_slot(i)] = {T.twiddle[2 * i], T.twiddle[2 * i + 1]};
    __syncthreads();

    // ---- symbol_demod: one wave per symbol (fft256.h), next symbol's samples requested before the butterflies ----
    c2 n0 = {0, 0}, n1 = {0, 0}, n2 = {0, 0}, n3 = {0, 0};
    const Fft256CarrierLane fcl = fft256_carrier_lane(lane);
    const int nfw = carve.fft_waves;                                // waves that own an FFT work area
    if (wave < nfw && wave < Ns) {
        const c2* in = bb + size_t(wave) * 272 + 16;                // gi_remover
        n0 = in[lane]; n1 = in[lane + 64]; n2 = in[lane + 128]; n3 = in[lane + 192];
    }
    for (int s = wave; wave < nfw && s < Ns; s += nfw) {
        c2 r0 = n0, r1 = n1, r2 = n2, r3 = n3;
        if (s + nfw < Ns) {
            const c2* in = bb + size_t(s + nfw) * 272 + 16;
            n0 = in[lane]; n1 = in[lane + 64]; n2 = in[lane + 128]; n3 = in[lane + 192];
        }
        // 1/Nfft scale + zero_depadder: a lane ends with the one bin it keeps, if any (fft256.h: 50 of the 256 bins are carriers)
        const c2 x = wave_fft256_carriers(r0, r1, r2, r3, fftb + wave * FFT256_STRIDE, tw, lane, fcl);
        if (fcl.col >= 0) grid[s * Nc + fcl.col] = {x.re / 256.0, x.im / 256.0};
    }
    __syncthreads();
    for (int i = tid; i < G; i += FE_THREADS) type[i] = T.cell_type[i] ? (T.pilot_val[i] < 0 ? int8_t(-1) : int8_t(1)) : int8_t(0);
    __syncthreads();

    FE_STAMP();   // 1: FFT done
    // ---- automatic_gain_control ---------------------------------------------------------------
    if (T.agc) {
        for (int p = tid; p < T.nPilots; p += FE_THREADS) {
            const c2 y = grid[T.pilot_cell[p]];
            red[p] = sqrt(y.re * y.re + y.im * y.im);
        }
        __syncthreads();
        if (tid == 0) {
            double amp = serial_sum(red, T.nPilots);
            amp /= T.nPilots;
            scal[0] = boost / amp;
        }
        __syncthreads();
        const double agc = scal[0];
        for (int c = tid; c < G; c += FE_THREADS) grid[c] = {grid[c].re * agc, grid[c].im * agc};
        __syncthreads();
        if (taps.agc_gain && tid == 0) taps.agc_gain[f] = agc;
    }
    if (taps.grid) for (int c = tid; c < G; c += FE_THREADS) { taps.grid[(size_t(f) * G + c) * 2] = grid[c].re; taps.grid[(size_t(f) * G + c) * 2 + 1] = grid[c].im; }

    FE_STAMP();   // 2: AGC done
    // ---- channel estimate at the pilots -------------------------------------------------------
    const int hw = T.lsw / 2;
    const bool ls_fast = T.estimator != 0 && T.regular_lattice;
    int ls_rows = 0;     // 1: every window row holds >= 3 pilots and every pilot of the frame is finite -> the branch-free row loop below
    if (ls_fast) {       // x*y for the LS sums: the pilot's sign applied once ((-w)*y == w*(-y) exactly), in pilot order
        int finite = 1;
        for (int p = tid; p < T.nPilots + 8; p += FE_THREADS) {
            if (p >= T.nPilots) { yp[p] = {0.0, 0.0}; continue; }
            const int q = T.pilot_cell[p];
            const c2 y = grid[q];
            yp[p] = type[q] < 0 ? c2{-y.re, -y.im} : y;
            finite &= (fabs(y.re) < __builtin_inf()) & (fabs(y.im) < __builtin_inf());
        }
        ls_rows = __syncthreads_and(finite) && T.regular_lattice == 2;
    }
    for (int p = tid; p < T.nPilots; p += FE_THREADS) {
        const int c = T.pilot_cell[p], i = c / Nc, j = c - i * Nc;
        if (T.estimator == 0) {            // ZF: Y / (x + 0i) reduces to two real divisions in __divdc3
            const double x = type[c] < 0 ? -boost : boost;
            Hp[p] = {grid[c].re / x, grid[c].im / x};
        } else {                           // LS over the (clipped) 21x21 window, row-major order
            const int k0 = max(i - hw, 0), k1 = min(i + hw, Ns - 1), l0 = max(j - hw, 0), l1 = min(j + hw, Nc - 1);
            double hr = 0, hi = 0;
            if (ls_fast) {
                // pilots of row k sit at columns == k (mod 3); rows hold 17,17,16 pilots cyclically, so the pilot
                // index of (k, l) is 50*(k/3) + {0,17,34}[k%3] + (l - k%3)/3 and a row's window pilots are contiguous
                // per column residue r = k % 3: first window column == r (mod 3), how many pilots, and the offset of
                // that first pilot inside its row; computed once per pilot, then rows just cycle through r
                int cntr[3], offr[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int first = l0 + ((r - l0) % 3 + 3) % 3;
                    cntr[r] = first <= l1 ? (l1 - first) / 3 + 1 : 0;
                    offr[r] = (r == 0 ? 0 : r == 1 ? 17 : 34) + (first - r) / 3;
                }
                int n = 0;
                {
                    int r = k0 % 3;
                    for (int k = k0; k <= k1; ++k) { n += r == 0 ? cntr[0] : r == 1 ? cntr[1] : cntr[2]; r = r == 2 ? 0 : r + 1; }
                }
                const double w = T.ls_weight[n];
                if (ls_rows) {
                    // Rows k0, k0+1, k0+2, k0+3, ... cycle through the three column residues, so a lane's (pilot count, first pilot) pair of a
                    // row depends only on the row's place in that cycle. A row's first three pilots are always inside the window; pilots
                    // four to seven are added with the weight w or +0.0: x + (+-0 * y) == x exactly for finite y, and the sums start at +0.0
                    // and can therefore never be -0.0. Same terms in the same order as the loop below, without its per-term branches.
                    const int r0 = k0 % 3;
                    int ptr[3];
                    double wq[3][4];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int r = r0 + q >= 3 ? r0 + q - 3 : r0 + q;
                        const int cn = r == 0 ? cntr[0] : r == 1 ? cntr[1] : cntr[2];
                        ptr[q] = 50 * ((k0 + q) / 3) + (r == 0 ? offr[0] : r == 1 ? offr[1] : offr[2]);
#pragma unroll
                        for (int m = 0; m < 4; ++m) wq[q][m] = m + 3 < cn ? w : 0.0;
                    }
                    auto add_row = [&](int q) {
                        const c2* row = yp + ptr[q];
                        const c2 v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3], v4 = row[4], v5 = row[5], v6 = row[6];
                        hr += w * v0.re; hi += w * v0.im;
                        hr += w * v1.re; hi += w * v1.im;
                        hr += w * v2.re; hi += w * v2.im;
                        hr += wq[q][0] * v3.re; hi += wq[q][0] * v3.im;
                        hr += wq[q][1] * v4.re; hi += wq[q][1] * v4.im;
                        hr += wq[q][2] * v5.re; hi += wq[q][2] * v5.im;
                        hr += wq[q][3] * v6.re; hi += wq[q][3] * v6.im;
                        ptr[q] += 50;
                    };
                    for (int k = k0; k <= k1; k += 3) {
                        add_row(0);
                        if (k + 1 <= k1) add_row(1);
                        if (k + 2 <= k1) add_row(2);
                    }
                    Hp[p] = {hr, hi};
                    continue;
                }
                int km = k0 % 3, rowbase = 50 * (k0 / 3);
                for (int k = k0; k <= k1; ++k) {
                    const int cnt = km == 0 ? cntr[0] : km == 1 ? cntr[1] : cntr[2];
                    const c2* row = yp + rowbase + (km == 0 ? offr[0] : km == 1 ? offr[1] : offr[2]);
                    if (km == 2) { km = 0; rowbase += 50; } else ++km;
                    if (cnt == 0) continue;
                    // a window row holds at most 7 pilots: fetch all seven at once (one LDS latency per row; reading past
                    // the row's end stays inside the LDS carve), add the first cnt in order
                    const c2 v0 = row[0], v1 = row[1], v2 = row[2], v3 = row[3], v4 = row[4], v5 = row[5], v6 = row[6];
                    hr += w * v0.re; hi += w * v0.im;
                    if (cnt > 1) { hr += w * v1.re; hi += w * v1.im; }
                    if (cnt > 2) { hr += w * v2.re; hi += w * v2.im; }
                    if (cnt > 3) { hr += w * v3.re; hi += w * v3.im; }
                    if (cnt > 4) { hr += w * v4.re; hi += w * v4.im; }
                    if (cnt > 5) { hr += w * v5.re; hi += w * v5.im; }
                    if (cnt > 6) { hr += w * v6.re; hi += w * v6.im; }
                }
            } else {
                int n = 0;
                for (int k = k0; k <= k1; ++k)
                    for (int l = l0; l <= l1; ++l) n += type[k * Nc + l] != 0;
                const double w = T.ls_weight[n];
                for (int k = k0; k <= k1; ++k)
                    for (int l = l0; l <= l1; ++l) {
                        const int q = k * Nc + l;
                        if (!type[q]) continue;
                        const double xw = type[q] < 0 ? -w : w;
                        hr += xw * grid[q].re;
                        hi += xw * grid[q].im;
                    }
            }
            Hp[p] = {hr, hi};
        }
    }
    __syncthreads();
    FE_STAMP();   // 3: estimate done
    // ---- interpolate_linear_col + restore_channel_amplitude + channel_equalizer, one cell at a time ----------------------
    if (taps.mean_H) {      // mean |H| over the MEASURED (pilot) cells in cell order, telecom_system.cc:1224-1243
        for (int p = tid; p < T.nPilots; p += FE_THREADS) { const c2 h = Hp[p]; red[p] = hypot(h.re, h.im); }
        __syncthreads();
        if (tid == 0) taps.mean_H[f] = T.nPilots > 0 ? serial_sum(red, T.nPilots) / T.nPilots : -1.0;
        __syncthreads();
    }
    FE_STAMP();   // 4: (interpolation is part of the equaliser pass below)
    // A data cell (i, j): column j has its pilots in the rows == j (mod 3) (checked on the host), so the two pilot rows it interpolates
    // between follow from (i - j) mod 3 (above the first / below the last pilot: extrapolation from the nearest two), and a pilot of
    // row r, column j has index first_pilot(r) + j / 3 in pilot order (rows hold 17, 17, 16 pilots cyclically). The cell's estimate
    // = lerp of the two pilot estimates (interpolator.cc:163-254), made a unit phasor in the PSK modes (ofdm.cc:1453-1466), divides the
    // received cell (ofdm.cc:1637-1647); the equalised value replaces the received one in the grid. Same operations per cell as the
    // reference's three passes over a full channel grid.
    const uint2* __restrict__ cell_lerp = reinterpret_cast<const uint2*>(T.cell_lerp);
    auto equalise_data = [&](int idx) {
        // the cell, the pilot rows a / b it interpolates between and their pilots' indices in its column: tabulated on the host (api.hip)
        const uint2 q = cell_lerp[idx];
        const int c = int(q.x & 0xfffu), pa = int((q.x >> 12) & 0x3ffu), pb = int(q.x >> 22);
        const int a = int(q.y & 0xffu), b = int((q.y >> 8) & 0xffu), i = int(q.y >> 16);
        c2 h = lerp(Hp[pa], double(a), Hp[pb], double(b), double(i));
        if (T.amp_restore) h = unit_phasor(h);
        if (taps.H) { taps.H[(size_t(f) * G + c) * 2] = h.re; taps.H[(size_t(f) * G + c) * 2 + 1] = h.im; }
        grid[c] = cdiv(grid[c], h);
    };
    // The pilot cells first, all of them in one pass: the term of the variance before amplitude restoration (PSK modes; SNR report),
    // the cell's own equalisation, and the term of the variance that scales the LLRs (from the un-equalised cell in the
    // baseband_test_EsN0 variant). Both variances are sums of nPilots terms in pilot order — dependent chains for one lane — so
    // wavefront 0 adds them while the other wavefronts equalise the data cells, handed out in runs of 64 from a counter in LDS;
    // wavefront 0 joins when its sums are done.
    double* red2 = reinterpret_cast<double*>(__builtin_assume_aligned(red + ((T.nPilots + 1) & ~1), 16));   // second term array (the signed pilots are no longer needed), on a 16-byte boundary
    int* queue = reinterpret_cast<int*>(scal + 3);
    if (tid == 0) *queue = 0;
    for (int p = tid; p < T.nPilots; p += FE_THREADS) {
        const int c = T.pilot_cell[p];
        const c2 g = grid[c];
        c2 h = Hp[p];
        const double x = type[c] < 0 ? -boost : boost;
        if (T.amp_restore) {                                         // measure_variance(equalized_data_without_amplitude_restoration)
            const c2 e0 = cdiv(g, h);
            const double dr = e0.re - x, di = e0.im - 0.0;
            red[p] = dr * dr + di * di;
            h = unit_phasor(h);
        }
        if (taps.H) { taps.H[(size_t(f) * G + c) * 2] = h.re; taps.H[(size_t(f) * G + c) * 2 + 1] = h.im; }
        const c2 e = cdiv(g, h);
        grid[c] = e;
        const c2 v = T.var_eq ? e : g;
        const double dr = v.re - x, di = v.im - 0.0;
        red2[p] = dr * dr + di * di;
    }
    __syncthreads();
    FE_STAMP();   // 5: pilot cells done
    if (tid == 0) {
        if (T.amp_restore) {
            double var = serial_sum(red, T.nPilots);
            var /= double(T.nPilots);
            scal[2] = var;
        }
        double var = serial_sum(red2, T.nPilots);
        var /= double(T.nPilots);
        scal[1] = var;
    }
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(queue, 64);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= T.nData) break;
        if (base + lane < T.nData) equalise_data(base + lane);
    }
    __syncthreads();
    c2* eq = grid;
    if (taps.eq) for (int c = tid; c < G; c += FE_THREADS) { taps.eq[(size_t(f) * G + c) * 2] = eq[c].re; taps.eq[(size_t(f) * G + c) * 2 + 1] = eq[c].im; }
    if (eqdata_out)   // de-framed equalised symbols, kept for the zero-forcing modes' post-decode SNR (ofdm_deframed_data)
        for (int i = tid; i < T.nData; i += FE_THREADS) { const c2 e = eq[T.data_cell[i]]; eqdata_out[(size_t(f) * T.nData + i) * 2] = e.re; eqdata_out[(size_t(f) * T.nData + i) * 2 + 1] = e.im; }
    const float variance = float(scal[1]);
    if (tid == 0) {
        variance_out[f] = variance;
        if (snr_variance_out) snr_variance_out[f] = T.amp_restore ? float(scal[2]) : variance;
        if (taps.variance) taps.variance[f] = scal[1];
    }

    FE_STAMP();   // 6: equalise + variance done
    // ---- deframe + time/freq de-interleave + max-log demap -------------------------------------
    const float inv_var = 1 / variance;
    const int M = T.M, bps = T.bps;
    // cl_psk::demod (psk.cc:278-326): squared distance to every constellation point in double, narrowed to float; per bit the smallest
    // distance among the points with that bit set / clear; LLR = (d1 - d0) / variance in float. Specialised per constellation size so that
    // "which of the two minima does point j feed" is a compile-time fact (one v_min_f32 per point and bit) — with M at run time it is a
    // bit test, two compares and two selects per point and bit, and the 32 points of mode 16 were half of that mode's front-end time.
    // fminf keeps the running minimum when the new distance is NaN, like the reference's `if (D < d)`.
    auto demap_all = [&](auto m_tag) {
        constexpr int MM = decltype(m_tag)::value;
        constexpr int BPS = MM == 2 ? 1 : MM == 4 ? 2 : MM == 8 ? 3 : MM == 16 ? 4 : 5;
        for (int k = tid; k < T.nData; k += FE_THREADS) {
            const c2 s = eq[T.sym_src[k]];
            if (taps.syms) { taps.syms[(size_t(f) * T.nData + k) * 2] = s.re; taps.syms[(size_t(f) * T.nData + k) * 2 + 1] = s.im; }
            float d0[BPS], d1[BPS];
#pragma unroll
            for (int b = 0; b < BPS; ++b) { d0[b] = __builtin_inff(); d1[b] = __builtin_inff(); }
#pragma unroll
            for (int j = 0; j < MM; ++j) {
                const double dr = s.re - T.constellation[2 * j], di = s.im - T.constellation[2 * j + 1];
                const float D = float(dr * dr + di * di);
#pragma unroll
                for (int b = 0; b < BPS; ++b) {
                    if ((j >> b) & 1) d1[b] = __builtin_fminf(d1[b], D);
                    else d0[b] = __builtin_fminf(d0[b], D);
                }
            }
#pragma unroll
            for (int b = 0; b < BPS; ++b) llr[k * BPS + (BPS - 1 - b)] = inv_var * (d1[b] - d0[b]);
        }
    };
    if (M == 2 && bps == 1) demap_all(std::integral_constant<int, 2>());
    else if (M == 4 && bps == 2) demap_all(std::integral_constant<int, 4>());
    else if (M == 8 && bps == 3) demap_all(std::integral_constant<int, 8>());
    else if (M == 16 && bps == 4) demap_all(std::integral_constant<int, 16>());
    else if (M == 32 && bps == 5) demap_all(std::integral_constant<int, 32>());
    else
    for (int k = tid; k < T.nData; k += FE_THREADS) {
        const c2 s = eq[T.sym_src[k]];
        if (taps.syms) { taps.syms[(size_t(f) * T.nData + k) * 2] = s.re; taps.syms[(size_t(f) * T.nData + k) * 2 + 1] = s.im; }
        float d0[5], d1[5];
#pragma unroll
        for (int b = 0; b < 5; ++b) { d0[b] = __builtin_inff(); d1[b] = __builtin_inff(); }
        for (int j = 0; j < M; ++j) {
            const double dr = s.re - T.constellation[2 * j], di = s.im - T.constellation[2 * j + 1];
            const float D = float(dr * dr + di * di);
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                if (b < bps) {
                    if ((j >> b) & 1) { if (D < d1[b]) d1[b] = D; }
                    else { if (D < d0[b]) d0[b] = D; }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 5; ++b)
            if (b < bps) llr[k * bps + (bps - 1 - b)] = inv_var * (d1[b] - d0[b]);
    }
    __syncthreads();
    if (taps.llr_demod) for (int i = tid; i < T.nBits; i += FE_THREADS) taps.llr_demod[size_t(f) * T.nBits + i] = llr[i];
    FE_STAMP();   // 7: demap done
    // ---- bit de-interleave + shortening re-pack -------------------------------------------------
    for (int p = tid; p < T.N; p += FE_THREADS) llr_out[size_t(f) * T.N + p] = llr[T.llr_src[p]];
    FE_STAMP();   // 8: end
#undef FE_STAMP
}

extern "C" __global__ __launch_bounds__(512, 6) void mgpu_frontend_kernel(
    MgpuDev T, const double* __restrict__ baseband, int F, float* __restrict__ llr_out,
    float* __restrict__ variance_out, float* __restrict__ snr_variance_out, double* __restrict__ eqdata_out, MgpuTapsDev taps) {
    fe_frame<512>(T, baseband, F, llr_out, variance_out, snr_variance_out, eqdata_out, taps);
}

extern "C" __global__ __launch_bounds__(1024, 4) void mgpu_frontend_kernel_t1024(
    MgpuDev T, const double* __restrict__ baseband, int F, float* __restrict__ llr_out,
    float* __restrict__ variance_out, float* __restrict__ snr_variance_out, double* __restrict__ eqdata_out, MgpuTapsDev taps) {
    fe_frame<1024>(T, baseband, F, llr_out, variance_out, snr_variance_out, eqdata_out, taps);
}
