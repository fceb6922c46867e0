// RX front-end of the MFSK modes (ROBUST_0..2 = cfg 100..102) for gfx950.
//
// Reference span: the M == MOD_MFSK branch of cl_telecom_system::receive_byte,
// telecom_system.cc:1132-1192 — symbol_demod per active symbol (ofdm.cc:862-867), cl_mfsk::demod
// (mfsk.cc:288-390: noise variance from the carriers outside the tone band, tone energies with the hop
// undone, max-log LLR per Gray-mapped bit clamped to +-5), zeroing of the punctured tail of a short
// control frame — followed by the bit de-interleaver and the shortening re-pack
// (interleaver.cc:77-92, telecom_system.cc:1298-1308), which for these modes is one permutation.
//
// There is no channel estimate, so nothing couples the symbols of a frame: a workgroup takes a run of
// MF_SYMS symbols of one frame, one wavefront per symbol (FFT in registers + wave-private LDS, fft256.h), and
// scatters the 5 or 8 LLRs of each symbol straight to their decoder-input positions. HBM sees every
// sample once (16 B/sample, coalesced) and 1600 floats out per frame. FP64, no FMA contraction, sums in
// the reference's order: the LLRs are bit-identical to the CPU path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"
#include "fft256.h"

#define MF_THREADS 256
#define MF_WAVES (MF_THREADS / 64)
#define MF_SYMS 16

extern "C" int mgpu_mfsk_syms_per_block() { return MF_SYMS; }

namespace {

// In every MFSK mode the tone band is carriers [9, 41) of the 50 (cl_mfsk::init, mfsk.cc:69-76: nStreams * M == 32
// bins centred in Nc == 50); the host checks this before launching (api.hip).
constexpr int kBandStart = 9, kBandEnd = 41, kNoiseBins = 50 - (kBandEnd - kBandStart);

__device__ __forceinline__ double readlane_f64(double v, int src) {
    const unsigned lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const unsigned hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(int(hi), int(lo));
}

// max of two non-NaN doubles (the operands are energies or the -1e30 floor): one v_max_f64
__device__ __forceinline__ double shfl_xor_max(double v, int d) { return __builtin_fmax(__shfl_xor(v, d), v); }

// M tones per stream (32 or 16), NS streams (1 or 2): M * NS == 32 tone lanes per symbol
template <int M, int NS>
__device__ __forceinline__ void mfsk_frontend(const MgpuDev& T, const double* __restrict__ baseband, int F, int chunks,
                                              float* __restrict__ llr_out, float* __restrict__ variance_out,
                                              float* __restrict__ snr_variance_out, const MgpuTapsDev& taps) {
    constexpr int NB = M == 32 ? 5 : 4, BPS = NB * NS, HOP = M == 32 ? 13 : 7;   // mfsk.cc:56-66
    static_assert(M * NS == kBandEnd - kBandStart, "tone band");
    __shared__ __attribute__((aligned(16))) c2 tw[128];      // 16-byte aligned: fft256_twiddle reads it as ds_read_b128
    __shared__ __attribute__((aligned(16))) c2 fftb[MF_WAVES * FFT256_STRIDE];
    __shared__ double en[MF_WAVES][128];         // |carrier|^2 of the wave's current pair of symbols, carrier order, 64 per symbol

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.x / chunks, chunk = blockIdx.x - f * chunks;
    if (f >= F) return;
    const int Nc = 50, ns = T.active_nsymb;
    const int s0 = chunk * MF_SYMS, s1 = min(ns, s0 + MF_SYMS);
    const c2* bb = reinterpret_cast<const c2*>(baseband) + size_t(f) * T.frame_samples;
    float* out = llr_out + size_t(f) * T.N;

    for (int i = tid; i < 128; i += MF_THREADS) tw[fft256_tw_slot(i)] = {T.twiddle[2 * i], T.twiddle[2 * i + 1]};
    if (chunk == 0) {
        // short control frame: the LLRs of the bits that were not sent are 0 (telecom_system.cc:1183-1191)
        for (int i = T.active_nbits + tid; i < T.nBits; i += MF_THREADS) {
            out[T.llr_dst[i]] = 0.0f;
            if (taps.llr_demod) taps.llr_demod[size_t(f) * T.nBits + i] = 0.0f;
        }
        if (tid == 0) {
            // no variance on this path; receive_stats.SNR = 0.0 for a decoded MFSK frame (telecom_system.cc:1362-1367)
            variance_out[f] = 0.0f;
            if (snr_variance_out) snr_variance_out[f] = 1.0f;
            if (taps.variance) taps.variance[f] = 0.0;
            if (taps.agc_gain) taps.agc_gain[f] = 0.0;
        }
    }
    __syncthreads();

    // The demapper needs 32 lanes per symbol (M * NS tone lanes), so a wavefront transforms two symbols one after the other and
    // demaps them side by side: lanes 0-31 the first, lanes 32-63 the second. Lane (st, m) of a half = data tone m of stream st.
    const int half = lane >> 5, hl = lane & 31;
    const int st = hl >= M ? 1 : 0, m = hl & (M - 1);
    const int gray_m = m ^ (m >> 1);
    const int tone_off = kBandStart + st * M;
    const Fft256CarrierLane fcl = fft256_carrier_lane(lane);    // the one FFT bin this lane keeps (fft256.h)
    // (volatile: the 18 / 32 consecutive reads below stay ds_read_b64 - 2.2 LDS-pipeline cycles each; paired into ds_read2_b64 they take 8: tools/ubench/lds_mask.hip)
    const volatile double* E = en[wave] + half * 64;             // |carrier|^2 of this half's symbol, carrier order

    for (int sa = s0 + wave; sa < s1; sa += 2 * MF_WAVES) {
        const int sb = sa + MF_WAVES;                                // the pair's second symbol (may lie past the run's end)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = h ? sb : sa;
            double* Eo = en[wave] + h * 64;
            if (s >= s1) {                                           // no second symbol: its half demaps zeros and writes nothing
                Eo[lane] = 0.0;
                continue;
            }
            // no prefetch of the following symbol: its 16 registers per lane cost a workgroup per compute unit (94 vs 106: five instead of
            // four), and five workgroups hide the load latency better than the prefetch did (ROBUST_1: 0.73 -> 0.68 ms per 4096 frames)
            const c2* in = bb + size_t(s) * 272 + 16;                // gi_remover
            c2 r0 = in[lane], r1 = in[lane + 64], r2 = in[lane + 128], r3 = in[lane + 192];
            {                                                        // 1/Nfft scale + zero_depadder + energy: the lane's one kept bin (fft256.h)
                const c2 x = wave_fft256_carriers(r0, r1, r2, r3, fftb + wave * FFT256_STRIDE, tw, lane, fcl);
                if (fcl.col >= 0) {
                    const double re = x.re / 256.0, im = x.im / 256.0;
                    Eo[fcl.col] = re * re + im * im;
                    if (taps.grid) {
                        double* g = taps.grid + (size_t(f) * T.G + size_t(s) * Nc + fcl.col) * 2;
                        g[0] = re; g[1] = im;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int s = half ? sb : sa;

        // ---- noise variance: the out-of-band energies added in carrier order (mfsk.cc:303-318); every lane adds its half's 18
        // terms itself (the same address across a half: broadcast reads), so both halves have their sum without an exchange ----
        double noise_sum = 0.0;
        int noise_bins = 0;
        {
            const double ev = hl < kNoiseBins ? E[hl < kBandStart ? hl : hl + (kBandEnd - kBandStart)] : 0.0;
            if (__builtin_amdgcn_ballot_w64(!isfinite(ev)) == 0) {   // the usual case: every term counts
#pragma unroll
                for (int k = 0; k < kNoiseBins; ++k) noise_sum += E[k < kBandStart ? k : k + (kBandEnd - kBandStart)];
                noise_bins = kNoiseBins;
            } else {                                                 // NaN / Inf samples: skip those terms
                for (int k = 0; k < Nc; ++k) {
                    if (k >= kBandStart && k < kBandEnd) continue;
                    const double e = E[k];
                    if (isfinite(e)) { noise_sum += e; ++noise_bins; }
                }
            }
        }
        double noise_var = noise_bins > 0 ? noise_sum / noise_bins : 1e-30;
        if (noise_var < 1e-30) noise_var = 1e-30;
        const double llr_scale = 1.0 / (2.0 * noise_var);

        // ---- max-log LLRs (mfsk.cc:322-388). Tone energies with the hop undone sit one per lane; bit k of a
        // stream needs the maxima over the tones whose Gray label g = m ^ (m >> 1) has bit j = NB-1-k set / clear.
        // g_j = m_j ^ m_(j+1): the two sets are unions of aligned 2^j-blocks in the pattern 0 1 1 0 | 0 1 1 0 ...
        // Maxima are order-independent, so a butterfly gives the reference's sequential result exactly:
        // blk = max over the lane's aligned 2^j-block (shared tree), then xor 3*2^j and xor 4*2^j, 8*2^j, ... stay
        // inside the lane's own set, and one exchange at xor 2^j fetches the other set's maximum. All exchanges are
        // at distances below 32, i.e. inside a half.
        double e;
        {
            const int hop = (s * HOP) & (M - 1);
            e = E[tone_off + ((m + hop) & (M - 1))];
            if (!isfinite(e)) e = 0.0;
        }
        double blk = e;                                              // max over the aligned 2^j-block, j = 0 now
        double diff = 0.0;                                           // max_E0 - max_E1 of the bit this lane writes
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            double own = blk;
            if ((2 << j) < M) own = shfl_xor_max(own, 3 << j);
#pragma unroll
            for (int d = 4 << j; d < M; d <<= 1) own = shfl_xor_max(own, d);
            double other;
            if ((2 << j) <= M && (1 << j) < M) other = __shfl_xor(own, 1 << j);
            else other = -1e30;
            const bool set1 = (gray_m >> j) & 1;
            const double max_E1 = set1 ? own : other, max_E0 = set1 ? other : own;
            if (m == NB - 1 - j) diff = max_E0 - max_E1;             // lane m writes bit k = m of its stream
            if (j + 1 < NB) blk = shfl_xor_max(blk, 1 << j);         // next level of the shared tree
        }
        if (s < s1 && m < NB) {
            double llr = diff * llr_scale;
            if (!isfinite(llr)) llr = 0.0;
            else if (llr > 5.0) llr = 5.0;
            else if (llr < -5.0) llr = -5.0;
            const int idx = s * BPS + st * NB + m;
            if (idx >= T.puncture_from) llr = 0.0;                  // test_puncture_nBits (telecom_system.cc:1186-1192)
            out[T.llr_dst[idx]] = float(llr);
            if (taps.llr_demod) taps.llr_demod[size_t(f) * T.nBits + idx] = float(llr);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" __global__ __launch_bounds__(MF_THREADS) void mgpu_mfsk_frontend_kernel_m32(
    MgpuDev T, const double* __restrict__ baseband, int F, int chunks, float* __restrict__ llr_out,
    float* __restrict__ variance_out, float* __restrict__ snr_variance_out, MgpuTapsDev taps) {
    mfsk_frontend<32, 1>(T, baseband, F, chunks, llr_out, variance_out, snr_variance_out, taps);
}

extern "C" __global__ __launch_bounds__(MF_THREADS) void mgpu_mfsk_frontend_kernel_m16x2(
    MgpuDev T, const double* __restrict__ baseband, int F, int chunks, float* __restrict__ llr_out,
    float* __restrict__ variance_out, float* __restrict__ snr_variance_out, MgpuTapsDev taps) {
    mfsk_frontend<16, 2>(T, baseband, F, chunks, llr_out, variance_out, snr_variance_out, taps);
}

// ---- carrier energies of every symbol slot of a capture window: the FFT half of cl_ofdm::time_sync_mfsk
// (ofdm.cc:2011-2024) and cl_ofdm::detect_ack_pattern (ofdm.cc:2097-2105). Slot s of window w starts at
// s * Nofdm * interp + Ngi * interp; its 256 samples are taken `interp` apart (the reference's decimation), transformed
// with the 1/Nfft scale, and |X|^2 of the 50 carriers is written in carrier order. The sliding-window sums and
// the arg-max over slots are a few hundred scalar operations per window and stay on the host (api.hip).
extern "C" __global__ __launch_bounds__(MF_THREADS) void mgpu_slot_energy_kernel(
    const double* __restrict__ baseband_interp, int size, int nslots, int interp, const double* __restrict__ twiddle,
    double* __restrict__ energy /*[W][nslots][50]*/) {
    __shared__ __attribute__((aligned(16))) c2 tw[128];      // 16-byte aligned: fft256_twiddle reads it as ds_read_b128
    __shared__ __attribute__((aligned(16))) c2 fftb[MF_WAVES * FFT256_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.y, s = blockIdx.x * MF_WAVES + wave;
    for (int i = tid; i < 128; i += MF_THREADS) tw[fft256_tw_slot(i)] = {twiddle[2 * i], twiddle[2 * i + 1]};
    __syncthreads();
    if (s >= nslots) return;
    const int offset = s * 272 * interp + 16 * interp;
    if (offset + 256 * interp > size) return;
    const c2* in = reinterpret_cast<const c2*>(baseband_interp) + size_t(w) * size + offset;
    c2 r0 = in[size_t(lane) * interp], r1 = in[size_t(lane + 64) * interp];
    c2 r2 = in[size_t(lane + 128) * interp], r3 = in[size_t(lane + 192) * interp];
    const Fft256CarrierLane fcl = fft256_carrier_lane(lane);
    const c2 x = wave_fft256_carriers(r0, r1, r2, r3, fftb + wave * FFT256_STRIDE, tw, lane, fcl);
    double* E = energy + (size_t(w) * nslots + s) * 50;
    if (fcl.col >= 0) {
        const double re = x.re / 256.0, im = x.im / 256.0;
        E[fcl.col] = re * re + im * im;
    }
}

// The other half of cl_ofdm::time_sync_mfsk (ofdm.cc:2026-2061) where the energies lie: for every start slot s the metric
// sum_p e_target(s + p) / e_total(s + p) over the np preamble symbols (e_total: the 50 carriers added in carrier order; e_target: the
// streams' tones of preamble symbol p added in stream order; a slot with e_total <= 0 adds nothing; the sum stops at the first symbol
// that does not fit the buffer), and the first slot with the largest metric from search_start on (the reference's strict ">"; nothing
// beats the initial -1 -> slot 0). One workgroup per window; same operations in the same order as mfsk_sync_from_energies (api.hip),
// which stays as the few-window path and the host-side statement the CPU tests pin. 66 MB of energies per 256 ROBUST_0 windows no
// longer cross PCIe for a few hundred scalar operations per window.
extern "C" __global__ __launch_bounds__(256) void mgpu_mfsk_sync_kernel(
    const double* __restrict__ energy /*[W][nslots][Nc]*/, int nslots, int size, MgpuMfskSync P, const int* __restrict__ search_start /*[W] or null*/,
    int* __restrict__ delay_out) {
    extern __shared__ double ratio[];                                // [np][nslots]
    __shared__ double best_v[256];
    __shared__ int best_s[256];
    const int w = blockIdx.x, tid = threadIdx.x;
    const double* E = energy + size_t(w) * nslots * P.Nc;
    for (int sl = tid; sl < nslots; sl += 256) {
        const double* e = E + size_t(sl) * P.Nc;
        double e_total = 0;
        for (int k = 0; k < P.Nc; ++k) e_total += e[k];
        for (int p = 0; p < P.np; ++p) {
            double e_target = 0;
            for (int st = 0; st < P.nstreams; ++st) e_target += e[P.off[st] + P.tones[p]];
            ratio[p * nslots + sl] = e_total > 0 ? e_target / e_total : 0.0;       // x + (+0.0) == x: the same as adding nothing
        }
    }
    __syncthreads();
    const int s0 = search_start ? max(search_start[w], 0) : 0;
    double bv = -1.0;
    int bs = 0x7fffffff;
    for (int s = s0 + tid; s <= nslots - P.np; s += 256) {
        double metric = 0;
        for (int p = 0; p < P.np; ++p) {
            if ((s + p) * P.sym_period + P.tail > size) break;
            metric += ratio[p * nslots + s + p];
        }
        if (metric > bv) { bv = metric; bs = s; }                   // a lane's slots ascend: strict ">" keeps its first maximum
    }
    best_v[tid] = bv; best_s[tid] = bs;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (tid < d) {
            const double ov = best_v[tid + d];
            const int os = best_s[tid + d];
            if (os != 0x7fffffff && (best_s[tid] == 0x7fffffff || ov > best_v[tid] || (ov == best_v[tid] && os < best_s[tid]))) { best_v[tid] = ov; best_s[tid] = os; }
        }
        __syncthreads();
    }
    if (tid == 0) delay_out[w] = (best_s[0] == 0x7fffffff ? 0 : best_s[0]) * P.sym_period;
}

