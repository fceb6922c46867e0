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

extern "C" __global__ __launch_bounds__(MF_THREADS) void mgpu_mfsk_frontend_kernel(
    MgpuDev T, const double* __restrict__ baseband, int F, int chunks, float* __restrict__ llr_out,
    float* __restrict__ variance_out, float* __restrict__ snr_variance_out, MgpuTapsDev taps) {
    __shared__ c2 tw[128];
    __shared__ c2 fftb[MF_WAVES * FFT256_STRIDE];
    __shared__ double en[MF_WAVES][64];          // |carrier|^2 of the wave's current symbol, carrier order

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.x / chunks, chunk = blockIdx.x - f * chunks;
    if (f >= F) return;
    const int Nc = 50, ns = T.active_nsymb, bps = T.bps, M = T.mfsk_M, nb = T.mfsk_nbits;
    const int s0 = chunk * MF_SYMS, s1 = min(ns, s0 + MF_SYMS);
    const c2* bb = reinterpret_cast<const c2*>(baseband) + size_t(f) * T.frame_samples;
    float* out = llr_out + size_t(f) * T.N;

    for (int i = tid; i < 128; i += MF_THREADS) tw[i] = {T.twiddle[2 * i], T.twiddle[2 * i + 1]};
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

    const int band_start = T.mfsk_off0;
    const int band_end = (T.mfsk_nstreams > 1 ? T.mfsk_off1 : T.mfsk_off0) + M;
    c2 n0 = {0, 0}, n1 = {0, 0}, n2 = {0, 0}, n3 = {0, 0};
    if (s0 + wave < s1) {
        const c2* in = bb + size_t(s0 + wave) * 272 + 16;            // gi_remover
        n0 = in[lane]; n1 = in[lane + 64]; n2 = in[lane + 128]; n3 = in[lane + 192];
    }
    for (int s = s0 + wave; s < s1; s += MF_WAVES) {
        c2 r0 = n0, r1 = n1, r2 = n2, r3 = n3;
        if (s + MF_WAVES < s1) {
            const c2* in = bb + size_t(s + MF_WAVES) * 272 + 16;
            n0 = in[lane]; n1 = in[lane + 64]; n2 = in[lane + 128]; n3 = in[lane + 192];
        }
        wave_fft256(r0, r1, r2, r3, fftb + wave * FFT256_STRIDE, tw, lane);
        double* E = en[wave];
        auto emit = [&](const c2& x, int p) {                        // 1/Nfft scale + zero_depadder + energy
            const int col = carrier_of_bin(brev8(p));
            if (col < 0) return;
            const double re = x.re / 256.0, im = x.im / 256.0;
            E[col] = re * re + im * im;
            if (taps.grid) {
                double* g = taps.grid + (size_t(f) * T.G + size_t(s) * Nc + col) * 2;
                g[0] = re; g[1] = im;
            }
        };
        emit(r0, 4 * lane); emit(r1, 4 * lane + 1); emit(r2, 4 * lane + 2); emit(r3, 4 * lane + 3);
        __builtin_amdgcn_wave_barrier();
        // noise variance: the out-of-band energies added in carrier order (every lane computes the same value)
        double noise_sum = 0.0;
        int noise_bins = 0;
        for (int k = 0; k < Nc; ++k) {
            if (k >= band_start && k < band_end) continue;
            const double e = E[k];
            if (isfinite(e)) { noise_sum += e; ++noise_bins; }
        }
        double noise_var = noise_bins > 0 ? noise_sum / noise_bins : 1e-30;
        if (noise_var < 1e-30) noise_var = 1e-30;
        const double llr_scale = 1.0 / (2.0 * noise_var);
        if (lane < bps) {                                            // one lane per bit of the symbol period
            const int st = lane / nb, k = lane - st * nb;
            const int off = st == 0 ? T.mfsk_off0 : T.mfsk_off1;
            const int hop = (s * T.mfsk_hop) % M;
            const int mask = 1 << (nb - 1 - k);
            double max_E1 = -1e30, max_E0 = -1e30;
            for (int m = 0; m < M; ++m) {
                double e = E[off + ((m + hop) % M)];
                if (!isfinite(e)) e = 0.0;
                const int gray_m = m ^ (m >> 1);
                if (gray_m & mask) { if (e > max_E1) max_E1 = e; }
                else { if (e > max_E0) max_E0 = e; }
            }
            double llr = (max_E0 - max_E1) * llr_scale;
            if (!isfinite(llr)) llr = 0.0;
            else if (llr > 5.0) llr = 5.0;
            else if (llr < -5.0) llr = -5.0;
            const int idx = s * bps + lane;
            out[T.llr_dst[idx]] = float(llr);
            if (taps.llr_demod) taps.llr_demod[size_t(f) * T.nBits + idx] = float(llr);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
