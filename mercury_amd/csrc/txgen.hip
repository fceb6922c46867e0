// The transmit chain from message bytes to baseband samples, in two roles: (1) the synthetic-workload generator (NOT part
// of the RX path; bench.py and the scale tests use it so that inputs are born in HBM), and (2) with the caller's bytes and the
// channel switched off, the first stage of transmit_byte (tx.hip, include/mercury_tx.h). Philox-keyed payload -> CRC16 -> scrambler -> IRA LDPC encode ->
// bit interleave -> constellation map -> time/freq interleave -> framer (pilots) -> IFFT + GI ->
// channel (AWGN, optional static 2-path). It mirrors the reference TX chain
// (telecom_system.cc:343-382 transmit_byte, :384-470 transmit_bit; ldpc.cc:111-132 encode;
// psk.cc:259-272 mod or, for the MFSK modes, mfsk.cc:232-285 mod; ofdm.cc:814-835 framer, :855-860
// symbol_mod) and the AWGN scaling of
// baseband_test_EsN0 (telecom_system.cc:141-153). Generator definition: DESIGN.md §Synthetic inputs;
// the CPU twin used to validate it is oracle/mercury_oracle.c:morc_gen_frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"
#include "fft256.h"

#define TX_THREADS 256
#define TX_WAVES (TX_THREADS / 64)

namespace {

__device__ void philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2_, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = uint32_t(seed), k1 = uint32_t(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = __umulhi(0xCD9E8D57u, c2_), l1 = 0xCD9E8D57u * c2_;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2_ = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2_; out[3] = c3;
}

__device__ __forceinline__ double gauss_bm(uint32_t a, uint32_t b) {
    const double u1 = (double(a) + 1.0) * (1.0 / 4294967296.0);
    const double u2 = double(b) * (1.0 / 4294967296.0);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

}  // namespace

// LDS: bits 1600 | enc 1600 | inter 1600 | par 1600 (bytes) | grid 16*G (G = 0 for MFSK) | fft 4*272*16 | tw 128*16 | pay 256
extern "C" size_t mgpu_txgen_lds_bytes(int G) { return 4 * 1600 + size_t(16) * G + TX_WAVES * FFT256_STRIDE * 16 + 128 * 16 + 256 + 64; }

extern "C" __global__ __launch_bounds__(TX_THREADS) void mgpu_txgen_kernel(
    MgpuDev T, uint64_t seed, uint64_t frame0, int F, double noise_amp, int channel,
    double* __restrict__ baseband, uint8_t* __restrict__ payload_out,
    const uint8_t* __restrict__ payload_in, int payload_in_stride, const int* __restrict__ nbytes_in, int out_stride, int out_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* bits = smem;               // data bits (scrambled, with virtual copy): K entries
    uint8_t* enc = bits + 1600;         // encoded word N
    uint8_t* inter = enc + 1600;        // interleaved nBits
    uint8_t* par = inter + 1600;        // info-part parity per check
    __shared__ uint8_t wpar[32];        // parity of each 64-check word (prefix-XOR carries)
    const bool mfsk = T.mfsk_M > 0;
    c2* grid = reinterpret_cast<c2*>(par + 1600);
    c2* fftb = grid + (mfsk ? 0 : T.G);
    c2* tw = fftb + TX_WAVES * FFT256_STRIDE;
    uint8_t* pay = reinterpret_cast<uint8_t*>(tw + 128);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (int(blockIdx.x) >= F) return;
    const uint64_t frame = frame0 + blockIdx.x;
    const uint32_t flo = uint32_t(frame), fhi = uint32_t(frame >> 32);
    const int K = T.K, P = T.P, nReal = T.nReal, fs = T.payload_bytes;
    // out_stride / out_offset (complex samples): where frame f starts; the transmit path leaves room for the preamble
    c2* out = reinterpret_cast<c2*>(baseband) + size_t(blockIdx.x) * (out_stride ? out_stride : T.frame_samples) + out_offset;

    for (int i = tid; i < 128; i += TX_THREADS) tw[fft256_tw_slot(i)] = {T.twiddle[2 * i], -T.twiddle[2 * i + 1]};  // conj (ofdm.cc:365)
    const bool raw_bits = payload_in && payload_in_stride < 0;   // transmit_bit: the caller's nReal data bits, one byte each, rows of -stride
    if (raw_bits) {                     // telecom_system.cc:384-396: no CRC, no padding - the bits as they are
        const uint8_t* in = payload_in + size_t(blockIdx.x) * size_t(-payload_in_stride);
        for (int j = tid; j < 200; j += TX_THREADS) pay[j] = 0;
        __syncthreads();
        for (int i = tid; i < nReal; i += TX_THREADS) bits[i] = (in[i] & 1) ^ T.scrambler[i];   // bit_energy_dispersal
        if (tid < (nReal + 7) / 8) {                                                              // the expected RX bytes, LSB first
            uint8_t v = 0;
            for (int k = 0; k < 8 && tid * 8 + k < nReal; ++k) v |= uint8_t(in[tid * 8 + k] & 1) << k;
            pay[tid] = v;
        }
        __syncthreads();
    } else {
    if (payload_in) {                   // transmit_byte: the caller's message, zero-padded to the frame (telecom_system.cc:354-366)
        const int nb = nbytes_in ? nbytes_in[blockIdx.x] : fs;
        for (int j = tid; j < fs; j += TX_THREADS) pay[j] = j < nb ? payload_in[size_t(blockIdx.x) * payload_in_stride + j] : uint8_t(0);
    } else {
        for (int j = tid; j < fs; j += TX_THREADS) {
            uint32_t w[4];
            philox4x32(seed, uint32_t(j >> 4), 0u, flo, fhi, w);
            pay[j] = uint8_t(w[(j >> 2) & 3] >> (8 * (j & 3)));
        }
    }
    __syncthreads();
    if (tid == 0) {                     // CRC16 over the payload (telecom_system.cc:365-372)
        unsigned crc = 0xffff;
        for (int j = 0; j < fs; ++j) {
            crc ^= pay[j];
            for (int i = 0; i < 8; ++i) crc = (crc & 1) ? ((crc >> 1) ^ 0xA001) : (crc >> 1);
        }
        pay[fs] = uint8_t(crc & 0xff);
        pay[fs + 1] = uint8_t(crc >> 8);
    }
    __syncthreads();
    for (int i = tid; i < nReal; i += TX_THREADS) {
        const int byte = i >> 3;
        const uint8_t b = byte < fs + 2 ? uint8_t((pay[byte] >> (i & 7)) & 1) : uint8_t(0);
        bits[i] = b ^ T.scrambler[i];   // bit_energy_dispersal
    }
    }
    if (payload_out)
        for (int b = tid; b < T.payload_stride; b += TX_THREADS) {
            // expected RX bytes: the nReal (un-scrambled) data bits packed LSB first
            uint8_t v = (raw_bits || b < fs + 2) ? pay[b] : uint8_t(0);
            if ((b + 1) * 8 > nReal) v &= uint8_t((1u << (nReal - b * 8)) - 1);
            payload_out[size_t(blockIdx.x) * T.payload_stride + b] = v;
        }
    __syncthreads();
    for (int i = tid; i < T.nVirtual; i += TX_THREADS) bits[nReal + i] = bits[i];
    __syncthreads();
    // IRA encode (ldpc.cc:111-132): parity i = XOR of the other entries of check row i
    for (int i = tid; i < K; i += TX_THREADS) enc[i] = bits[i];
    for (int c = tid; c < P; c += TX_THREADS) {
        uint8_t x = 0;
        for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) { const int v = T.cvar[e]; if (v < K) x ^= bits[v]; }
        par[c] = x;
    }
    __syncthreads();
    // IRA staircase: check c (c >= 1) holds, besides its own parity bit K+c, only parity bit K+c-1, so
    // parity[c] = XOR of par[0..c]: a prefix XOR done in three small steps (word parities, carries, per-bit)
    if (T.staircase) {
        const int nw = (P + 63) / 64;
        for (int k = tid; k < nw; k += TX_THREADS) {
            uint8_t x = 0;
            for (int c = k * 64; c < min(P, k * 64 + 64); ++c) x ^= par[c];
            wpar[k] = x;
        }
        __syncthreads();
        if (tid == 0) { uint8_t x = 0; for (int k = 0; k < nw; ++k) { const uint8_t y = wpar[k]; wpar[k] = x; x ^= y; } }
        __syncthreads();
        for (int c = tid; c < P; c += TX_THREADS) {
            uint8_t x = wpar[c >> 6];
            for (int q = c & ~63; q <= c; ++q) x ^= par[q];
            enc[K + c] = x;
        }
    } else if (tid == 0) {
        for (int c = 0; c < P; ++c) {
            uint8_t x = par[c];
            for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) { const int v = T.cvar[e]; if (v >= K && v != K + c) x ^= enc[v]; }
            enc[K + c] = x;
        }
    }
    __syncthreads();
    // parity moved down over the virtual bits, then bit interleaver (gather form)
    for (int pos = tid; pos < T.nBits; pos += TX_THREADS) {
        const int src = T.bit_il[pos];
        inter[pos] = src < nReal ? enc[src] : enc[src + T.nVirtual];
    }
    __syncthreads();
    // pilots + mapped data into the frame grid (the MFSK modes build each symbol's carriers on the fly)
    if (!mfsk) {
        for (int c = tid; c < T.G; c += TX_THREADS) if (T.cell_type[c]) grid[c] = {T.pilot_val[c], 0.0};
        for (int k = tid; k < T.nData; k += TX_THREADS) {
            unsigned loc = 0;
            for (int j = 0; j < T.bps; ++j) loc = (loc << 1) | inter[k * T.bps + j];
            grid[T.sym_src[k]] = {T.constellation[2 * loc], T.constellation[2 * loc + 1]};
        }
        if (channel < 0 && T.pre_eq) {      // transmit_bit only (telecom_system.cc:486-493): ofdm_framed_data[i*Nc+j] *= pre_equalization_channel[j]
            __syncthreads();
            for (int c = tid; c < T.G; c += TX_THREADS) {
                const c2 g = grid[c], h = {T.pre_eq[2 * (c % 50)], T.pre_eq[2 * (c % 50) + 1]};
                grid[c] = {g.re * h.re - g.im * h.im, g.re * h.im + g.im * h.re};      // std::complex operator*= on finite values
            }
        }
    }
    __syncthreads();
    // symbol_mod: zero_padder + unnormalised IFFT (fft256.h with conjugated twiddles) + gi_adder, one wave per symbol
    for (int s = wave; s < T.active_nsymb; s += TX_WAVES) {
        int tone0 = -1, tone1 = -1;                 // MFSK: active carrier of each stream this symbol (mfsk.cc:252-282)
        if (mfsk) {
            for (int st = 0; st < T.mfsk_nstreams; ++st) {
                int tone = 0;
                for (int b = 0; b < T.mfsk_nbits; ++b)
                    if (inter[s * T.bps + st * T.mfsk_nbits + b]) tone |= 1 << (T.mfsk_nbits - 1 - b);
                int bin = tone;
                for (int sh = 1; sh < T.mfsk_nbits; ++sh) bin ^= tone >> sh;          // Gray -> binary
                if (bin >= T.mfsk_M) bin = T.mfsk_M - 1;
                const int actual = (bin + s * T.mfsk_hop) & (T.mfsk_M - 1);           // tone hopping (M is 16 or 32)
                if (st == 0) tone0 = T.mfsk_off0 + actual; else tone1 = T.mfsk_off1 + actual;
            }
        }
        auto carrier = [&](int i) -> c2 {            // zero_padder (ofdm.cc:379-400): FFT input bin i
            const int col = carrier_of_bin(i);
            if (col < 0) return {0.0, 0.0};
            if (mfsk) return (col == tone0 || col == tone1) ? c2{T.mfsk_amp, 0.0} : c2{0.0, 0.0};
            return grid[s * 50 + col];
        };
        c2 r0 = carrier(lane), r1 = carrier(lane + 64), r2 = carrier(lane + 128), r3 = carrier(lane + 192);
        wave_fft256(r0, r1, r2, r3, fftb + wave * FFT256_STRIDE, tw, lane);
        c2* y = out + size_t(s) * 272;
        auto emit = [&](const c2& x, int p) {        // gi_adder (ofdm.cc:412-422): sample n and its cyclic prefix copy
            const int n = brev8(p);
            y[n + 16] = x;
            if (n >= 240) y[n - 240] = x;
        };
        emit(r0, 4 * lane); emit(r1, 4 * lane + 1); emit(r2, 4 * lane + 2); emit(r3, 4 * lane + 3);
    }
    if (channel < 0) return;            // transmit path: no channel
    __syncthreads();
    // channel, back to front in chunks so the 6-sample echo always reads clean samples
    const int n = T.frame_samples;
    c2 h1 = {0.0, 0.0};
    if (channel == 1) {
        uint32_t w[4];
        philox4x32(seed, 0u, 2u, flo, fhi, w);
        const double phi = 2.0 * M_PI * (double(w[0]) * (1.0 / 4294967296.0));
        h1 = {0.5 * cos(phi), 0.5 * sin(phi)};
    }
    for (int base = ((n - 1) / TX_THREADS) * TX_THREADS; base >= 0; base -= TX_THREADS) {
        const int i = base + tid;
        c2 x = {0.0, 0.0};
        if (i < n) {
            x = out[i];
            if (channel == 1 && i >= 6) { const c2 e = cmul(h1, out[i - 6]); x = {x.re + e.re, x.im + e.im}; }
            uint32_t w[4];
            philox4x32(seed, uint32_t(i), 1u, flo, fhi, w);
            const double nr = noise_amp * gauss_bm(w[0], w[1]), ni = noise_amp * gauss_bm(w[2], w[3]);
            x = {(x.re / 16.0 + nr) * 16.0, (x.im / 16.0 + ni) * 16.0};
        }
        __syncthreads();
        if (i < n) out[i] = x;
        __syncthreads();
    }
}

// cl_ldpc::encode alone (ldpc.cc:111-132) for F words: bits [F][K] (one byte per bit) -> enc [F][N]. Same two steps as in the
// transmit kernel above: the information part of every check's parity in parallel, then the staircase as a prefix XOR.
extern "C" __global__ __launch_bounds__(TX_THREADS) void mgpu_ldpc_encode_kernel(MgpuDev T, const uint8_t* __restrict__ bits_in, int F,
                                                                               uint8_t* __restrict__ enc_out) {
    __shared__ uint8_t bits[1600], par[1600], wpar[32];
    const int tid = threadIdx.x, K = T.K, P = T.P;
    if (int(blockIdx.x) >= F) return;
    const uint8_t* in = bits_in + size_t(blockIdx.x) * K;
    uint8_t* enc = enc_out + size_t(blockIdx.x) * (K + P);
    for (int i = tid; i < K; i += TX_THREADS) { const uint8_t b = in[i] & 1; bits[i] = b; enc[i] = b; }
    __syncthreads();
    for (int c = tid; c < P; c += TX_THREADS) {
        uint8_t x = 0;
        for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) { const int v = T.cvar[e]; if (v < K) x ^= bits[v]; }
        par[c] = x;
    }
    __syncthreads();
    if (T.staircase) {
        const int nw = (P + 63) / 64;
        for (int k = tid; k < nw; k += TX_THREADS) {
            uint8_t x = 0;
            for (int c = k * 64; c < min(P, k * 64 + 64); ++c) x ^= par[c];
            wpar[k] = x;
        }
        __syncthreads();
        if (tid == 0) { uint8_t x = 0; for (int k = 0; k < nw; ++k) { const uint8_t y = wpar[k]; wpar[k] = x; x ^= y; } }
        __syncthreads();
        for (int c = tid; c < P; c += TX_THREADS) {
            uint8_t x = wpar[c >> 6];
            for (int q = c & ~63; q <= c; ++q) x ^= par[q];
            enc[K + c] = x;
        }
    } else {                          // general lower-triangular parity part: rows in order, as the reference does
        __shared__ uint8_t pbit[1600];
        if (tid == 0)
            for (int c = 0; c < P; ++c) {
                uint8_t x = par[c];
                for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) { const int v = T.cvar[e]; if (v >= K && v != K + c) x ^= pbit[v - K]; }
                pbit[c] = x;
            }
        __syncthreads();
        for (int c = tid; c < P; c += TX_THREADS) enc[K + c] = pbit[c];
    }
}

// ---- helpers of the passband self-simulation (cl_telecom_system::passband_test_EsN0, telecom_system.cc:231-330) -----------------
// the generator's payload bytes of frames frame0 .. frame0+F-1 (the same Philox stream 0 the frame generator above draws them from)
extern "C" __global__ __launch_bounds__(256) void mgpu_gen_payload_kernel(uint64_t seed, uint64_t frame0, int F, int nbytes, int stride,
                                                                          uint8_t* __restrict__ out) {
    const int f = blockIdx.x;
    if (f >= F) return;
    const uint64_t fr = frame0 + uint64_t(f);
    const uint32_t flo = uint32_t(fr), fhi = uint32_t(fr >> 32);
    for (int j = threadIdx.x; j < stride; j += blockDim.x) {
        uint8_t v = 0;
        if (j < nbytes) {
            uint32_t w[4];
            philox4x32(seed, uint32_t(j >> 4), 0u, flo, fhi, w);
            v = uint8_t(w[(j >> 2) & 3] >> (8 * (j & 3)));
        }
        out[size_t(f) * stride + j] = v;
    }
}

// cl_awgn::apply_with_delay (awgn.cc:65-77) on real audio, one capture window per frame: `delay` samples of randomly picked signal
// samples + noise, the frame + noise, and — where the reference's buffer keeps whatever an earlier frame left there — noise alone up to
// the end of the window. Noise = ampl * N(0,1) from Philox stream 3 (counter = sample index, frame).
extern "C" __global__ __launch_bounds__(256) void mgpu_passband_channel_kernel(const double* __restrict__ audio, int total, int delay, int window,
                                                                               double ampl, uint64_t seed, uint64_t frame0, int F,
                                                                               double* __restrict__ out) {
    const int f = blockIdx.y;
    if (f >= F) return;
    const uint64_t fr = frame0 + uint64_t(f);
    const uint32_t flo = uint32_t(fr), fhi = uint32_t(fr >> 32);
    const double* a = audio + size_t(f) * total;
    double* o = out + size_t(f) * window;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < window; i += gridDim.x * blockDim.x) {
        uint32_t w[4];
        philox4x32(seed, uint32_t(i), 3u, flo, fhi, w);
        double x = 0.0;
        if (i < delay) x = a[w[2] % uint32_t(total)];
        else if (i < delay + total) x = a[i - delay];
        o[i] = x + ampl * gauss_bm(w[0], w[1]);
    }
}

