// Post-decode SNR for the zero-forcing modes (CONFIG_15/16): the reference re-encodes the decoded
// bits, re-maps them and measures the error vector against the equalised data symbols
// (telecom_system.cc:1374-1396 -> cl_ofdm::measure_SNR, ofdm.cc:1622-1635):
//     bits -> bit_energy_dispersal -> virtual-bit copy -> ldpc.encode -> parity move -> interleaver
//          -> psk.mod -> interleaver(complex) ;  SNR = -10 log10( mean |mod_il[i] - deframed[i]|^2 )
// One workgroup per frame; the error terms are produced in parallel and summed by one lane in the
// reference's order (i = 0..nData-1, de-framed order) so the double result is reproducible.
// (For the LS modes the SNR is 10 log10(1/variance) and is produced by the decoder's epilogue.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"

#define ST_THREADS 256

extern "C" size_t mgpu_zfsnr_lds_bytes(int nData) { return 4 * 1600 + size_t(8) * nData + 64; }

extern "C" __global__ __launch_bounds__(ST_THREADS) void mgpu_zf_snr_kernel(
    MgpuDev T, const uint8_t* __restrict__ payload, const double* __restrict__ eqdata, int F,
    MgpuStatsDev* __restrict__ stats, double* __restrict__ var_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint8_t* bits = smem;               // K data bits (re-scrambled, virtual copy)
    uint8_t* enc = bits + 1600;         // N encoded bits
    uint8_t* inter = enc + 1600;        // nBits interleaved
    uint8_t* par = inter + 1600;
    __shared__ uint8_t wpar[32];        // info part of each parity check
    double* term = reinterpret_cast<double*>(par + 1600);
    const int tid = threadIdx.x, f = blockIdx.x;
    if (f >= F) return;
    if (!stats[f].message_decoded) return;          // SNR stays -99.9 (telecom_system.cc:1347)
    const int K = T.K, P = T.P, nReal = T.nReal;
    const uint8_t* pl = payload + size_t(f) * T.payload_stride;
    for (int i = tid; i < nReal; i += ST_THREADS) bits[i] = uint8_t(((pl[i >> 3] >> (i & 7)) & 1) ^ T.scrambler[i]);
    __syncthreads();
    for (int i = tid; i < T.nVirtual; i += ST_THREADS) bits[nReal + i] = bits[i];
    __syncthreads();
    for (int i = tid; i < K; i += ST_THREADS) enc[i] = bits[i];
    for (int c = tid; c < P; c += ST_THREADS) {
        uint8_t x = 0;
        for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) { const int v = T.cvar[e]; if (v < K) x ^= bits[v]; }
        par[c] = x;
    }
    __syncthreads();
    // IRA staircase: check c (c >= 1) holds, besides its own parity bit K+c, only parity bit K+c-1, so
    // parity[c] = XOR of par[0..c]: a prefix XOR done in three small steps (word parities, carries, per-bit)
    if (T.staircase) {
        const int nw = (P + 63) / 64;
        for (int k = tid; k < nw; k += ST_THREADS) {
            uint8_t x = 0;
            for (int c = k * 64; c < min(P, k * 64 + 64); ++c) x ^= par[c];
            wpar[k] = x;
        }
        __syncthreads();
        if (tid == 0) { uint8_t x = 0; for (int k = 0; k < nw; ++k) { const uint8_t y = wpar[k]; wpar[k] = x; x ^= y; } }
        __syncthreads();
        for (int c = tid; c < P; c += ST_THREADS) {
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
    for (int pos = tid; pos < T.nBits; pos += ST_THREADS) {
        const int src = T.bit_il[pos];
        inter[pos] = src < nReal ? enc[src] : enc[src + T.nVirtual];
    }
    __syncthreads();
    const double* eq = eqdata + size_t(f) * T.nData * 2;
    for (int i = tid; i < T.nData; i += ST_THREADS) {
        const int k = T.tf_inv[i];                  // modulated symbol that the TX interleaver puts at de-framed position i
        unsigned loc = 0;
        for (int j = 0; j < T.bps; ++j) loc = (loc << 1) | inter[k * T.bps + j];
        const double dr = eq[2 * i] - T.constellation[2 * loc], di = eq[2 * i + 1] - T.constellation[2 * loc + 1];
        term[i] = dr * dr + di * di;               // diff = in_n - in_s ; pow(re,2)+pow(im,2)
    }
    __syncthreads();
    if (tid == 0) {
        double var = 0;
        for (int i = 0; i < T.nData; ++i) var += term[i];
        var /= T.nData;
        stats[f].snr_db = float(-10.0 * log10(var));
        if (var_out) var_out[f] = var;             // receive_byte's double SNR takes the logarithm on the host (the reference's libm)
    }
}

// cl_error_rate::check (error_rate.cc:48-70) for F frames at once: bits that differ between the sent and the decoded payload
// (nReal bits per frame, packed LSB first as the payload is; the scrambler is a bijection, so payload bits differ exactly where the
// reference's data_bit / hd_decoded_data_bit differ), frames with at least one such bit, and the iteration counts, added into
// acc[0..3] = {bit errors, frame errors, iterations, frames whose CRC self-check passed}. One lane per frame.
extern "C" __global__ __launch_bounds__(256) void mgpu_error_count_kernel(
    const uint8_t* __restrict__ sent, const uint8_t* __restrict__ got, const MgpuStatsDev* __restrict__ stats, int stride, int nReal, int F,
    unsigned long long* __restrict__ acc) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long be = 0, fe = 0, it = 0, ok = 0;
    if (f < F) {
        const uint8_t* a = sent + size_t(f) * stride;
        const uint8_t* b = got + size_t(f) * stride;
        const int full = nReal >> 3, rem = nReal & 7;
        unsigned e = 0;
        for (int i = 0; i < full; ++i) e += __popc(unsigned(a[i] ^ b[i]));
        if (rem) e += __popc(unsigned((a[full] ^ b[full]) & ((1u << rem) - 1u)));
        be = e; fe = e != 0; it = unsigned(stats[f].iterations_done); ok = stats[f].message_decoded != 0;
    }
    // wave reduction, then one atomic per wave and counter
    for (int d = 32; d >= 1; d >>= 1) {
        be += __shfl_xor(be, d); fe += __shfl_xor(fe, d); it += __shfl_xor(it, d); ok += __shfl_xor(ok, d);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&acc[0], be); atomicAdd(&acc[1], fe); atomicAdd(&acc[2], it); atomicAdd(&acc[3], ok);
    }
}
