// Device-side view of the constant tables (pointers into HBM; all tiny and L2-resident) plus the
// scalar mode parameters. Passed by value as a kernel argument.
#pragma once
#include <stdint.h>

struct MgpuDev {
    // front-end
    const uint8_t* cell_type;      // [G]
    const double* pilot_val;       // [G]
    const uint16_t* pilot_cell;    // [nPilots] row-major list of pilot cells
    const double* constellation;   // [M][2]
    const double* twiddle;         // [128][2]
    const double* pre_eq;          // [Nc][2] pre_equalization_channel the transmit path multiplies the carrier grid with, or NULL (none)
    const uint16_t* sym_src;       // [nData]
    const uint16_t* llr_src;       // [1600]
    const double* ls_weight;       // [lsw*lsw+1]
    const uint8_t* scrambler;      // [1600]
    const uint16_t* llr_dst;       // [nBits] MFSK modes: decoder position of demodulated LLR i
    // generator
    const uint16_t* bit_il;        // [nBits]
    const uint16_t* tf_inv;        // [nData] modulated-symbol index landing at de-framed position i
    const uint16_t* data_cell;     // [nData] grid cell of de-framed position i
    const uint32_t* cell_lerp;     // [nData][2] per data cell, built on the host from the pilot lattice (interpolate_linear_col, interpolator.cc:163-254):
                                   // word 0 = cell | pilot index of the upper row << 12 | of the lower row << 22 (pilot order, both in the cell's
                                   // column); word 1 = upper row | lower row << 8 | the cell's row << 16. OFDM modes; NULL otherwise
    // LDPC graph
    const uint32_t* cptr;          // [P+1] check-major edge list in the reference's row order
    const uint16_t* cvar;          // [E]   variable of edge e
    int S;
    int M, bps, K, P, N, E;
    int Nsymb, G, nData, nBits, nPilots, nVirtual, nReal;
    int estimator, amp_restore, lsw;
    int payload_bytes, payload_stride, frame_samples;
    int agc, var_eq, max_iters;
    int staircase;                 // LDPC parity part is the plain IRA staircase (check c holds parity c-1 and c only)
    int regular_lattice;           // 1: pilots exactly where (row - col) % 3 == 0 (true for all 17 modes); 2: and the LS window is wide enough for >= 3 pilots per row and residue
    double pilot_boost;
    float minsum_alpha;
    // MFSK modes (mfsk_M == 0 for the OFDM modes)
    int mfsk_M, mfsk_nbits, mfsk_nstreams, mfsk_hop, mfsk_off0, mfsk_off1;
    int active_nsymb, active_nbits;
    int puncture_from;             // demodulated LLRs from this position on are erasures: min(active_nbits, test_puncture_nBits)
    double mfsk_amp;
};

// The fp64 decoder's hard-frame shortcut (ldpc.hip: every |LLR| >= kSpaHardLlr -> the iterations are skipped) rests on three facts that are set
// in three places; they are tied together HERE so that changing one without the others does not compile:
//   * a variable has at most kSpaMaxVarDegree edges (tables.cpp refuses a graph with more: the unrolled variable update handles 9, and so does this);
//   * |R| <= kSpaClampR: the decoder clamps a product of +-1 to +-0.9999999 (ldpc_decoder_SPA.cc:150-156), 2 atanh(0.9999999) = 16.8114 (spa_math.h);
//   * tanh(Q/2) is +-1 exactly from |Q| = 44 on (s_tanh.c's |x| >= 22 answer).
// With them |Q| = |LLR + sum of the other R| >= kSpaHardLlr - kSpaMaxVarDegree * kSpaClampR >= 44 in every iteration: T = sign(LLR) for ever.
constexpr int kSpaMaxVarDegree = 9;
constexpr float kSpaHardLlr = 200.0f;
constexpr double kSpaClampR = 16.82;
constexpr double kSpaTanhSaturates = 44.0;
static_assert(double(kSpaHardLlr) - kSpaMaxVarDegree * kSpaClampR >= kSpaTanhSaturates,
              "hard-frame shortcut: |LLR| >= kSpaHardLlr no longer keeps every Q in tanh's saturated range - raise kSpaHardLlr or lower the degree limit");

// fp64 decoder, LDS layout: the first kSpaOnesBytes bytes of the workgroup's LDS hold 48 doubles 1.0 (at LDS address 0, so that the address is an
// inline constant of the select that forms a walk step's read address: ldpc.hip spa_walk); the posteriors follow, then the messages. The host's
// address tables (tables.cpp: sadr) carry absolute LDS byte addresses and are built with the same constant.
constexpr uint32_t kSpaOnesBytes = 48 * 8;

// Slim argument block for the decoder kernels: only what they touch, so the kernarg does not
// inflate the SGPR allocation (occupancy on gfx950 drops below 8 waves/SIMD above 80 SGPRs).
struct LdpcDev {
    const uint32_t* cptr; const uint16_t* cvar;                           // plain check-major lists (GBF)
    const uint8_t* scrambler;
    const uint16_t* crc_tab;   // [nReal/8][8] what message bit 8b+j set contributes to the CRC register after nReal/8 bytes (ldpc.hip: decode_tail)
    uint32_t crc_init;         // the register after nReal/8 zero bytes
    // fp32 decoders (sum-product and min-sum), grouped layout (tables.hpp: LdpcGraph::gdesc / gkpack / vinfo_g)
    const uint32_t* gdesc; const uint64_t* gkpack; const uint32_t* vinfo_g;
    int Sg;
    // fp64 sum-product kernel (tables.hpp: LdpcGraph::sadr / bhead / bmask / vinfo2)
    const uint32_t* sadr;    // [(NE+1)*1024][2] LDS byte offset of the slot's posterior, LDS byte address of the first message of the slot's check
    const uint64_t* bhead;   // [(NE+1)*16][2] per bin: lanes in use, lanes holding the last edge of a check
    const uint64_t* bmask;   // [NE*16][DM] execution mask of product-walk step j of a bin
    const uint32_t* vinfo2;  // vinfo with LDS byte offsets
    int DM;
    int S, N, P, K, E, nReal, payload_stride, max_iters;
    // fp64 decoder (ldpc.hip "adaptive"): a judged look that counts at least _min (_two) odd checks in bins 0..15 sends the next look (the next two)
    // into the check pass; ..0: the thresholds of the look at the channel's hard decisions
    // (one byte each, lowest first: min0, two0, min, two - ONE kernel argument: the compiler keeps every argument the loop uses in a scalar
    // register from the kernel's first instruction on, and the fp64 decoder's bin loop has none to spare)
    unsigned spec_sample_pack;
    float minsum_alpha;
    unsigned long long* hard_frames;   // [64] fp64 decoder: +1 (in counter frame % 64) per frame decided before its first iteration (every |LLR| >= 200 and an odd parity check), mgpu_decoder_hard_frames
};

struct MgpuTapsDev {
    double* grid; double* H; double* eq; double* syms; float* llr_demod; double* variance; double* agc_gain;
    long long* cycles;   // optional: s_memtime stamps at the phase boundaries of a frame from the middle of the batch (profiling aid)
    double* mean_H;      // optional: mean |H| over the pilot cells after the estimator (receive_byte's gate, telecom_system.cc:1224-1243)
};

struct MgpuStatsDev {  // must match mgpu_frame_stats
    int iterations_done, crc, all_zeros, message_decoded;
    float variance, snr_db;
};

// parameters of mgpu_mfsk_sync_kernel (mfsk.hip): preamble symbols, streams and their carrier offsets, the preamble's tones (mfsk.cc:82-95),
// carriers per slot, samples per symbol slot, samples a slot's FFT needs behind its start (Ngi + Nfft, interpolated)
struct MgpuMfskSync { int np, nstreams, off[4], tones[8], Nc, sym_period, tail; };
