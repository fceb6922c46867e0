// LDPC belief-propagation decoders for Mercury's N=1600 IRA codes on gfx950.
//
// Layout decision: ONE WORKGROUP PER CODEWORD, all messages resident in LDS.
// A codeword's whole Tanner graph state (E <= 6604 edges) is 53 KB in fp64 / 26 KB in fp32, so it
// never needs to round-trip through HBM: HBM sees 1600 LLR floats in and ~200 B out per frame.
// Early termination is per workgroup, so converged frames cost nothing further and there is no
// divergence between frames. Graph index tables (<60 KB per rate) are shared by every frame and
// stay L2-resident.
//
// Decoders:
//   spa    — the reference's flooding sum-product decoder in double precision
//            (ldpc_decoder_SPA.cc:25-218), restructured edge-parallel but performing the same
//            arithmetic per message: tanh(0.5*Q) once per edge, the check product over the other
//            edges in ascending row order starting from 1.0, the +-1 clamp to +-0.9999999,
//            2*atanh, and the variable sum llr + R[slot 0] + R[slot 1] + ... in slot order.
//   minsum — normalised min-sum in fp32 (not in the reference; BASELINE.json north_star variant).
//   gbf    — gradient bit flipping (ldpc_decoder_GBF.cc:25-117), float, bit-exact.
//
// The tail (bit_energy_dispersal interleaver.cc:111-117, bit_to_byte misc.cc:107-130, all-zeros
// test and CRC16 telecom_system.cc:1319-1345, crc16_modbus_rtu.cc:25-45) is fused into the
// decoder's epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_tables.h"
// Profiling aid (variant builds only: tools/build_variants.sh census:"-DSPA_CENSUS_ON=1"): every case branch of spa_math.h counts the
// wavefronts that enter it (first active lane) and the lanes they enter it with; tools/spa_census.py reads the counters back through
// mgpu_debug_spa_census and prints what a wavefront of the launch actually executes. The product build carries none of it.
#ifdef SPA_CENSUS_ON
#include <hip/hip_runtime.h>
__device__ unsigned long long g_spa_census[64];
#define SPA_CENSUS(i) do { const unsigned long long em_ = __builtin_amdgcn_ballot_w64(true); \
    if (int(threadIdx.x & 63) == __builtin_ctzll(em_)) { atomicAdd(&g_spa_census[i], 1ull); atomicAdd(&g_spa_census[32 + (i)], (unsigned long long)__builtin_popcountll(em_)); } } while (0)
extern "C" int mgpu_debug_spa_census(unsigned long long* out, int clear) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spa_census), sizeof(g_spa_census)) != hipSuccess) return -1;
    if (clear) { static unsigned long long z[64]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_spa_census), z, sizeof(z)) != hipSuccess) return -1; }
    return 64;
}
#endif
#include "spa_math.h"

#define LDPC_THREADS 1024

// Profiling aid (variant builds only: tools/build_variants.sh name:"-DSPA_STAMPS=1"): lane 0 of every wavefront of eight workgroups spread
// over the launch appends (code << 56 | s_memrealtime: 100 MHz ticks) at the decoder's phase boundaries; tools/spa_stamps.py reads them back
// through mgpu_debug_spa_stamps. The product build carries none of it.
#ifdef SPA_STAMPS
#define SPA_STAMP_WGS 8
#define SPA_STAMP_MAX 192
__device__ unsigned long long g_spa_stamps[SPA_STAMP_WGS * 16 * SPA_STAMP_MAX];
#define SPA_STAMP_DECL(F_) int stamp_n = 0; const int stamp_wg = (blockIdx.x % ((F_) / SPA_STAMP_WGS > 0 ? (F_) / SPA_STAMP_WGS : 1)) == ((F_) / (2 * SPA_STAMP_WGS)) ? int(blockIdx.x / ((F_) / SPA_STAMP_WGS > 0 ? (F_) / SPA_STAMP_WGS : 1)) : -1
#define SPA_STAMP(code) do { if (stamp_wg >= 0 && stamp_wg < SPA_STAMP_WGS && (threadIdx.x & 63) == 0 && stamp_n < SPA_STAMP_MAX) { \
    g_spa_stamps[(stamp_wg * 16 + (threadIdx.x >> 6)) * SPA_STAMP_MAX + stamp_n] = (static_cast<unsigned long long>(code) << 56) | (__builtin_amdgcn_s_memrealtime() & 0x00ffffffffffffffull); } ++stamp_n; } while (0)
extern "C" int mgpu_debug_spa_stamps(unsigned long long* out, int clear) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spa_stamps), sizeof(g_spa_stamps)) != hipSuccess) return -1;
    if (clear) { static unsigned long long z[SPA_STAMP_WGS * 16 * SPA_STAMP_MAX]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_spa_stamps), z, sizeof(z)) != hipSuccess) return -1; }
    return SPA_STAMP_WGS * 16 * SPA_STAMP_MAX;
}
#else
#define SPA_STAMP_DECL(F_) do {} while (0)
#define SPA_STAMP(code) do {} while (0)
#endif
#ifndef SPA_SPEC_START
#define SPA_SPEC_START 8
#endif
#ifndef SPA_VR_RESIDENT
#define SPA_VR_RESIDENT 1
#endif

namespace {

// Epilogue shared by all decoders: hard[] (N bytes in LDS) -> bits / payload / stats. bytes_lds: 256 bytes + 16 ints of scratch.
// The CRC (crc16_modbus_rtu.cc:25-45 over the first nReal/8 bytes, telecom_system.cc:1319-1345) is linear over GF(2): the register
// after the message = the register after an all-zero message of that length (crc_init) xor the contributions of the message's set
// bits, each a constant of the bit's position (crc_tab, built on the host by running the bitwise algorithm on single-bit messages).
// Every byte's lane looks up its eight constants, a wavefront xor-reduces them: a dozen instructions instead of one lane's
// 600 dependent shift steps (10 us per frame, a third of a frame's decode time at the operating points).
__device__ void decode_tail(const LdpcDev& T, int f, const uint8_t* hard, uint8_t* bytes_lds, int iterations,
                            uint8_t* __restrict__ bits_out, int* __restrict__ iters_out,
                            uint8_t* __restrict__ payload_out, MgpuStatsDev* __restrict__ stats_out,
                            const float* __restrict__ variance_in, const float* __restrict__ snr_variance_in) {
    const int tid = threadIdx.x;
    if (bits_out) for (int i = tid; i < T.K; i += blockDim.x) bits_out[size_t(f) * T.K + i] = hard[i];
    if (iters_out && tid == 0) iters_out[f] = iterations;
    if (!payload_out && !stats_out) return;
    const int nb = T.nReal, nbytes = (nb + 7) / 8, full = nb / 8;
    unsigned x = 0, nz = 0;
    for (int b = tid; b < nbytes; b += blockDim.x) {
        unsigned v = 0;
        for (int j = 0; j < 8; ++j) {
            const int i = b * 8 + j;
            if (i < nb) v |= unsigned(hard[i] ^ T.scrambler[i]) << j;
        }
        if (payload_out) payload_out[size_t(f) * T.payload_stride + b] = uint8_t(v);
        if (b < full) {
            nz |= v;
            const uint4 t = reinterpret_cast<const uint4*>(T.crc_tab)[b];
            x ^= (v & 1 ? t.x & 0xffffu : 0u) ^ (v & 2 ? t.x >> 16 : 0u) ^ (v & 4 ? t.y & 0xffffu : 0u) ^ (v & 8 ? t.y >> 16 : 0u)
               ^ (v & 16 ? t.z & 0xffffu : 0u) ^ (v & 32 ? t.z >> 16 : 0u) ^ (v & 64 ? t.w & 0xffffu : 0u) ^ (v & 128 ? t.w >> 16 : 0u);
        }
    }
    if (!stats_out) return;
    int* scratch = reinterpret_cast<int*>(bytes_lds + 256);           // one word per wavefront (the decoders' flags are done with)
    for (int o = 32; o; o >>= 1) x ^= unsigned(__shfl_xor(int(x), o));
    x |= __any(nz != 0) ? 0x10000u : 0u;
    if ((tid & 63) == 0) scratch[tid >> 6] = int(x);
    __syncthreads();
    if (tid == 0) {
        unsigned acc = 0;
        for (int w = 0; w < int(blockDim.x >> 6); ++w) { const unsigned y = unsigned(scratch[w]); acc = ((acc ^ y) & 0xffffu) | ((acc | y) & 0x10000u); }
        const int all_zeros = !(acc & 0x10000u);
        const unsigned crc = all_zeros ? 0u : ((acc & 0xffffu) ^ T.crc_init);
        MgpuStatsDev s;
        s.iterations_done = iterations;
        s.crc = int(crc);
        s.all_zeros = all_zeros;
        s.message_decoded = !(all_zeros || crc != 0);
        s.variance = variance_in ? variance_in[f] : 0.0f;
        // telecom_system.cc:1347 / :1366-1372: 10*log10(1/variance) with the float variance widened
        const float sv = snr_variance_in ? snr_variance_in[f] : s.variance;
        s.snr_db = s.message_decoded ? float(10.0 * log10(1.0 / double(sv))) : -99.9f;
        stats_out[f] = s;
    }
}

}  // namespace

extern "C" size_t mgpu_spa_lds_bytes(int S, int N) {
    return size_t(kSpaOnesBytes) + size_t(8) * S + size_t(8) * N + size_t(4) * N + ((N + 15) & ~15) + 256 + 64;
}

// ---------------------------------------------------------------------------------------------
// Sum-product, double precision, reference arithmetic.
//
// One message array M[] in LDS holds, alternately, R and T = tanh(0.5*Q) for every edge. Edges
// live in a padded, wave-private layout: whole checks are bin-packed into 64-slot bins, and a bin
// is always processed by one wavefront in one instruction stream, so everything that happens to a
// check between two variable updates needs no workgroup barrier:
//     per bin:  Q = posterior - R  ->  T = tanh(0.5*Q) -> M   (own T stays in a register)
//               wave barrier (every lane's T is in LDS)
//               product of the check's other T values, in slot order -> clamp -> R = 2*atanh
//               wave barrier (every lane has read its check's T values)
//               R -> M
// The syndrome costs nothing extra: in that pass every lane holds the posterior of its edge's
// variable, one 64-bit ballot of the sign bits gives every check its parity.
// Per iteration: [syndrome +] tanh + check update | barrier | variable update | barrier [| syndrome | barrier | verdict].
//
// Instruction budget (profiles/r03_instruction_mix.json, r02_valu_cycles.json): a wavefront alone keeps its SIMD busy about a
// quarter of the time and eight of them reach 81 %, so every instruction of the loop counts, scalar ones included. Per edge and
// iteration the reference's arithmetic needs ~105 fp64 operations + 5 reciprocal seeds (spa_math.h). Round 3 moved what was not
// that arithmetic out of the vector unit (174 -> 157 vector instructions per 64-slot bin-iteration, 2.89 G per headline launch):
//   * per-slot addresses come from a table (LdpcGraph::sadr: LDS offset of the slot's posterior, LDS address of its check's first
//     message), one buffer load per round with the round offset in a scalar: no field extraction, no lane needs its degree or its
//     position inside the check;
//   * what is uniform over a bin comes through scalar loads from constant-address-space pointers: the lanes in use and the lanes
//     ending a check (bhead) and, per product-walk step, the lanes that multiply that factor in (bmask: position != step and
//     degree > step, tabulated on the host). "valid" is the bin's lane mask applied as exec (inverse ballot); a walk step is one
//     LDS broadcast read + one v_mul_f64 under s_and_b64 exec - no compare; the walk ends at the first all-zero mask;
//   * padding lanes and fdlibm's case distinctions are execution-mask branches (scalar instructions only), not selects;
//   * variable records carry LDS byte offsets and come through buffer loads (rows past N read as zeros = degree 0).
constexpr int kN = 1600;             // every Mercury code has N = 1600 (checked on the host); fixes the LDS layout below

__device__ __forceinline__ void spa_masked_mul2(double& t, double a0, double a1, uint64_t m0, uint64_t m1) {
    uint64_t sv;
    asm volatile(
        "s_and_saveexec_b64 %1, %4\n\t"
        "v_mul_f64 %0, %0, %2\n\t"
        "s_and_b64 exec, %1, %5\n\t"
        "v_mul_f64 %0, %0, %3\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(t), "=&s"(sv)
        : "v"(a0), "v"(a1), "s"(m0), "s"(m1)
        : "scc");
}

typedef uint32_t spa_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t spa_u32x2 __attribute__((ext_vector_type(2)));
// The bin tables are read through constant-address-space pointers: a uniform load from there is a scalar load whatever
// barriers the loop contains (through a global pointer every load behind a workgroup barrier counts as clobbered and takes
// the vector memory path), and the compiler keeps track of the outstanding loads itself.
typedef const uint64_t __attribute__((address_space(4))) * spa_cptr64;
// ... at a 32-bit BYTE offset from a table's base: the scalar load then takes base + offset register + immediate as it stands, where a 64-bit
// element index is a multiplication in two halves, a shift and an add-with-carry in front of it (SPA_OFF32 0: the round 2-5 form)
#ifndef SPA_OFF32
#define SPA_OFF32 1
#endif
#ifndef SPA_ALLLANES
#define SPA_ALLLANES 1
#endif
#ifndef SPA_ALLLANES8
#define SPA_ALLLANES8 0      /* the same at rate 14/16 (the wavefront-wide shortcuts then look at the bin's real lanes only): bit-exact, 30 scalar instructions fewer per bin, and 0.3 - 1.2 % SLOWER (profiles/r06_ab_al8.txt) - a quarter of that kernel's lanes are padding, and what they compute along drags wavefronts into branches */
#endif
typedef const char __attribute__((address_space(4))) * spa_cptr8;
__device__ __forceinline__ spa_cptr64 spa_at(const void* base, uint32_t byte_off) {
    return (spa_cptr64)((spa_cptr8)(base) + byte_off);
}

// The product walk. Step j multiplies slot j of the lane's check into the lane's product unless j is the lane's own slot, lies past the
// check's degree, or the lane is padding: the bin's tabulated lane mask for step j (bmask, a scalar load) says which lanes take the factor.
// Two ways of applying it:
//  * as the execution mask of the multiplication (spa_masked_mul2: s_and_saveexec / v_mul / s_and / v_mul / s_mov per step pair) - scalar-unit work;
//  * on the vector side (spa_walk_factor, round 6): a lane that does not take the factor reads a 1.0 instead - its read address is selected (one
//    v_cndmask_b32 with the mask as its condition operand) between the check's first message and LDS address 0, where 48 doubles 1.0 stand
//    (the step's offset 8 j is the read's immediate) - and every lane multiplies: x * 1.0 == x exactly, so a lane's product is the same sequence
//    of roundings.
// Which one is cheaper depends on which unit the kernel loads more. Round 6 measured what the scalar unit is worth here (profiles/NOTES.md R6.2:
// twelve more s_mov_b32 per bin-iteration +4 % on the headline, twelve more v_mov_b32 +1 %, 24 s_nop nothing; the CU's one scalar unit is 67 %
// busy at rate 6/16 and 84 % at rate 14/16 in noise). The short walks (rates 1/16 .. 8/16: 5-8 steps) stay on the scalar side (the vector form
// measured +-0 at rate 6/16 in noise, +0.4 % at -1 dB, +1 % on mode 11); the 46-step walk of rate 14/16 splits its steps between the two (spa_walk4).
// Steps 2C and 2C+1 and, recursively, the rest (unrolled by construction: every step's LDS offset is an immediate): masks m0 / m1 belong to this
// pair, the next pair's are fetched behind it; an all-zero mask ends the walk.
__device__ __forceinline__ double spa_walk_factor(uint32_t achk, uint64_t m, int step) {
    typedef __attribute__((address_space(3))) double lds_f64;
    uint32_t base = __builtin_amdgcn_inverse_ballot_w64(m) ? achk : 0u;
    asm volatile("" : "+v"(base));          // the select stays a select of the BASE (the optimiser would push the step's offset into both arms: an addition per step); the offset is the read's immediate
    return *reinterpret_cast<const lds_f64*>(base + 8u * uint32_t(step));
}
template <int C, int CMAX>
__device__ __forceinline__ void spa_walk(double& temp, uint32_t achk, spa_cptr64 bm, uint64_t m0, uint64_t m1) {
    typedef __attribute__((address_space(3))) double lds_f64;
    if (m0 == 0) return;
    uint64_t n0 = 0, n1 = 0;
    if constexpr (C + 1 < CMAX) { n0 = bm[2 * C + 2]; n1 = bm[2 * C + 3]; }
    const volatile lds_f64* chk = reinterpret_cast<const volatile lds_f64*>(achk);      // volatile: see spa_walk4
    const double a0 = chk[2 * C], a1 = chk[2 * C + 1];
    spa_masked_mul2(temp, a0, a1, m0, m1);
    if constexpr (C + 1 < CMAX) spa_walk<C + 1, CMAX>(temp, achk, bm, n0, n1);
}
// The same in groups of four steps (the high-degree graphs: half as many load round trips per check)
template <int C, int CMAX>
__device__ __forceinline__ void spa_walk4(double& temp, uint32_t achk, spa_cptr64 bm, uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3) {
    typedef __attribute__((address_space(3))) double lds_f64;
    if (m0 == 0) return;
    uint64_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
    if constexpr (C + 1 < CMAX) { n0 = bm[4 * C + 4]; n1 = bm[4 * C + 5]; n2 = bm[4 * C + 6]; n3 = bm[4 * C + 7]; }
    // Steps 4C, 4C+1 take their mask on the vector side (the selected read address: spa_walk_factor), steps 4C+2, 4C+3 on the scalar side (the
    // execution mask: spa_masked_mul2) - the walk's cost is split between the two units this kernel loads most. Measured on mode 16
    // (baseband_test, 4096 x 50, ms): all four under the execution mask 8.09 / 9.94 / 10.50 at -15 / 8 / 13 dB (scalar unit 84 % busy in noise),
    // all four through selected addresses 7.68 / 9.78 / 10.39 (vector unit 91 % busy at 13 dB), two and two 7.70 / 9.65 / 10.24.
    // volatile keeps the two plain reads two ds_read_b64: left alone the compiler pairs them into ds_read2_b64, and the LDS pipeline takes 8 cycles
    // for one of those against 2.2 for a ds_read_b64 - tools/ubench/lds_mask.hip, profiles/r05_lds_mask.json (the selected reads have an address
    // register each and cannot be paired).
    const volatile lds_f64* chk = reinterpret_cast<const volatile lds_f64*>(achk);
    const double a0 = spa_walk_factor(achk, m0, 4 * C), a1 = spa_walk_factor(achk, m1, 4 * C + 1), a2 = chk[4 * C + 2], a3 = chk[4 * C + 3];
    temp *= a0; SPA_KEEP(temp);
    temp *= a1; SPA_KEEP(temp);
    if (m2 != 0) spa_masked_mul2(temp, a2, a3, m2, m3);
    if constexpr (C + 1 < CMAX) spa_walk4<C + 1, CMAX>(temp, achk, bm, n0, n1, n2, n3);
}

template <int NE, int DMX>
__device__ __forceinline__ void spa_decode(const LdpcDev& T, const float* __restrict__ llr_in, int F,
                                            uint8_t* __restrict__ bits_out, int* __restrict__ iters_out,
                                            uint8_t* __restrict__ payload_out, MgpuStatsDev* __restrict__ stats_out,
                                            const float* __restrict__ variance_in, const float* __restrict__ snr_variance_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = T.S;
    constexpr int N = kN;
    constexpr int kMoff = kSpaOnesBytes + N * 8;        // LDS byte offset of the message array
    double* Lt = reinterpret_cast<double*>(smem + kSpaOnesBytes);       // LLRtmp per variable (behind the 48 doubles 1.0 at address 0: spa_walk)
    double* M = Lt + N;                                 // R or T per padded edge slot
    float* Li = reinterpret_cast<float*>(M + S);        // channel LLR
    uint8_t* hard = reinterpret_cast<uint8_t*>(Li + N);
    uint8_t* bytes = hard + ((N + 15) & ~15);
    int* flag = reinterpret_cast<int*>(bytes + 256);

    const int tid = threadIdx.x, f = blockIdx.x;
    if (f >= F) return;
    if (uint32_t(uintptr_t(smem)) != 0) __builtin_trap();
    if (tid < int(kSpaOnesBytes / 8)) reinterpret_cast<double*>(smem)[tid] = SPA_ALLLANES != 0 && NE != 8 ? 0x1p-10 : 1.0;      // (rate 14/16: the walk's own 1.0s - its padding lanes then read 1.0, see kAllLanes)       // the walk's neutral factors (published by the barrier in front of the first pass); kAllLanes: what a padding lane reads as its "posterior"
    // Rate 14/16 (the only code with checks of degree 46; the only one the zero-forcing modes 15 / 16 use): its wavefronts meet the regimes in
    // which a whole wavefront gets one of tanh's / atanh's immediate answers (spa_math.h: spa_tanh_half_wave, spa_atanh_x2_wave) - round 5:
    // 10.97 -> 8.50 ms per 4096 x 50 on mode 16's hard (+-Inf) LLRs, 10.04 -> 9.22 on mode 14 in noise. The other kernels keep the plain calls:
    // with the shortcuts' extra paths the register allocator spills the resident variable record (+6 % on the headline, measured).
    // Here the lane's first variable record is fetched per iteration like the second one (below), which frees the registers the
    // shortcuts need.
    constexpr bool kWaveShortcuts = NE == 8;
    constexpr bool kVaResident = !kWaveShortcuts;
    // The adaptive look policy (below) is built into the kernels of rates 1/16 .. 8/16 only. Rate 14/16 keeps the fixed schedule of rounds 2-5
    // (judged looks in a frame's first kSpecStart iterations): its 200 checks fill 116 bins, so the 16 sampled bins hold some 20 checks - too
    // few to steer anything - and modes 12 / 14 leave after one or two iterations at their operating points (nothing to skip); measured with the
    // policy switched on: +1 ... +2 % on mode 12 at 7.5 / 8.5 dB (profiles/r06_ab_spec_sweep.txt).
    constexpr bool kAdaptive = !kWaveShortcuts;
    // Rates 1/16 .. 8/16: the 1-2 % of a bin's lanes that are padding are NOT masked out of the tanh / atanh / the two message stores (three
    // execution-mask regions = nine scalar instructions per bin). They compute on harmless small values instead - their table entries make
    // them read LDS address 0 as "posterior", where these kernels keep 2^-10, they enter the product walk with 2^-10 and take no factor (the
    // walk's masks exclude them), so they sit in the routines' cheapest branches (tanh: k = 0; atanh: |x| < 0.5, direct log1p) and drag
    // their wavefront into no other - and write to their own padding slots of the message array, which no check and no variable reads. The syndrome's ballot is masked as before. (Rate 14/16 keeps the masks: its wavefront-wide shortcuts
    // must not see padding lanes.)
    constexpr bool kAllLanes = SPA_ALLLANES != 0 && (!kWaveShortcuts || SPA_ALLLANES8 != 0);
    constexpr int kSpecStart = SPA_SPEC_START;      // from this iteration on every look at the posteriors is taken inside the next check pass
    SPA_STAMP_DECL(F);
    SPA_STAMP(1);                                   // 1: start
    // Before the first iteration every R is zero, so Q = posterior - R is the channel LLR on every edge of a variable and T = tanh(Q/2) is
    // one value per VARIABLE (1600) where the check pass would compute it per EDGE (3574 - 6604): it is computed here, once per variable,
    // and the first check pass takes it from the posterior array instead of evaluating tanh (x - 0.0 == x exactly, so the value is the
    // one the pass would have computed; tanh keeps the sign, so the syndrome of the channel LLRs reads the same hard decisions). The
    // messages are never zeroed: nothing reads them before the first pass has written them. Round 4: -6 % at a mode's operating point.
    // A HARD frame - every |LLR| >= 200, no NaN - is decided before the first iteration. tanh(Q/2) is +-1 exactly from |Q| = 44 on
    // (s_tanh.c), every product of +-1 is clamped to +-0.9999999 (ldpc_decoder_SPA.cc:150-156) and R = 2 atanh(..) = +-16.81; a variable
    // has at most nine edges, so |Q| = |LLR + sum of the other R| >= 200 - 9 x 16.82 = 48.6 in every iteration: T = sign(LLR) for ever,
    // the posteriors keep their signs and the syndrome of the first look is the syndrome of every later one. Such a frame either leaves at
    // iteration 0 or runs all its iterations without changing a bit: the iterations are skipped, bits and iteration count are those the loop
    // would deliver. This is what the zero-forcing modes (15, 16) hand the decoder behind RX_SHM at ANY signal-to-noise ratio: their pilots
    // equalise onto themselves, the measured variance is ~1e-33 and the LLRs are +-1e30 .. +-Inf (telecom_system.cc:1289-1295,
    // ofdm.cc:1500-1521) - a frame of theirs with one wrong bit cost the reference's 50 iterations and changed nothing.
    const float* lin = llr_in + size_t(f) * N;
    int hard_lane = 1;
    for (int v = tid; v < N; v += LDPC_THREADS) {
        const float l = lin[v];
        Li[v] = l;
        Lt[v] = spa_tanh_half(double(l));
        hard_lane &= int(__builtin_fabsf(l) >= kSpaHardLlr);     // false for a NaN (device_tables.h ties the threshold to the degree limit and the clamp)
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    struct VarRec { uint32_t vi, w0, w1, w2, w3, w4; };
    const __amdgpu_buffer_rsrc_t vrec = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(T.vinfo2), 0, N * 32, 0x00020000);
    auto load_var = [&](int i) -> VarRec {          // rows past N read as zeros (the buffer's bounds check): degree 0
        const spa_u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(vrec, i * 32, 0, 0);
        const spa_u32x2 hi = __builtin_amdgcn_raw_buffer_load_b64(vrec, i * 32, 16, 0);
        return VarRec{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
    };
    // LDS accessed by integer byte offset: the kernel has no static LDS, so the dynamic block starts at address 0 (checked
    // below), and an address formed from an integer spares the "+ base" addition per access the symbol would cost
    typedef __attribute__((address_space(3))) double lds_f64;
    auto ldsd = [](uint32_t byte_off) -> lds_f64* { return reinterpret_cast<lds_f64*>(byte_off); };
    auto Mb = [&](uint32_t byte_off) -> double { return *ldsd(kMoff + byte_off); };
    auto var_update = [&](const VarRec& q) {
        const uint32_t v = q.vi & 0x7ff;
        uint32_t deg = q.vi >> 11;
        asm volatile("" : "+v"(deg));
        double s = Li[v];
        if (deg > 0) { s += Mb(q.w0 & 0xffff); SPA_KEEP(s); }
        if (deg > 1) { s += Mb(q.w0 >> 16); SPA_KEEP(s); }
        if (deg > 2) {
            s += Mb(q.w1 & 0xffff); SPA_KEEP(s);
            if (deg > 3) { s += Mb(q.w1 >> 16); SPA_KEEP(s); }
            if (deg > 4) { s += Mb(q.w2 & 0xffff); SPA_KEEP(s); }
            if (deg > 5) {
                s += Mb(q.w2 >> 16); SPA_KEEP(s);
                if (deg > 6) { s += Mb(q.w3 & 0xffff); SPA_KEEP(s); }
                if (deg > 7) { s += Mb(q.w3 >> 16); SPA_KEEP(s); }
                if (deg > 8) { s += Mb(q.w4 & 0xffff); SPA_KEEP(s); }
            }
        }
        Lt[v] = s;
    };
    auto var_update4 = [&](uint32_t vi, uint32_t w0, uint32_t w1) {      // rows from 1024 on: at most four edges (checked on the host)
        const uint32_t v = vi & 0x7ff;
        uint32_t deg = vi >> 11;
        asm volatile("" : "+v"(deg));
        double s = Li[v];
        if (deg > 0) { s += Mb(w0 & 0xffff); SPA_KEEP(s); }
        if (deg > 1) { s += Mb(w0 >> 16); SPA_KEEP(s); }
        if (deg > 2) {
            s += Mb(w1 & 0xffff); SPA_KEEP(s);
            if (deg > 3) { s += Mb(w1 >> 16); SPA_KEEP(s); }
        }
        Lt[v] = s;
    };
    if (tid == 0) { flag[0] = 0; flag[1] = 0; flag[2] = LDPC_THREADS / 64; flag[3] = LDPC_THREADS / 64; }
    // (no __syncthreads_and: it brings static LDS, and this kernel's byte-offset addressing needs the dynamic block at address 0) one word per
    // wavefront in the epilogue's scratch bytes, nothing to initialise
    int* hard_w = reinterpret_cast<int*>(bytes);
    const int hard_wave = __builtin_amdgcn_ballot_w64(hard_lane == 0) == 0;       // (the whole wavefront votes: not under the lane-0 branch)
    if ((tid & 63) == 0) hard_w[tid >> 6] = hard_wave;
    __syncthreads();
    const bool hard_frame = __builtin_amdgcn_ballot_w64((tid & 63) < LDPC_THREADS / 64 && hard_w[tid & 15] == 0) == 0;

    // m: the sign bits of a bin's posteriors (lanes in use only). The inclusive prefix parity, taken at the lanes that end a check, is the parity of
    // all checks up to that one: some check of the bin is odd <=> some prefix is (bin_unsat). HOW MANY are odd (bin_odd_count) takes each
    // prefix against its predecessor: the predecessor's bit, moved one lane up, stands on the check's first lane; added to the bin's non-end lanes
    // (all ones below the check's end lane) it ripples up to exactly that end lane and no further.
    auto bin_prefix = [&](unsigned long long m, unsigned long long ends) -> unsigned long long {
        m ^= m << 1; m ^= m << 2; m ^= m << 4; m ^= m << 8; m ^= m << 16; m ^= m << 32;
        return m & ends;
    };
    auto bin_unsat = [&](unsigned long long m, unsigned long long ends) -> bool { return bin_prefix(m, ends) != 0; };
    auto bin_odd_count = [&](unsigned long long pre, unsigned long long used, unsigned long long ends) -> int {
        return __builtin_popcountll(pre ^ (((pre << 1) + (used & ~ends)) & ends));
    };
    // What a JUDGED look at the posteriors (syn_judge) leaves in flag[p & 1]: 0 = every check satisfied; otherwise the number of wavefronts that
    // saw an odd check (low byte) + 256 x the number of odd checks in bins 0..15 (every wavefront's first round: a SAMPLE of the syndrome's
    // weight, 16 of the code's 56-116 bins). The sample steers only where the next looks are taken (below: "adaptive"), never what they see.
    // A look taken inside a check pass leaves 0 or 1.
    auto flag_report = [&](int p, bool unsat, int cnt) {
        if (unsat && (tid & 63) == 0) __hip_atomic_fetch_add(&flag[p & 1], 1 + (cnt << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // Per-slot addresses come straight from a table (no field extraction): the LDS offset of the slot's posterior and the
    // LDS address of its check's first message, one buffer load per round (lane offset in a register, round offset in a
    // scalar). What is uniform over a bin comes through scalar loads: the lanes in use and the lanes ending a check (bhead),
    // the walk masks (bmask); "valid" is the bin's lane mask applied as exec.
    const __amdgpu_buffer_rsrc_t sadr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(T.sadr), 0, 0x7fffffff, 0x00020000);
    constexpr int kRound = LDPC_THREADS * 8;                       // bytes of address table per round
    const spa_cptr64 bhead0 = (spa_cptr64)(T.bhead) + size_t(wave) * 4;

    // Syndrome only (before the first iteration and, in a frame's first eight iterations, after every variable update): pure latency — one L2
    // round trip for the rounds' addresses and lane masks, one LDS round trip for the posteriors, a dependent run of scalar instructions —
    // and at a mode's operating point there is one of these per iteration. It comes in two halves: syn_fetch requests all rounds' addresses
    // and masks (issued in FRONT of the work that precedes the pass, the variable update, so the L2 round trip is over when the barrier
    // opens: round 4), syn_judge reads the posteriors and forms the parities.
    struct SynPre { uint32_t alt[NE]; unsigned long long vmask[NE], ends[NE]; };
    auto syn_fetch = [&](SynPre& q) {
        uint32_t off = tid * 8;
        asm volatile("" : "+v"(off));                  // a fetch per pass: hoisted out of the iteration loop these would be 6 + 24 registers to spill
#pragma unroll
        for (int r = 0; r < NE; ++r) q.alt[r] = __builtin_amdgcn_raw_buffer_load_b32(sadr, off, r * kRound, 0);
#pragma unroll
        for (int r = 0; r < NE; ++r) { q.vmask[r] = bhead0[r * 64]; q.ends[r] = bhead0[r * 64 + 1]; }
    };
    auto syn_pin = [&](SynPre& q) {                    // keeps the requests above where they were written (ahead of a barrier)
#pragma unroll
        for (int r = 0; r < NE; ++r) asm volatile("" : "+v"(q.alt[r]));
    };
    auto syn_judge = [&](const SynPre& q, int p) {
        double lt[NE];
#pragma unroll
        for (int r = 0; r < NE; ++r) lt[r] = *ldsd(q.alt[r]);              // padding reads variable 0; masked out below
#pragma unroll
        for (int r = 0; r < NE; ++r) SPA_KEEP(lt[r]);                      // all of them here: the optimiser would sink each load to its (conditional) use
        const unsigned long long pre = bin_prefix(__ballot(lt[0] < 0) & q.vmask[0], q.ends[0]);
        const int cnt = kAdaptive ? bin_odd_count(pre, q.vmask[0], q.ends[0]) : 0;
        bool unsat = pre != 0;
#pragma unroll
        for (int r = 1; r < NE; ++r)
            if (!unsat) unsat = bin_unsat(__ballot(lt[r] < 0) & q.vmask[r], q.ends[r]);
        flag_report(p, unsat, cnt);
    };
    auto syndrome_pass = [&](int p) { SynPre q; syn_fetch(q); syn_judge(q, p); };
    // The check pass. Bins are handed out on demand: wavefront w starts with bin w, every further bin goes to whoever asks first (a counter
    // in LDS, asked for while the current bin is being worked on). The hardware issues the oldest wavefront of a SIMD first, so with a
    // fixed split (bin w + 16 r in round r: rounds 1-2) the first wavefronts of a workgroup finish long before the last ones and wait at the
    // barrier: -0.4 % at rate 6/16 (89 bins), -3.6 % at rate 14/16 (116 bins: 7.25 per wavefront). Which wavefront works a bin does not
    // change a bit of its arithmetic.
    auto cn_pass = [&](bool with_syndrome, int p, auto first_tag) {
        constexpr bool first = decltype(first_tag)::value;      // the first pass is a copy of the loop of its own: the steady-state loop carries no test for it
                                                                // (measured: the test as a run-time flag costs the headline 1.6 %; a copy per syndrome mode as well costs 0.7 %)
        bool unsat = false;
        const int nbins = T.S >> 6;
        const uint32_t lane8 = (tid & 63) * 8;
        int b = wave;
        spa_u32x2 ad = __builtin_amdgcn_raw_buffer_load_b64(sadr, lane8, b * 512, 0);
        int* ctr = &flag[2 + (p & 1)];
        const spa_cptr64 bh0 = (spa_cptr64)(T.bhead);
        unsigned long long vm, en;
        if constexpr (SPA_OFF32 != 0) { const spa_cptr64 h = spa_at(T.bhead, uint32_t(b) * 32u); vm = h[0]; en = h[1]; }
        else { vm = bh0[size_t(b) * 4]; en = bh0[size_t(b) * 4 + 1]; }
#pragma unroll 1
        while (b < nbins) {
            const uint32_t alt = ad.x, achk = ad.y;
            const uint32_t own = kMoff + lane8 + uint32_t(b) * 512u;
            spa_cptr64 bm;
            if constexpr (SPA_OFF32 != 0) { bm = spa_at(T.bmask, uint32_t(b) * (uint32_t(T.DM) * 8u)); asm volatile("" : "+s"(bm)); }     // (opaque: the bin's base address is formed ONCE - folded into the walk's loads it is re-added in front of every step pair)
            else bm = (spa_cptr64)(T.bmask) + size_t(b) * T.DM;
            const bool valid = __builtin_amdgcn_inverse_ballot_w64(vm);
            const unsigned long long lanes_in_use = vm;                 // (vm is overwritten with the next bin's mask further down)
            double lt;
            if (kAllLanes || valid) lt = *ldsd(alt);                   // (a padding lane's table entries are 0: it reads a 1.0 at LDS address 0)
            int nxt;                                                   // the next bin, asked for behind this bin's first read
            SPA_UNDEF(nxt);                                            // (only lane 0's value is ever looked at)
            if ((tid & 63) == 0) nxt = atomicAdd(ctr, 1);
            // one odd check settles the pass, so the wave's later bins skip the test: the ballot -> prefix-XOR chain is a dependent run of
            // scalar instructions on the bin's critical path (6.27 -> 6.06 ms per 4096 x 50 on the headline, where the first bin settles it)
            if (with_syndrome && !unsat) unsat = bin_unsat(__ballot(lt < 0) & vm, en);
            if (kAllLanes || valid) {
                double t;
                if constexpr (first) t = lt;                           // the posterior array holds T itself (see the top of the kernel)
                else if constexpr (kWaveShortcuts) t = spa_tanh_half_wave(lt - *ldsd(own), kAllLanes ? lanes_in_use : ~0ull);
                else t = spa_tanh_half(lt - *ldsd(own));
                *ldsd(own) = t;
            }
            __builtin_amdgcn_wave_barrier();
            // Product of the check's OTHER T values in slot order, starting from 1.0 (the reference's temp *= ...):
            // every lane of a check reads the check's slots in order (a broadcast) and multiplies under the bin's
            // tabulated mask for that step (own slot, slots past the check's degree and padding lanes excluded).
            // Uniform control flow: the masks come through scalar loads, padding lanes read slot 0 and never multiply.
            double temp = 1;
            if constexpr (kAllLanes) temp = SPA_MAKE(valid ? 0x3ff00000u : 0x3f500000u, 0u);          // padding lanes: 2^-10 (they take no factor), see kAllLanes
            if constexpr (DMX > 16) spa_walk4<0, DMX / 4>(temp, achk, bm, bm[0], bm[1], bm[2], bm[3]);
            else spa_walk<0, DMX / 2>(temp, achk, bm, bm[0], bm[1]);
            // the next bin's addresses and lane masks land in the registers this one is done with, behind the atanh (the tables have a spare round)
            nxt = __builtin_amdgcn_readfirstlane(nxt);
            const int nb = nxt < nbins ? nxt : nbins;
            ad = __builtin_amdgcn_raw_buffer_load_b64(sadr, lane8, nb * 512, 0);
            if constexpr (SPA_OFF32 != 0) { const spa_cptr64 h = spa_at(T.bhead, uint32_t(nb) * 32u); vm = h[0]; en = h[1]; }
            else { vm = bh0[size_t(nb) * 4]; en = bh0[size_t(nb) * 4 + 1]; }
            double rr;
            if (kAllLanes || valid) {
                if constexpr (kWaveShortcuts) rr = spa_atanh_x2_wave(temp, kAllLanes ? lanes_in_use : ~0ull);
                else rr = spa_atanh_x2(temp);
            }
            __builtin_amdgcn_wave_barrier();
            if (kAllLanes || valid) *ldsd(own) = rr;
            b = nxt;
        }
        if (with_syndrome && unsat && (tid & 63) == 0) flag[p & 1] = 1;        // (no sample from here: counted inside the bin loop it cost every launch 0.7-2 %, NOTES R6.8)
    };
#if SPA_VR_RESIDENT
    // the record of the lane's first variable stays in registers (6); the second one (3 words, rows from 1024 on) is requested every iteration
    // in front of the check pass's barrier, where its L2 round trip costs nothing (round 5, same-box: 5.585 vs 5.587 ms on the headline) - the
    // four registers are the allocator's slack: at 63 of 64 every small change of the loop spilled something (+6 % when it was this record)
    VarRec va;
    if constexpr (kVaResident) va = load_var(tid);
    spa_u32x4 vb;
#endif
    int iteration = 0;
    SPA_STAMP(2);                                   // 2: inputs in LDS, tables requested
    syndrome_pass(0);
    SPA_STAMP(3);                                   // 3: syndrome pass done (before its barrier)
    __syncthreads();
    SPA_STAMP(4);                                   // 4: behind the barrier
    int look = flag[0];                             // what the last look at the posteriors reported (flag_report)
    if (look && hard_frame) {
        iteration = T.max_iters + 1;
        if (tid == 0 && T.hard_frames) atomicAdd(T.hard_frames + (f & 63), 1ull);      // so that a benchmark can tell reported from executed iterations (64 counters: a launch of hard frames does not queue on one address)
    }
    else if (look) {
        // Where the look at an iteration's posteriors is taken - "judged": a syndrome-only pass behind the variable update (one more barrier, but a
        // frame that has converged leaves at once); "speculative": inside the next iteration's check pass (no pass of its own, but a frame that has
        // converged has paid a check pass in vain). Rounds 2-5: judged in a frame's first kSpecStart iterations, speculative afterwards. Round 6:
        // ADAPTIVE in the first kSpecStart iterations - a judged look also samples the syndrome's weight (syn_judge), and with many odd checks
        // left the next iteration - or the next two - cannot be the last one in practice (mode 8 at its operating point: 266 -> 64 -> 23 -> 2 -> 0
        // odd checks of 1000; no frame of any mode went from more than 41 to 0 in one iteration, from more than 66 at the first one:
        // tests/tools/unsat_profile.py), so those iterations' posteriors are looked at speculatively and the next judged look comes after them.
        // Both forms test the same syndrome of the same posteriors: bits and iteration counts do not depend on the choice
        // (tests/test_gpu_parity.py runs seven threshold sets against the oracle at nine iteration caps). Mode 8: -2.8 % at 3.5 dB, -4 % at 6 dB,
        // -2.3 % at 1.5 dB, the 50-iteration launches unchanged (profiles/r06_ab_spec.txt). What did NOT work, and shaped this form (NOTES R6.8):
        // a sample taken by the speculative looks as well (inside the bin loop: +0.7 ... +2 % on every launch - the loop has no scalar
        // register to spare), and separate kernel arguments for the thresholds (each one is a scalar register from the kernel's first
        // instruction on: +6 literal moves per bin).
        // thresholds: bytes of T.spec_sample_pack (255 = never)
        auto run_after = [&](int lk, int shift) -> int {
            const int n = lk >> 8, one = int(T.spec_sample_pack >> shift) & 255, two = int(T.spec_sample_pack >> (shift + 8)) & 255;
            return !kAdaptive ? 0 : (n >= two && two != 255) ? 2 : (n >= one && one != 255) ? 1 : 0;
        };
        // The loop's whole control state is ONE scalar register across the check pass: the iteration (low 16 bits), "this iteration's check pass
        // carries the look at the previous iteration's posteriors" (kCtlSpec) and how many of the coming iterations go without a judged look
        // (from kCtlRun up). Kept as separate variables they are three more scalar registers across the bin loop, and the allocator - at its
        // limit there - answered by rematerialising three of fdlibm's double constants in every bin (+6 scalar moves, NOTES R6.8).
        constexpr int kCtlSpec = 1 << 16, kCtlRun = 1 << 20;
        int ctl = 1 + kCtlRun * run_after(look, 0);
        for (;; ++ctl) {
            asm volatile("" : "+s"(ctl));
            const int it = ctl & 0xffff;
            const bool spec = (ctl & kCtlSpec) != 0;
            const bool skip = it >= kSpecStart || ctl >= kCtlRun;            // this iteration's posteriors get no judged look
            if (it == 1) cn_pass(false, 0, std::true_type());
            else if (it <= T.max_iters) cn_pass((ctl & kCtlSpec) != 0, it - 1, std::false_type());
            else syndrome_pass(it - 1);
            asm volatile("" : "+s"(ctl));
#if !SPA_VR_RESIDENT
            const VarRec va = load_var(tid), vb = load_var(tid + LDPC_THREADS);
#else
            if constexpr (!kVaResident) va = load_var(tid);
            vb = __builtin_amdgcn_raw_buffer_load_b128(vrec, (tid + LDPC_THREADS) * 32, 0, 0);
#endif
            SPA_STAMP(5);                           // 5: check pass done
            __syncthreads();
            SPA_STAMP(6);                           // 6: behind its barrier
            if (spec) {
                look = flag[(it - 1) & 1];
                if (!look) { iteration = it - 1; break; }
                if (it > T.max_iters) { iteration = T.max_iters + 1; break; }
            }
            if (tid == 0) { flag[it & 1] = 0; flag[2 + (it & 1)] = LDPC_THREADS / 64; }
            SynPre pre;
            if (!skip) syn_fetch(pre);
#if SPA_VR_RESIDENT
            // the records' fields are unpacked here, every iteration: unpacked once in front of the loop (what the optimiser prefers) they are
            // thirteen more values to keep across the check pass, i.e. spills
            VarRec qa = va;
            spa_u32x4 qb = vb;
            asm volatile("" : "+v"(qa.vi), "+v"(qa.w0), "+v"(qa.w1), "+v"(qa.w2), "+v"(qa.w3), "+v"(qa.w4), "+v"(qb));
            var_update(qa);
            if (tid + LDPC_THREADS < N) var_update4(qb.x, qb.y, qb.z);
#else
            var_update(va);
            if (tid + LDPC_THREADS < N) var_update(vb);
#endif
            if (!skip) syn_pin(pre);
            SPA_STAMP(7);                           // 7: variable update done
            __syncthreads();
            SPA_STAMP(8);                           // 8: behind its barrier
            ctl = skip ? (ctl | kCtlSpec) - (ctl >= kCtlRun ? kCtlRun : 0) : ctl & ~kCtlSpec;
            if (!skip) {
                syn_judge(pre, it);
                SPA_STAMP(3);
                __syncthreads();
                SPA_STAMP(4);
                look = flag[it & 1];
                if (!look) { iteration = it; break; }
                if (it == T.max_iters) { iteration = T.max_iters + 1; break; }
                ctl += kCtlRun * run_after(look, 16);
            }
        }
    }
    SPA_STAMP(9);                                   // 9: loop left
    for (int v = tid; v < N; v += LDPC_THREADS) hard[v] = Lt[v] < 0;
    __syncthreads();
    decode_tail(T, f, hard, bytes, iteration, bits_out, iters_out, payload_out, stats_out, variance_in, snr_variance_in);
    SPA_STAMP(10);                                  // 10: tail done
}

#define SPA_KERNEL(NE, DMX)                                                                                    \
    extern "C" __global__ __launch_bounds__(LDPC_THREADS, 8) void mgpu_ldpc_spa_kernel_ne##NE(               \
        LdpcDev T, const float* __restrict__ llr_in, int F, uint8_t* __restrict__ bits_out,                     \
        int* __restrict__ iters_out, uint8_t* __restrict__ payload_out, MgpuStatsDev* __restrict__ stats_out,   \
        const float* __restrict__ variance_in, const float* __restrict__ snr_variance_in) {                     \
        spa_decode<NE, DMX>(T, llr_in, F, bits_out, iters_out, payload_out, stats_out, variance_in, snr_variance_in); \
    }
SPA_KERNEL(4, 16)
SPA_KERNEL(5, 16)
SPA_KERNEL(6, 16)
SPA_KERNEL(7, 16)
SPA_KERNEL(8, 48)
extern "C" int mgpu_spa_max_degree(int ne) { return ne == 8 ? 48 : 16; }

// ---------------------------------------------------------------------------------------------
// Sum-product, single precision ("spa_fast"; BASELINE.json north_star's fast variant, SURVEY.md §7.3-1: "SPA-equivalent
// check node in FP32"). NOT the reference's arithmetic: the same flooding schedule, the same tanh rule
// R = 2*atanh(prod_others tanh(Q/2)) and the same iteration / early-exit convention as ldpc_decoder_SPA.cc:25-218, evaluated
// in fp32 with the hardware's exp2 / log2 / rcp:
//   * T = tanh(Q/2) = (1 - e)/(1 + e), e = exp(-|Q|): one v_exp_f32 + one v_rcp_f32; |T| is kept inside [2^-30, 1 - 2^-24] so that
//     a product never contains an exact zero or one;
//   * R = 2*atanh(p) = ln2 * log2((1 + p)/(1 - p)): one v_rcp_f32 + one v_log_f32; 1 - |p| is exact (Sterbenz) and >= 2^-24, which
//     caps |R| at 17.3 (the fp64 decoder's own clamp of +-1 to +-0.9999999 caps it at 16.8).
// Round 3: the GROUPED layout (LdpcGraph::gdesc / gkind / vinfo_g) replaced the bin-packed one this decoder shared with the fp64
// kernel (rate 14/16: 6.36 -> 3.0 ms per 4096 x 50; rate 6/16: 2.14 -> 1.98). A check occupies an aligned group of
// 2^k lanes (k = 1..6), a bin holds groups of one size. The product over a check's edges is then an all-reduce inside the wavefront -
// lane ^ 1, lane ^ 2 (quad permutes), half-row mirror, row mirror (DPP), and for 32 / 64 lanes the four row products through scalar
// registers - instead of a walk through LDS: T never leaves the registers, M holds only R, and nothing between the tanh and the atanh
// waits for memory. The own factor is divided out of the check's total (|T| is kept inside [2^-30, 1 - 2^-24], so never zero); padding
// lanes contribute 1. The order of the multiplications differs from the reference's; this decoder's parity is statistical (same decode
// rate and payloads as the fp64 decoder, tests/test_gpu_parity.py::test_spa_fast_*), unlike the fp64 kernel's.
// The syndrome is the parity of the sign bits of each group: a fold of the ballot by the group size (2 log2(g) scalar instructions).
template <int CTRL>
__device__ __forceinline__ float spag_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
extern "C" size_t mgpu_spa_fast_lds_bytes(int Sg, int N) {
    return size_t(4) * Sg + size_t(4) * N + ((N + 15) & ~15) + 256 + 64;
}

// tanh(q/2) of the fp32 sum-product decoder: (1 - e)/(1 + e), e = exp(-|q|), magnitude kept inside [2^-30, 1 - 2^-24]
__device__ __forceinline__ float spaf_tanh_half(float q) {
    const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * __builtin_fabsf(q));
    float a = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    a = __builtin_amdgcn_fmed3f(a, 0x1.0p-30f, 0x1.fffffep-1f);
    return __builtin_copysignf(a, q);
}

template <int THREADS, int RULE>     // RULE 0: sum-product ("spa_fast"), 1: normalised min-sum
__device__ __forceinline__ void spa_fast_decode(const LdpcDev& T, const float* __restrict__ llr_in, int F,
                                                 uint8_t* __restrict__ bits_out, int* __restrict__ iters_out,
                                                 uint8_t* __restrict__ payload_out, MgpuStatsDev* __restrict__ stats_out,
                                                 const float* __restrict__ variance_in, const float* __restrict__ snr_variance_in) {
    static_assert(THREADS == 512, "the grouped tables are laid out in rounds of 8 bins");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int S = T.Sg, N = T.N;
    const int NE = (S + THREADS - 1) / THREADS;       // rounds: wave w works on bin w + 8 r in round r
    float* M = reinterpret_cast<float*>(smem);        // R per slot
    float* Lt = M + S;                                // posterior per variable
    uint8_t* hard = reinterpret_cast<uint8_t*>(Lt + N);
    uint8_t* bytes = hard + ((N + 15) & ~15);
    int* flag = reinterpret_cast<int*>(bytes + 256);
    const int tid = threadIdx.x, f = blockIdx.x, lane = tid & 63;
    if (f >= F) return;
    SPA_STAMP_DECL(F);
    SPA_STAMP(1);
    constexpr int kRows = (kN + THREADS - 1) / THREADS;      // channel LLRs in registers: row i of the variable records belongs to one lane
    // A posterior lives at its variable's ROW (LdpcGraph::vinfo_g's order), not at the variable's number: the variable update's stores are
    // then consecutive, and the host orders the rows so that the 32 posteriors a half-wavefront of the check pass gathers lie in 32
    // different LDS banks (tables.cpp, "bank-aware placement"). The frame's LLRs are fetched by variable number, once.
    uint32_t vrow[kRows];
    float li[kRows];
#pragma unroll
    for (int k = 0; k < kRows; ++k) {
        const int i = tid + k * THREADS;
        vrow[k] = i < N ? T.vinfo_g[size_t(i) * 8] & 0x7ff : 0u;
        li[k] = i < N ? llr_in[size_t(f) * N + vrow[k]] : 0.0f;
        // sum-product: before the first iteration Q is the channel LLR on every edge of a variable, so T = tanh(Q/2) is computed here once per
        // variable and the first check pass takes it from the posterior array (same expression, same value; the sign - all the syndrome and
        // the hard decisions read - is the LLR's)
        // (an LLR of -0.0 decides bit 0 in the reference and in the fp64 kernel, `LLR < 0` being false; the clamp inside spaf_tanh_half would
        // hand copysign its sign and decide 1: zero enters as +0.0)
        if (i < N) Lt[i] = RULE == 0 ? spaf_tanh_half(li[k] == 0.0f ? 0.0f : li[k]) : li[k];
    }
    const uint32_t* __restrict__ gdesc = T.gdesc;
    // the group sizes of the wavefront's bins, 3 bits per round, in one scalar register pair (a scalar load per round put its latency
    // at the head of every bin)
    typedef const uint64_t __attribute__((address_space(4))) * cptr64;
    const uint64_t kpack = ((cptr64)(T.gkpack))[__builtin_amdgcn_readfirstlane(tid >> 6)];
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();

    // any group of 2^kind lanes with an odd number of set bits? fold the ballot onto each group's lowest lane
    auto groups_unsat = [&](unsigned long long m, uint32_t kind) -> bool {
        m ^= m >> 1;
        unsigned long long lead = 0x5555555555555555ull;
        if (kind >= 2) { m ^= m >> 2; lead = 0x1111111111111111ull; }
        if (kind >= 3) { m ^= m >> 4; lead = 0x0101010101010101ull; }
        if (kind >= 4) { m ^= m >> 8; lead = 0x0001000100010001ull; }
        if (kind >= 5) { m ^= m >> 16; lead = 0x0000000100000001ull; }
        if (kind >= 6) { m ^= m >> 32; lead = 1ull; }
        return (m & lead) != 0;
    };
    auto syndrome_pass = [&](int p) {
        bool unsat = false;
        uint32_t k = gdesc[tid];
#pragma unroll 1
        for (int r = 0; r < NE; ++r) {
            const uint32_t kn = gdesc[(r + 1) * THREADS + tid];
            const uint32_t kind = uint32_t(kpack >> (3 * r)) & 7u;
            const float lt = Lt[k & 0x7ff];
            if (!unsat && kind) unsat = groups_unsat(__ballot(lt < 0 && int32_t(k) < 0), kind);
            k = kn;
        }
        if (unsat && lane == 0) flag[p & 1] = 1;
    };
    // Every pass tests the syndrome of the posteriors it starts from (the first one: of the channel LLRs), so a frame costs one check
    // pass more than it has iterations and no syndrome-only passes; the messages start as zeros that are never stored (first).
    auto cn_pass = [&](int p, auto first_tag) {
        constexpr bool first = decltype(first_tag)::value;       // the first pass is a loop of its own (no message reads; sum-product: no tanh)
        constexpr bool with_syndrome = true;
        bool unsat = false;
        uint32_t k = gdesc[tid];
        uint32_t slot = tid;
#pragma unroll 1
        for (int r = 0; r < NE; ++r, slot += THREADS) {
            const uint32_t kn = gdesc[(r + 1) * THREADS + tid];
            const uint32_t kind = uint32_t(kpack >> (3 * r)) & 7u;
            if (kind == 0) { k = kn; continue; }                     // an empty bin (only in the last round)
            const bool valid = int32_t(k) < 0;
            const float lt = Lt[k & 0x7ff];                          // padding lanes read variable 0
            if (with_syndrome && !unsat) unsat = groups_unsat(__ballot(lt < 0 && valid), kind);
            float q;
            if constexpr (first) q = lt; else q = lt - M[slot];
            if constexpr (RULE == 0) {
            float t;
            if constexpr (first) t = valid ? lt : 1.0f;                // the posterior array holds T itself (see the top of the kernel)
            else t = valid ? spaf_tanh_half(q) : 1.0f;
            // the check's total: all-reduce over its aligned group
            float x = t * spag_dpp<0xB1>(t);                                          // lane ^ 1  (quad_perm [1,0,3,2])
            if (kind >= 2) x *= spag_dpp<0x4E>(x);                                    // lane ^ 2  (quad_perm [2,3,0,1])
            if (kind >= 3) x *= spag_dpp<0x141>(x);                                   // row_half_mirror: the other quad of the 8
            if (kind >= 4) x *= spag_dpp<0x140>(x);                                   // row_mirror: the other half of the 16
            if (kind >= 5) {
                const int xi = __builtin_bit_cast(int, x);                            // the four row products, through scalar registers
                const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 16));
                const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 48));
                const float pa = r0 * r1, pb = r2 * r3;
                x = kind == 6 ? pa * pb : (lane < 32 ? pa : pb);
            }
            const float pe = x * __builtin_amdgcn_rcpf(t);                            // product over the other edges
            const float pa = fminf(__builtin_fabsf(pe), 0x1.fffffep-1f);
            const float l2 = __builtin_amdgcn_logf((1.0f + pa) * __builtin_amdgcn_rcpf(1.0f - pa));
            if (valid) M[slot] = __builtin_copysignf(0.693147180559945309f * l2, pe);
            } else {
            // normalised min-sum: R = alpha * (smallest |Q| among the check's other edges) with the sign parity of the others. The two
            // smallest magnitudes (m1 <= m2) and the sign parity of the whole check by the same all-reduce; the own edge is taken out
            // afterwards: min over the others = (|own| == m1) ? m2 : m1 (a tie leaves m2 == m1).
            // Magnitudes are non-negative floats, so their bit patterns order like unsigned integers: the (m1, m2) merge runs on integer
            // minima / maxima - no NaN-quieting instruction in front of every float minimum, the rows' representatives merge on the scalar unit.
            const uint32_t a = valid ? __float_as_uint(q) & 0x7fffffffu : 0x7f800000u;
            uint32_t m1 = a, m2 = 0x7f800000u;
            uint32_t sg = valid ? __float_as_uint(q) & 0x80000000u : 0u;
            auto umin = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
            auto umax = [](uint32_t x, uint32_t y) { return x > y ? x : y; };
            auto dppu = [](auto ctrl, uint32_t v) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), decltype(ctrl)::value, 0xf, 0xf, false)); };
            auto step = [&](auto ctrl) {
                const uint32_t p1 = dppu(ctrl, m1), p2 = dppu(ctrl, m2), ps = dppu(ctrl, sg);
                const uint32_t hi = umax(m1, p1);
                m1 = umin(m1, p1);
                m2 = umin(hi, umin(m2, p2));
                sg ^= ps;
            };
            step(std::integral_constant<int, 0xB1>());
            if (kind >= 2) step(std::integral_constant<int, 0x4E>());
            if (kind >= 3) step(std::integral_constant<int, 0x141>());
            if (kind >= 4) step(std::integral_constant<int, 0x140>());
            if (kind >= 5) {
                auto rs = [](uint32_t v, int l) { return uint32_t(__builtin_amdgcn_readlane(int(v), l)); };
                const uint32_t a1 = rs(m1, 0), a2 = rs(m2, 0), b1 = rs(m1, 16), b2 = rs(m2, 16), c1 = rs(m1, 32), c2 = rs(m2, 32), d1 = rs(m1, 48), d2 = rs(m2, 48);
                const uint32_t sa = rs(sg, 0) ^ rs(sg, 16), sc = rs(sg, 32) ^ rs(sg, 48);
                const uint32_t ab1 = umin(a1, b1), ab2 = umin(umax(a1, b1), umin(a2, b2)), cd1 = umin(c1, d1), cd2 = umin(umax(c1, d1), umin(c2, d2));
                if (kind == 6) { m1 = umin(ab1, cd1); m2 = umin(umax(ab1, cd1), umin(ab2, cd2)); sg = sa ^ sc; }
                else { m1 = lane < 32 ? ab1 : cd1; m2 = lane < 32 ? ab2 : cd2; sg = lane < 32 ? sa : sc; }
            }
            float mag = __uint_as_float((a == m1) ? m2 : m1);
            // Messages saturate at 2^12 (round 4). The rate-1/16 graph has a check of degree 1, whose min-sum message is alpha * min over NO
            // other edge = +Inf; the Inf walked down the degree-2 checks of the accumulator chain and the next Q = Inf - Inf was a NaN that the
            // float minima then ignored. With the cap the posteriors stay finite ((x + 2^12) - 2^12 keeps x to 2^-12), the rate-1/16 modes decode
            // more frames in fewer iterations (mode 0 at -7.5 dB: 1817 -> 1867 of 2048; every other rate: identical results), and the float and
            // the integer form of the merge agree bit for bit.
            mag = fminf(mag * T.minsum_alpha, 4096.0f);
            if (valid) M[slot] = __uint_as_float(__float_as_uint(mag) | ((sg ^ __float_as_uint(q)) & 0x80000000u));
            }
            k = kn;
        }
        if (with_syndrome && unsat && lane == 0) flag[p & 1] = 1;
    };
    // A lane's variable records (which slots hold the messages of the variables in its four rows) stay in registers: loaded per iteration
    // they put an L2 round trip behind every barrier (1.93 -> 1.60 ms per 4096 x 50 on mode 8). The rows are sorted by degree, and in
    // every Mercury code the rows from 512 on have at most 6, from 1024 on at most 4, from 1536 on at most 2 edges (checked on the
    // host: LdpcGraph, grouped layout), so the four records take 6 + 4 + 3 + 2 registers.
    struct VarRec { uint32_t vi, w0, w1, w2, w3, w4; };
    auto load_var = [&](auto maxs_tag, int i) -> VarRec {
        constexpr int MAXS = decltype(maxs_tag)::value;
        VarRec q{0, 0, 0, 0, 0, 0};
        if (i >= N) return q;
        const uint32_t* row = T.vinfo_g + size_t(i) * 8;
        if constexpr (MAXS <= 2) { const uint2 a = *reinterpret_cast<const uint2*>(row); q.vi = a.x; q.w0 = a.y; }
        else {
            const uint4 a = *reinterpret_cast<const uint4*>(row);
            q.vi = a.x; q.w0 = a.y; q.w1 = a.z;
            if constexpr (MAXS > 4) q.w2 = a.w;
            if constexpr (MAXS > 6) { const uint2 c = *reinterpret_cast<const uint2*>(row + 4); q.w3 = c.x; q.w4 = c.y; }
        }
        return q;
    };
    auto var_update = [&](auto maxs_tag, const VarRec& q, float s, int row) {       // s: the variable's channel LLR
        constexpr int MAXS = decltype(maxs_tag)::value;
        const uint32_t deg = q.vi >> 11;
        const float m0 = M[q.w0 & 0xffff], m1 = M[q.w0 >> 16];
        s = deg > 0 ? s + m0 : s; s = deg > 1 ? s + m1 : s;
        if constexpr (MAXS > 2) {
            if (deg > 2) {      // rows sorted by degree: wavefronts of degree-2 parity bits skip the rest
                const float m2 = M[q.w1 & 0xffff], m3 = M[q.w1 >> 16];
                s += m2;
                s = deg > 3 ? s + m3 : s;
                if constexpr (MAXS > 4) { const float m4 = M[q.w2 & 0xffff]; s = deg > 4 ? s + m4 : s; }
            }
        }
        if constexpr (MAXS > 5) {
            if (deg > 5) {
                const float m5 = M[q.w2 >> 16];
                s += m5;
                if constexpr (MAXS > 6) {
                    const float m6 = M[q.w3 & 0xffff], m7 = M[q.w3 >> 16], m8 = M[q.w4 & 0xffff];
                    s = deg > 6 ? s + m6 : s; s = deg > 7 ? s + m7 : s; s = deg > 8 ? s + m8 : s;
                }
            }
        }
        Lt[row] = s;
    };
    static_assert(kRows == 4, "the variable records are specialised for four rows of 512");
    const std::integral_constant<int, 9> s9; const std::integral_constant<int, 6> s6; const std::integral_constant<int, 4> s4; const std::integral_constant<int, 2> s2;
    const VarRec vr0 = load_var(s9, tid), vr1 = load_var(s6, tid + THREADS), vr2 = load_var(s4, tid + 2 * THREADS), vr3 = load_var(s2, tid + 3 * THREADS);
    int iteration = 0;
    SPA_STAMP(2);
    for (int it = 1;; ++it) {
        if (it == 1) cn_pass(0, std::true_type());
        else if (it <= T.max_iters) cn_pass(it - 1, std::false_type());
        else syndrome_pass(it - 1);
        SPA_STAMP(5);
        __syncthreads();
        SPA_STAMP(6);
        if (!flag[(it - 1) & 1]) { iteration = it - 1; break; }
        if (it > T.max_iters) { iteration = T.max_iters + 1; break; }
        if (tid == 0) flag[it & 1] = 0;
        var_update(s9, vr0, li[0], tid);
        var_update(s6, vr1, li[1], tid + THREADS);
        var_update(s4, vr2, li[2], tid + 2 * THREADS);
        if (tid + 3 * THREADS < N) var_update(s2, vr3, li[3], tid + 3 * THREADS);
        SPA_STAMP(7);
        __syncthreads();
        SPA_STAMP(8);
    }
    SPA_STAMP(9);
#pragma unroll
    for (int k = 0; k < kRows; ++k)
        if (tid + k * THREADS < N) hard[vrow[k]] = Lt[tid + k * THREADS] < 0;       // hard decisions by variable number
    __syncthreads();
    decode_tail(T, f, hard, bytes, iteration, bits_out, iters_out, payload_out, stats_out, variance_in, snr_variance_in);
    SPA_STAMP(10);
}
extern "C" __global__ __launch_bounds__(512, 8) void mgpu_ldpc_spa_fast_kernel_t512(
    LdpcDev T, const float* __restrict__ llr_in, int F, uint8_t* __restrict__ bits_out,
    int* __restrict__ iters_out, uint8_t* __restrict__ payload_out, MgpuStatsDev* __restrict__ stats_out,
    const float* __restrict__ variance_in, const float* __restrict__ snr_variance_in) {
    spa_fast_decode<512, 0>(T, llr_in, F, bits_out, iters_out, payload_out, stats_out, variance_in, snr_variance_in);
}
// Normalised min-sum (BASELINE.json's north_star names it; alpha = mgpu_config.minsum_alpha, default 0.8) on the same skeleton.
extern "C" __global__ __launch_bounds__(512, 8) void mgpu_ldpc_minsum_kernel_t512(
    LdpcDev T, const float* __restrict__ llr_in, int F, uint8_t* __restrict__ bits_out,
    int* __restrict__ iters_out, uint8_t* __restrict__ payload_out, MgpuStatsDev* __restrict__ stats_out,
    const float* __restrict__ variance_in, const float* __restrict__ snr_variance_in) {
    spa_fast_decode<512, 1>(T, llr_in, F, bits_out, iters_out, payload_out, stats_out, variance_in, snr_variance_in);
}

// device probe of spa_math.h for tests: out_t[i] = tanh(in[i]); out_a[i] = atanh(in[i]) for |in[i]| < 1 else 0
extern "C" __global__ void mgpu_spa_math_probe_kernel(const double* __restrict__ in, double* __restrict__ out_t,
                                                      double* __restrict__ out_a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    out_t[i] = spa_tanh(x);
    out_a[i] = (spa_fabs(x) < 1.0) ? spa_atanh(x) : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Gradient bit flipping (ldpc_decoder_GBF.cc:25-117), eta = 0.5 (physical_config.cc:73).
extern "C" size_t mgpu_gbf_lds_bytes(int N) { return size_t(4) * N * 2 + ((N + 15) & ~15) + 256 + 64; }

extern "C" __global__ __launch_bounds__(LDPC_THREADS) void mgpu_ldpc_gbf_kernel(
    LdpcDev T, const float* __restrict__ llr_in, int F, uint8_t* __restrict__ bits_out,
    int* __restrict__ iters_out, uint8_t* __restrict__ payload_out, MgpuStatsDev* __restrict__ stats_out,
    const float* __restrict__ variance_in, const float* __restrict__ snr_variance_in) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = T.N, P = T.P;
    float* Lt = reinterpret_cast<float*>(smem);
    int* delta = reinterpret_cast<int*>(Lt + N);
    uint8_t* hard = reinterpret_cast<uint8_t*>(delta + N);
    uint8_t* bytes = hard + ((N + 15) & ~15);
    int* flag = reinterpret_cast<int*>(bytes + 256);
    const int tid = threadIdx.x, f = blockIdx.x;
    if (f >= F) return;
    const float eta = 0.5f;
    for (int v = tid; v < N; v += LDPC_THREADS) { const float l = llr_in[size_t(f) * N + v]; Lt[v] = l; hard[v] = l < 0; delta[v] = 0; }
    if (tid == 0) flag[0] = 0;
    __syncthreads();
    {
        int bad = 0;
        for (int c = tid; c < P; c += LDPC_THREADS) {
            int x = 0;
            for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) x ^= hard[T.cvar[e]];
            bad |= x;
        }
        if (bad) flag[0] = 1;
    }
    __syncthreads();
    int iteration = 0;
    if (flag[0]) {
        for (iteration = 1; iteration <= T.max_iters; ++iteration) {
            __syncthreads();
            if (tid == 0) flag[0] = 0;
            __syncthreads();
            int bad = 0;
            for (int c = tid; c < P; c += LDPC_THREADS) {
                int x = 0;
                for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) x ^= hard[T.cvar[e]];
                bad |= x;
                for (uint32_t e = T.cptr[c]; e < T.cptr[c + 1]; ++e) atomicAdd(&delta[T.cvar[e]], 2 * x - 1);
            }
            if (bad) flag[0] = 1;
            __syncthreads();
            if (!flag[0]) break;
            for (int v = tid; v < N; v += LDPC_THREADS) {
                const int d = delta[v];
                const int k = (d > 0) * (2 * (Lt[v] < 0) - 1) * d;
                Lt[v] += float(k) * eta;
                delta[v] = 0;
                hard[v] = Lt[v] < 0;
            }
        }
    }
    __syncthreads();
    decode_tail(T, f, hard, bytes, iteration, bits_out, iters_out, payload_out, stats_out, variance_in, snr_variance_in);
}
