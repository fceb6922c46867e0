// 256-point radix-2 DIT FFT of one OFDM symbol per wavefront, bit-identical to the reference's
// _fft_fast / _ifft_fast (ofdm.cc:310-340, :343-377; twiddle table :256-290).
//
// The reference permutes the input by bit reversal and then runs stages size = 2..256 in place. Here the data
// stay in natural order: element idx of the permuted array lives at position p = brev8(idx), so stage st pairs
// positions p and p + 2^(8-st), and bin k ends up at position brev8(k). Every butterfly has the same operands
// and the same twiddle (index (brev8(p0) & (2^(st-1)-1)) * 2^(8-st)) as the reference's, hence the same
// roundings. A lane holds four elements and does two stages in registers between LDS transpositions
// (partners 64, 16, 4 and 1 positions apart in turn): 3 round trips through LDS instead of 8. A 16-byte LDS
// access is served 16 lanes (reads: the groups {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32) or 8
// consecutive lanes (writes) per cycle, without conflict when those lanes touch distinct 16-byte columns
// (address / 16 mod 16). Position p is therefore kept at entry (p & ~15) | ((p & 15) ^ X[(p >> 4) & 7]) with
// X = xor of {4, 1, 14} over the set bits: of all 4x4 bit matrices this one makes every one of the seven access
// patterns below conflict-free (the earlier padding p + p/16 paid 224 LDS cycles per transform where 144 are
// needed), and twiddle j lies at entry j ^ (j >> 4) (144 -> 48 cycles: the late stages read 64 different
// twiddles whose indices are bit-reversed lane numbers). The work buffer is private to the wavefront and the
// LDS operations of one wavefront execute in order, so only wave-level scheduling barriers are needed.
#pragma once
#include <hip/hip_runtime.h>

struct c2 { double re, im; };

__device__ __forceinline__ c2 cmul(c2 a, c2 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// A twiddle from the LDS table, which every caller keeps on a 16-byte boundary: one ds_read_b128 (4 LDS-pipeline cycles). Read as a c2 (alignment
// 8) through a lane's precomputed address it becomes a ds_read2_b64, which takes 8 (tools/ubench/lds_mask.hip).
__device__ __forceinline__ c2 fft256_twiddle(const c2* __restrict__ tw, int slot) {
    typedef double fft_v2d __attribute__((ext_vector_type(2)));
    const fft_v2d w = *reinterpret_cast<const fft_v2d*>(tw + slot);
    return {w.x, w.y};
}

#define FFT256_STRIDE 256   // c2 entries of wave-private LDS work space

// entry of twiddle j (0..127) in the LDS table every caller fills: tw[fft256_tw_slot(j)] = exp(-+ 2 pi i j / 256)
__host__ __device__ __forceinline__ int fft256_tw_slot(int j) { return j ^ (j >> 4); }

__device__ __forceinline__ int brev8(int p) { return int(__brev(unsigned(p)) >> 24); }

// In:  r_k = x[lane + 64 k]  (natural order).
// Out: r_k = X[brev8(4 lane + k)]  (unscaled).
// tw:  128 twiddles in LDS: exp(-2 pi i j / 256) for the forward transform, their conjugates for the inverse.
// kCarriersOnly: stop after the last LDS round (r_k = position 4 lane + k BEFORE stages 7 and 8) - wave_fft256_carriers below finishes it
template <bool kCarriersOnly = false>
__device__ __forceinline__ void wave_fft256(c2& r0, c2& r1, c2& r2, c2& r3, c2* __restrict__ v, const c2* __restrict__ tw, int lane) {
    auto bfly_w = [](c2& lo, c2& hi, const c2& w) {
        const c2 t = cmul(w, hi);
        const c2 u = lo;
        hi = {u.re - t.re, u.im - t.im};
        lo = {u.re + t.re, u.im + t.im};
    };
    auto bfly = [&](c2& lo, c2& hi, int p0, int st) {
        const int j = brev8(p0) & ((1 << (st - 1)) - 1);
        bfly_w(lo, hi, fft256_twiddle(tw, fft256_tw_slot(j << (8 - st))));
    };
    auto padded = [](int p) {       // the entry of position p (see the header of this file)
        const int h = p >> 4;
        return (p & ~15) | ((p & 15) ^ ((h & 1 ? 4 : 0) ^ (h & 2 ? 1 : 0) ^ (h & 4 ? 14 : 0)));
    };
    // stages 1, 2: positions lane + 64k; their twiddle indices are wave-uniform (0, 0, 0 and 64)
    const c2 w0 = fft256_twiddle(tw, 0), w64 = fft256_twiddle(tw, fft256_tw_slot(64));
    bfly_w(r0, r2, w0); bfly_w(r1, r3, w0);
    bfly_w(r0, r1, w0); bfly_w(r2, r3, w64);
    v[padded(lane)] = r0; v[padded(lane + 64)] = r1; v[padded(lane + 128)] = r2; v[padded(lane + 192)] = r3;
    __builtin_amdgcn_wave_barrier();
    // stages 3, 4: positions pb + 16k
    const int pb = (lane >> 4) * 64 + (lane & 15);
    r0 = v[padded(pb)]; r1 = v[padded(pb + 16)]; r2 = v[padded(pb + 32)]; r3 = v[padded(pb + 48)];
    bfly(r0, r2, pb, 3); bfly(r1, r3, pb + 16, 3);
    bfly(r0, r1, pb, 4); bfly(r2, r3, pb + 32, 4);
    v[padded(pb)] = r0; v[padded(pb + 16)] = r1; v[padded(pb + 32)] = r2; v[padded(pb + 48)] = r3;
    __builtin_amdgcn_wave_barrier();
    // stages 5, 6: positions pc + 4k
    const int pc = (lane >> 2) * 16 + (lane & 3);
    r0 = v[padded(pc)]; r1 = v[padded(pc + 4)]; r2 = v[padded(pc + 8)]; r3 = v[padded(pc + 12)];
    bfly(r0, r2, pc, 5); bfly(r1, r3, pc + 4, 5);
    bfly(r0, r1, pc, 6); bfly(r2, r3, pc + 8, 6);
    v[padded(pc)] = r0; v[padded(pc + 4)] = r1; v[padded(pc + 8)] = r2; v[padded(pc + 12)] = r3;
    __builtin_amdgcn_wave_barrier();
    // stages 7, 8: positions 4*lane + k
    const int pd = 4 * lane;
    r0 = v[padded(pd)]; r1 = v[padded(pd + 1)]; r2 = v[padded(pd + 2)]; r3 = v[padded(pd + 3)];
    if constexpr (!kCarriersOnly) {
        bfly(r0, r2, pd, 7); bfly(r1, r3, pd + 1, 7);
        bfly(r0, r1, pd, 8); bfly(r2, r3, pd + 2, 8);
    }
    __builtin_amdgcn_wave_barrier();
}

// The receive side keeps 50 of the 256 bins (zero_depadder, ofdm.cc:401-411: bins 231..255 and 1..25). After the last LDS round a lane
// holds positions 4 lane + k, i.e. bins b + {0, 128, 64, 192} for k = 0..3 with b = brev6(lane): bins 64..191 are never kept, and of b and
// b + 192 at most one is (b in 1..25, or b in 39..63; the 14 lanes with b = 0 or 26..38 keep nothing). The last two stages therefore
// compute ONE output per lane instead of four:
//     bin b       = (r0 + w7 r2) + w8a (r1 + w7 r3)         bin b + 192 = (r0 - w7 r2) - w8b (r1 - w7 r3)
// (w7 = twiddle 2b for both stage-7 butterflies, w8a = twiddle b, w8b = twiddle b + 64) - 3 complex multiplications and 3 additions
// instead of 4 and 8. Every surviving value goes through the reference's own butterfly chain - the same operands, the same operations in
// the same order; a lane's choice between "u + t" and "u - t" is made by flipping t's sign bit (an XOR on the high word) in front of one
// written addition - so the bits are those of _fft_fast (ofdm.cc:310-340). What is constant per lane - b, which of the two bins, its
// carrier column, the twiddle slots, the sign mask - is computed once per kernel (fft256_carrier_lane), not per symbol.
struct Fft256CarrierLane {
    int col;             // carrier column 0..49 of the lane's live bin, -1: the lane keeps nothing
    uint32_t sign;       // 0x80000000 for the b + 192 form, 0 for the b form
    int slot7, slot8;    // LDS twiddle slots of w7 and w8a / w8b
};
__device__ __forceinline__ int carrier_of_bin(int bin);
__device__ __forceinline__ Fft256CarrierLane fft256_carrier_lane(int lane) {
    const int b = int(__brev(unsigned(lane)) >> 26);
    const bool upper = b >= 32;
    Fft256CarrierLane c;
    c.col = carrier_of_bin(upper ? b + 192 : b);
    c.sign = upper ? 0x80000000u : 0u;
    c.slot7 = fft256_tw_slot(2 * b);
    c.slot8 = fft256_tw_slot(upper ? b + 64 : b);
    return c;
}
// In: as wave_fft256. Returns the lane's live bin, unscaled (garbage-free but meaningless where c.col < 0).
__device__ __forceinline__ c2 wave_fft256_carriers(c2 r0, c2 r1, c2 r2, c2 r3, c2* __restrict__ v, const c2* __restrict__ tw, int lane,
                                                   const Fft256CarrierLane& c) {
    wave_fft256<true>(r0, r1, r2, r3, v, tw, lane);
    auto flip = [&](double x) { return __hiloint2double(int(uint32_t(__double2hiint(x)) ^ c.sign), __double2loint(x)); };
    const c2 w7 = fft256_twiddle(tw, c.slot7), w8 = fft256_twiddle(tw, c.slot8);
    c2 t = cmul(w7, r2), u = cmul(w7, r3);
    // a - t is a + (-t) in IEEE arithmetic (signed zeros included), so flipping the subtrahend's sign turns the one written addition into
    // the butterfly's "lo" (bin b) or "hi" (bin b + 192) output per lane: three flips of complex values, six XORs
    t = {flip(t.re), flip(t.im)};
    u = {flip(u.re), flip(u.im)};
    const c2 x = {r0.re + t.re, r0.im + t.im};                     // stage 7, butterfly (r0, r2): r0 +- w7 r2
    const c2 y = {r1.re + u.re, r1.im + u.im};                     // stage 7, butterfly (r1, r3): r1 +- w7 r3
    c2 z = cmul(w8, y);                                            // stage 8: x +- w8 y
    z = {flip(z.re), flip(z.im)};
    return {x.re + z.re, x.im + z.im};
}

// zero_depadder (ofdm.cc:401-411) for Nc = 50, start_shift = 1: carrier column of FFT bin `bin`, or -1
__device__ __forceinline__ int carrier_of_bin(int bin) {
    if (bin >= 256 - 25) return bin - (256 - 25);
    if (bin >= 1 && bin <= 25) return 25 + bin - 1;
    return -1;
}
