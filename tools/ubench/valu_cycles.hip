// Micro-benchmark: issue cost (shader cycles per wave64 instruction per SIMD) of the instruction kinds the
// sum-product decoder is made of, measured on the box it runs on. VERDICT r01 "next" item 1(a): the VALU
// ceiling of `roofline.secondary` must come from measured per-opcode costs, not from one constant.
//
// Method: every CU gets 2 workgroups of 1024 threads (= 8 waves per SIMD, the decoder's occupancy; also run
// with 1 wave per SIMD). Each wave executes REPS x 64 copies of one instruction on 8 independent register
// sets (no dependent chain shorter than 8 instructions), bracketed by s_memtime. With all 8 waves of a SIMD
// in the same loop the SIMD's issue port is the only resource, so
//     cycles per wave-instruction = (t1 - t0) / (REPS * 64 * waves_per_simd)
// using the slowest wave's interval of a workgroup as the SIMD's busy time. The wall-clock figure
// (instructions / SIMD / second) is printed beside it so the shader clock can be read off.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_cycles tools/ubench/valu_cycles.hip     Output: one JSON object.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#define REPS 4096

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// 8 independent 64-bit accumulators d0..d7 and 8 independent 32-bit ones i0..i7; operands x, y (double) and p, q (int).
#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define R64(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP) R8(OP)

#define KERNEL(NAME, BODY)                                                                                    \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t* __restrict__ out, double x, double y, int p, int q) { \
        double d0 = x + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6,  \
               d7 = d0 + 7;                                                                                   \
        int i0 = p + threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6,      \
            i7 = i0 + 7;                                                                                      \
        float f0 = float(d0), f1 = float(d1), f2 = float(d2), f3 = float(d3), f4 = float(d4), f5 = float(d5),      \
              f6 = float(d6), f7 = float(d7);                                                                 \
        const float fx = float(x), fy = float(y);                                                             \
        __shared__ double lds[2048];                                                                          \
        lds[threadIdx.x] = d0; lds[threadIdx.x + 1024] = d1;                                                  \
        const int la = (threadIdx.x * 8) & 16383;                                                             \
        (void)la; (void)fx; (void)fy; (void)f0; (void)f1; (void)f2; (void)f3; (void)f4; (void)f5; (void)f6; (void)f7;  \
        __syncthreads();                                                                                      \
        uint64_t t0, r0, r1;                                                                                  \
        r0 = __builtin_amdgcn_s_memrealtime();                                                                \
        t0 = __builtin_amdgcn_s_memtime();                                                                    \
        _Pragma("unroll 1") for (int r = 0; r < REPS; ++r) { R64(BODY) }                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                    \
        const uint64_t t1 = __builtin_amdgcn_s_memtime();                                                     \
        r1 = __builtin_amdgcn_s_memrealtime();                                                                \
        if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = r1 - r0;                                            \
        double s = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7) +                  \
                   double(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);                                             \
        if (s == 1.2345e-300) out[1] = 1;                                                                     \
        if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                  \
    }

#define B_FMA_F64(n) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d##n) : "v"(x), "v"(y));
#define B_MUL_F64(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##n) : "v"(x));
#define B_ADD_F64(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d##n) : "v"(x));
#define B_RCP_F64(n) asm volatile("v_rcp_f64 %0, %0" : "+v"(d##n));
#define B_RSQ_F64(n) asm volatile("v_rsq_f64 %0, %0" : "+v"(d##n));
#define B_SQRT_F64(n) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d##n));
#define B_CVT_I32_F64(n) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(i##n) : "v"(d##n));
#define B_CVT_F64_I32(n) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d##n) : "v"(i##n));
#define B_CMP_F64(n) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d##n), "v"(x) : "vcc");
#define B_CMP_U32(n) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(i##n), "v"(p) : "vcc");
#define B_CNDMASK(n) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(i##n) : "v"(q));
#define B_CNDMASK_E64(n) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(i##n) : "v"(q));
#define B_CMP_E64_CND(n) asm volatile("v_cmp_lt_u32_e64 s[20:21], %1, %2\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %3, s[20:21]" : "+v"(i##n) : "v"(i0), "v"(p), "v"(q) : "s20", "s21");
#define B_CMP_VCC_CND(n) asm volatile("v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(i##n) : "v"(i0), "v"(p), "v"(q) : "vcc");
#define B_AND_B32(n) asm volatile("v_and_b32 %0, %0, %1" : "+v"(i##n) : "v"(q));
#define B_ADD_U32(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(i##n) : "v"(q));
#define B_LSHL_ADD(n) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(i##n) : "v"(q));
#define B_MOV_B32(n) asm volatile("v_mov_b32 %0, %1" : "=v"(i##n) : "v"(q));
#define B_BFE_U32(n) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(i##n));
#define B_MUL_LO(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i##n) : "v"(q));
#define B_FMA_F32(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f##n) : "v"(fx), "v"(fy));
#define B_PK_FMA_F32(n) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d##n) : "v"(x), "v"(y));
#define B_MUL_F32(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f##n) : "v"(fx));
#define B_EXP_F32(n) asm volatile("v_exp_f32 %0, %0" : "+v"(f##n));
#define B_LOG_F32(n) asm volatile("v_log_f32 %0, %0" : "+v"(f##n));
#define B_RCP_F32(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(f##n));
#define B_MED3_F32(n) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f##n) : "v"(fx), "v"(fy));
#define B_MIN_F32(n) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f##n) : "v"(fx));
#define B_LDEXP_F64(n) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d##n) : "v"(q));
#define B_READLANE(n) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(i##n) : "s20");
#define B_DPP_MOV(n) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(i##n) : "v"(q));
#define B_DS_READ_B64(n) asm volatile("ds_read_b64 %0, %1" : "=v"(d##n) : "v"(la) : "memory");
#define B_DS_READ_B32(n) asm volatile("ds_read_b32 %0, %1" : "=v"(i##n) : "v"(la) : "memory");
#define B_DS_WRITE_B64(n) asm volatile("ds_write_b64 %0, %1" : : "v"(la), "v"(d##n) : "memory");
#define B_DS_WRITE_B32(n) asm volatile("ds_write_b32 %0, %1" : : "v"(la), "v"(i##n) : "memory");
#define B_DS_BPERMUTE(n) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(i##n) : "v"(la) : "memory");
#define B_SALU(n) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
// mixed: one fp64 FMA + one select per pair (does a 32-bit op hide inside the fp64 op's issue time?)
#define B_MIX_FMA_CND(n) asm volatile("v_fma_f64 %0, %2, %3, %0\n v_cndmask_b32 %1, %1, %4, vcc" : "+v"(d##n), "+v"(i##n) : "v"(x), "v"(y), "v"(q) : "vcc");
#define B_MIX_FMA_LDS(n) asm volatile("v_fma_f64 %0, %2, %3, %0\n ds_read_b64 %1, %4" : "+v"(d##n), "=v"(d7) : "v"(x), "v"(y), "v"(la) : "memory");

#define ALL(X)                                                                                                  \
    X(FMA_F64) X(MUL_F64) X(ADD_F64) X(RCP_F64) X(RSQ_F64) X(SQRT_F64) X(CVT_I32_F64) X(CVT_F64_I32) X(CMP_F64)  \
    X(CMP_U32) X(CNDMASK) X(CNDMASK_E64) X(CMP_E64_CND) X(CMP_VCC_CND) X(AND_B32) X(ADD_U32) X(LSHL_ADD) X(MOV_B32) X(BFE_U32) X(MUL_LO) X(FMA_F32)           \
    X(PK_FMA_F32) X(MUL_F32) X(EXP_F32) X(LOG_F32) X(RCP_F32) X(MED3_F32) X(MIN_F32) X(LDEXP_F64) X(READLANE)    \
    X(DPP_MOV) X(DS_READ_B64) X(DS_READ_B32) X(DS_WRITE_B64) X(DS_WRITE_B32) X(DS_BPERMUTE) X(SALU)              \
    X(MIX_FMA_CND)

#define DEF(N) KERNEL(N, B_##N)
ALL(DEF)

struct Entry { const char* name; void (*fn)(uint64_t*, double, double, int, int); int per_body; };
#define ENT(N) {#N, k_##N, 1},
static Entry entries[] = {ALL(ENT)};

int main(int argc, char** argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    uint64_t* d_out;
    const int max_blocks = cus * 2;
    CK(hipMalloc(&d_out, (2 + size_t(max_blocks) * 16) * 8));
    std::vector<uint64_t> h(2 + size_t(max_blocks) * 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("{\"device\": \"%s\", \"cus\": %d, \"reps\": %d, \"results\": {\n", prop.gcnArchName, cus, REPS);
    bool first = true;
    for (const Entry& e : entries) {
        for (int occ = 0; occ < 2; ++occ) {
            // occ 0: 2 workgroups of 1024 per CU = 8 waves per SIMD; occ 1: 256-thread workgroups, one per CU = 1 wave per SIMD
            const int threads = occ == 0 ? 1024 : 256, blocks = occ == 0 ? cus * 2 : cus;
            const int waves_per_simd = occ == 0 ? 8 : 1;
            CK(hipMemset(d_out, 0, h.size() * 8));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001, 1e-9, 12345, 77);   // warm
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001, 1e-9, 12345, 77);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> per;
            for (int b = 0; b < blocks; ++b) {
                uint64_t mx = 0;
                for (int w = 0; w < threads / 64; ++w) mx = std::max(mx, h[2 + size_t(b) * 16 + w]);
                per.push_back(double(mx));
            }
            std::sort(per.begin(), per.end());
            const double med = per[per.size() / 2];
            const double n_inst = double(REPS) * 64;
            const double ticks_per_inst = med / (n_inst * waves_per_simd);
            const double inst_per_simd_per_s = n_inst * waves_per_simd * (occ == 0 ? 1.0 : 1.0) / (ms * 1e-3) * (occ == 0 ? 1.0 : 1.0);
            // occ 0 puts 2 workgroups x 16 waves on 4 SIMDs = 8 waves per SIMD; the launch covers every SIMD once
            // s_memrealtime runs at 100 MHz; the wave's memtime interval over its realtime interval gives the tick rate of
            // s_memtime, the kernel's wall time gives the issue rate per SIMD; cycles at the clock the box reports = 2.4 GHz
            const double real_s = double(h[0]) / 100e6;
            printf("%s  \"%s/%dw\": {\"memtime_ticks_per_wave_inst\": %.4f, \"memtime_tick_hz\": %.4e, \"kernel_ms\": %.4f, "
                   "\"wave_inst_per_simd_per_s\": %.4e, \"cycles_at_2p4ghz\": %.3f}",
                   first ? "" : ",\n", e.name, waves_per_simd, ticks_per_inst, double(h[2]) / real_s, ms, inst_per_simd_per_s,
                   2.4e9 / inst_per_simd_per_s);
            first = false;
        }
    }
    printf("\n}}\n");
    return 0;
}
