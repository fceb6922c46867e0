// Micro-benchmark: what an LDS read costs the CU's LDS pipeline as a function of the lanes that are active when it issues.
// Every CU gets one workgroup of 1024 threads; every wavefront issues REPS x 16 ds_read_b128 (all lanes the SAME address: the
// decoder's product walk is a broadcast read) under an execution mask of 64 / 48 / 32 / 16 / 4 / 1 lanes, and the same with per-lane
// consecutive addresses. Output: LDS cycles per instruction and CU (slowest wavefront's s_memtime interval / instructions issued by the CU).
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_mask tools/ubench/lds_mask.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define REPS 2048
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BYTES>
__global__ __launch_bounds__(1024) void k(uint64_t* out, unsigned long long mask, int stride, double seed) {
    __shared__ __attribute__((aligned(16))) double lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = seed + i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t addr = uint32_t(uintptr_t(lds)) + uint32_t(lane * stride) + uint32_t((threadIdx.x >> 6) * 1024);
    double acc0 = 0, acc1 = 0;
    const bool on = (mask >> lane) & 1;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    if (on) {
#pragma unroll 1
        for (int r = 0; r < REPS; ++r) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if constexpr (BYTES == 16) {
                    double a, b;
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(*(reinterpret_cast<__attribute__((ext_vector_type(2))) double*>(&a))) : "v"(addr), "n"(0));
                    (void)b;
                    asm volatile("" :: "v"(a));
                } else if constexpr (BYTES == 4) {
                    float a;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(a) : "v"(addr));
                    asm volatile("" :: "v"(a));
                } else if constexpr (BYTES == 5) {        // ds_read2_b32 offsets 0 and 1 (x 4 bytes)
                    __attribute__((ext_vector_type(2))) float a;
                    asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(a) : "v"(addr));
                    asm volatile("" :: "v"(a));
                } else if constexpr (BYTES == 24) {       // ds_write_b64
                    asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(seed));
                } else if constexpr (BYTES == 25) {       // ds_write2_b64 offsets 0 and 1
                    asm volatile("ds_write2_b64 %0, %1, %1 offset1:1" :: "v"(addr), "v"(seed));
                } else if constexpr (BYTES == 26) {       // ds_write_b128
                    __attribute__((ext_vector_type(2))) double w = {seed, seed};
                    asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(w));
                } else if constexpr (BYTES == 17) {       // ds_read2_b64: two 8-byte reads, offsets 0 and 1 (x 8 bytes)
                    __attribute__((ext_vector_type(2))) double a;
                    asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(a) : "v"(addr));
                    asm volatile("" :: "v"(a));
                } else if constexpr (BYTES == 18) {       // ds_read2_b64 with the two reads far apart
                    __attribute__((ext_vector_type(2))) double a;
                    asm volatile("ds_read2_b64 %0, %1 offset1:33" : "=v"(a) : "v"(addr));
                    asm volatile("" :: "v"(a));
                } else {
                    double a;
                    asm volatile("ds_read_b64 %0, %1" : "=v"(a) : "v"(addr));
                    asm volatile("" :: "v"(a));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if (acc0 + acc1 == 1.2345e-300) out[0] = 1;
    if (lane == 0) out[1 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    int cus = 256;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); cus = p.multiProcessorCount;
    uint64_t* d; CK(hipMalloc(&d, (1 + cus * 16) * 8));
    std::vector<uint64_t> h(1 + cus * 16);
    struct M { const char* name; unsigned long long m; } masks[] = {{"64", ~0ull}, {"48 (lanes 0-47)", (1ull << 48) - 1}, {"32 (lanes 0-31)", 0xffffffffull}, {"16 (lanes 0-15)", 0xffffull},
                                                               {"16 (every 4th)", 0x1111111111111111ull}, {"4 (one per row)", 0x0001000100010001ull}, {"1", 1ull}};
    printf("{\"device\": \"%s\", \"note\": \"LDS-pipeline cycles per instruction and CU (16 wavefronts of one workgroup issuing back to back; s_memtime ticks of the slowest wavefront / (16 x instructions per wavefront))\", \"results\": {\n", p.gcnArchName);
    bool first = true;
    for (int bytes : {8, 16, 17, 18, 4, 5, 24, 25, 26}) for (int stride : {0, bytes == 4 ? 4 : (bytes == 8 || bytes == 5 || bytes == 24) ? 8 : 16}) for (auto& mk : masks) {
        if (bytes != 8 && bytes != 16 && mk.m != ~0ull) continue;
        for (int rep = 0; rep < 2; ++rep) {
            if (bytes == 16) hipLaunchKernelGGL(k<16>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else if (bytes == 17) hipLaunchKernelGGL(k<17>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else if (bytes == 18) hipLaunchKernelGGL(k<18>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else if (bytes == 4) hipLaunchKernelGGL(k<4>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else if (bytes == 5) hipLaunchKernelGGL(k<5>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else if (bytes == 24) hipLaunchKernelGGL(k<24>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else if (bytes == 25) hipLaunchKernelGGL(k<25>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else if (bytes == 26) hipLaunchKernelGGL(k<26>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            else hipLaunchKernelGGL(k<8>, dim3(cus), dim3(1024), 0, 0, d, mk.m, stride, 1.0);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> per;
        for (int b = 0; b < cus; ++b) { uint64_t mx = 0; for (int w = 0; w < 16; ++w) mx = std::max(mx, h[1 + b * 16 + w]); per.push_back(double(mx) / (16.0 * REPS * 16)); }
        std::sort(per.begin(), per.end());
        printf("%s  \"%s %s, active lanes %s\": %.3f", first ? "" : ",\n", bytes == 8 ? "ds_read_b64" : bytes == 16 ? "ds_read_b128" : bytes == 17 ? "ds_read2_b64 offset1:1" : bytes == 18 ? "ds_read2_b64 offset1:33" : bytes == 4 ? "ds_read_b32" : bytes == 5 ? "ds_read2_b32 offset1:1" : bytes == 24 ? "ds_write_b64" : bytes == 25 ? "ds_write2_b64 offset1:1" : "ds_write_b128", stride ? "consecutive addresses" : "one address (broadcast)", mk.name, per[per.size() / 2]);
        first = false;
    }
    printf("\n}}\n");
    return 0;
}
