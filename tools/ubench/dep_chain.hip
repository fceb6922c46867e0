// Latency of a dependent fp64 addition chain on one wavefront (what bounds the sequential |x|^2 sums of measure_signal_stregth and the
// front-end's in-order reductions): N dependent v_add_f64 (and v_fma_f64, v_add_f32 for comparison), timed with s_memtime.
//   hipcc --offload-arch=gfx950 -O2 -o dep_chain dep_chain.hip && ./dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void chain(double* out, uint64_t* cyc, double x, int reps) {
    double a = out[0];
    float af = float(a), xf = float(x);
    uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(x));
            if (MODE == 1) asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(a) : "v"(x));
            if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(af) : "v"(xf));
            if (MODE == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(x));
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[1] = a + double(af); cyc[0] = t1 - t0; }
}
int main() {
    double* d; uint64_t* c;
    hipMalloc(&d, 16); hipMalloc(&c, 8);
    double h[2] = {1.0, 0.0};
    const char* names[4] = {"v_add_f64", "v_fma_f64", "v_add_f32", "v_mul_f64"};
    for (int lanes : {64, 1}) {
        for (int m = 0; m < 4; ++m) {
            hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
            const int reps = 4096;
            for (int pass = 0; pass < 2; ++pass) {
                if (m == 0) chain<0><<<1, lanes>>>(d, c, 1e-9, reps);
                if (m == 1) chain<1><<<1, lanes>>>(d, c, 1e-9, reps);
                if (m == 2) chain<2><<<1, lanes>>>(d, c, 1e-9, reps);
                if (m == 3) chain<3><<<1, lanes>>>(d, c, 1.0000001, reps);
                hipDeviceSynchronize();
            }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (m == 0) chain<0><<<1, lanes>>>(d, c, 1e-9, reps);
            if (m == 1) chain<1><<<1, lanes>>>(d, c, 1e-9, reps);
            if (m == 2) chain<2><<<1, lanes>>>(d, c, 1e-9, reps);
            if (m == 3) chain<3><<<1, lanes>>>(d, c, 1.0000001, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            uint64_t cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            printf("{\"op\": \"%s\", \"lanes\": %d, \"dependent_ops\": %d, \"counter_ticks_per_op\": %.2f, \"ns_per_op\": %.2f}\n", names[m], lanes, reps * 64,
                   double(cy) / (reps * 64.0), ms * 1e6 / (reps * 64.0));
        }
    }
    return 0;
}
