// Host side of include/mercury_stages.h: copy in, one stage kernel (stages.hip), copy out.
#include "../../include/mercury_stages.h"
#include "ctx.hpp"

extern "C" __global__ void mgpu_stage_symbol_demod_kernel(const double*, int, const double*, double*);
extern "C" __global__ void mgpu_stage_agc_kernel(MgpuDev, double*);
extern "C" __global__ void mgpu_stage_estimate_kernel(MgpuDev, const double*, double*);
extern "C" __global__ void mgpu_stage_restore_amplitude_kernel(MgpuDev, double*);
extern "C" __global__ void mgpu_stage_equalize_kernel(MgpuDev, const double*, const double*, double*);
extern "C" __global__ void mgpu_stage_variance_kernel(MgpuDev, const double*, int, double*);
extern "C" __global__ void mgpu_stage_deframe_kernel(MgpuDev, const double*, double*);
extern "C" __global__ void mgpu_stage_deinterleave_kernel(const unsigned char*, int, int, int, unsigned char*);
extern "C" __global__ void mgpu_stage_psk_demod_kernel(MgpuDev, const double*, const float*, float*);

extern "C" __global__ void mgpu_stage_dispersal_kernel(const uint8_t*, const uint8_t*, int, uint8_t*);
extern "C" __global__ void mgpu_stage_bit_to_byte_kernel(const uint8_t*, int, uint8_t*);
extern "C" __global__ void mgpu_stage_crc16_kernel(const uint8_t*, int, int, uint16_t*);

namespace {

// run `launch(d_in..., d_out)` between an upload of the inputs and a download of the output
struct Io {
    mgpu_ctx* c;
    hipStream_t s;
    explicit Io(mgpu_ctx* ctx) : c(ctx), s(ctx->stream) {}
    void up(DevBuf& d, const void* h, size_t bytes) { HIPCK(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, s)); }
    void down(void* h, DevBuf& d, size_t bytes) { HIPCK(hipMemcpyAsync(h, d.p, bytes, hipMemcpyDeviceToHost, s)); HIPCK(hipStreamSynchronize(s)); }
};

void ofdm_only(mgpu_ctx* c) { need(c->tab.mfsk_M == 0, "the per-method stages exist for the OFDM modes (cfg 0..16)"); }

}  // namespace

extern "C" {

int mgpu_symbol_demod(mgpu_ctx* c, const double* in, int n, double* out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(in && out && n > 0, "bad argument");
        const auto& t = c->tab;
        Io io(c);
        DevBuf d_in(size_t(n) * t.Nofdm * 16), d_out(size_t(n) * t.Nc * 16);
        io.up(d_in, in, size_t(n) * t.Nofdm * 16);
        hipLaunchKernelGGL(mgpu_stage_symbol_demod_kernel, dim3((n + 3) / 4), dim3(256), 0, io.s, d_in.as<double>(), n, c->dev.twiddle, d_out.as<double>());
        HIPCK(hipGetLastError());
        io.down(out, d_out, size_t(n) * t.Nc * 16);
    });
}

int mgpu_automatic_gain_control(mgpu_ctx* c, double* grid, int F) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(grid && F > 0, "bad argument");
        ofdm_only(c);
        const size_t bytes = size_t(F) * c->dev.G * 16;
        Io io(c);
        DevBuf d(bytes);
        io.up(d, grid, bytes);
        hipLaunchKernelGGL(mgpu_stage_agc_kernel, dim3(F), dim3(256), 0, io.s, c->dev, d.as<double>());
        HIPCK(hipGetLastError());
        io.down(grid, d, bytes);
    });
}

int mgpu_channel_estimator(mgpu_ctx* c, const double* grid, int F, double* H) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(grid && H && F > 0, "bad argument");
        ofdm_only(c);
        const size_t bytes = size_t(F) * c->dev.G * 16;
        Io io(c);
        DevBuf d_in(bytes), d_out(bytes);
        io.up(d_in, grid, bytes);
        hipLaunchKernelGGL(mgpu_stage_estimate_kernel, dim3(F), dim3(256), 0, io.s, c->dev, d_in.as<double>(), d_out.as<double>());
        HIPCK(hipGetLastError());
        io.down(H, d_out, bytes);
    });
}

int mgpu_restore_channel_amplitude(mgpu_ctx* c, double* H, int F) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(H && F > 0, "bad argument");
        ofdm_only(c);
        const size_t bytes = size_t(F) * c->dev.G * 16;
        Io io(c);
        DevBuf d(bytes);
        io.up(d, H, bytes);
        hipLaunchKernelGGL(mgpu_stage_restore_amplitude_kernel, dim3(F), dim3(256), 0, io.s, c->dev, d.as<double>());
        HIPCK(hipGetLastError());
        io.down(H, d, bytes);
    });
}

int mgpu_channel_equalizer(mgpu_ctx* c, const double* grid, const double* H, int F, double* out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(grid && H && out && F > 0, "bad argument");
        ofdm_only(c);
        const size_t bytes = size_t(F) * c->dev.G * 16;
        Io io(c);
        DevBuf d_g(bytes), d_h(bytes), d_o(bytes);
        io.up(d_g, grid, bytes);
        io.up(d_h, H, bytes);
        hipLaunchKernelGGL(mgpu_stage_equalize_kernel, dim3(F), dim3(256), 0, io.s, c->dev, d_g.as<double>(), d_h.as<double>(), d_o.as<double>());
        HIPCK(hipGetLastError());
        io.down(out, d_o, bytes);
    });
}

int mgpu_measure_variance(mgpu_ctx* c, const double* grid, int F, double* variance) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(grid && variance && F > 0, "bad argument");
        ofdm_only(c);
        const size_t bytes = size_t(F) * c->dev.G * 16;
        Io io(c);
        DevBuf d_g(bytes), d_v(size_t(F) * 8);
        io.up(d_g, grid, bytes);
        hipLaunchKernelGGL(mgpu_stage_variance_kernel, dim3((F + 63) / 64), dim3(64), 0, io.s, c->dev, d_g.as<double>(), F, d_v.as<double>());
        HIPCK(hipGetLastError());
        io.down(variance, d_v, size_t(F) * 8);
    });
}

int mgpu_deframer(mgpu_ctx* c, const double* grid, int F, double* data) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(grid && data && F > 0, "bad argument");
        ofdm_only(c);
        const auto& t = c->tab;
        Io io(c);
        DevBuf d_g(size_t(F) * c->dev.G * 16), d_o(size_t(F) * t.nData * 16);
        io.up(d_g, grid, size_t(F) * c->dev.G * 16);
        hipLaunchKernelGGL(mgpu_stage_deframe_kernel, dim3(F), dim3(256), 0, io.s, c->dev, d_g.as<double>(), d_o.as<double>());
        HIPCK(hipGetLastError());
        io.down(data, d_o, size_t(F) * t.nData * 16);
    });
}

static int deinterleave(mgpu_ctx* c, const void* in, int F, int n, int bs, int elem, void* out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(in && out && F > 0 && n > 0 && bs > 0 && bs <= n, "bad argument");
        const size_t bytes = size_t(F) * n * elem;
        Io io(c);
        DevBuf d_i(bytes), d_o(bytes);
        io.up(d_i, in, bytes);
        hipLaunchKernelGGL(mgpu_stage_deinterleave_kernel, dim3(F), dim3(256), 0, io.s, d_i.as<unsigned char>(), n, bs, elem, d_o.as<unsigned char>());
        HIPCK(hipGetLastError());
        io.down(out, d_o, bytes);
    });
}
int mgpu_deinterleaver_c128(mgpu_ctx* c, const double* in, int F, int n, int bs, double* out) { return deinterleave(c, in, F, n, bs, 16, out); }
int mgpu_deinterleaver_f32(mgpu_ctx* c, const float* in, int F, int n, int bs, float* out) { return deinterleave(c, in, F, n, bs, 4, out); }

int mgpu_psk_demod(mgpu_ctx* c, const double* syms, int F, const float* variance, float* llr) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(syms && variance && llr && F > 0, "bad argument");
        ofdm_only(c);
        const auto& t = c->tab;
        Io io(c);
        DevBuf d_s(size_t(F) * t.nData * 16), d_v(size_t(F) * 4), d_l(size_t(F) * t.nBits * 4);
        io.up(d_s, syms, size_t(F) * t.nData * 16);
        io.up(d_v, variance, size_t(F) * 4);
        hipLaunchKernelGGL(mgpu_stage_psk_demod_kernel, dim3(F), dim3(256), 0, io.s, c->dev, d_s.as<double>(), d_v.as<float>(), d_l.as<float>());
        HIPCK(hipGetLastError());
        io.down(llr, d_l, size_t(F) * t.nBits * 4);
    });
}

int mgpu_bit_energy_dispersal(mgpu_ctx* c, const uint8_t* bits, int F, int n, uint8_t* out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(bits && out && F > 0 && n > 0 && n <= c->tab.N, "bad argument");
        Io io(c);
        DevBuf d_i(size_t(F) * n), d_o(size_t(F) * n);
        io.up(d_i, bits, size_t(F) * n);
        hipLaunchKernelGGL(mgpu_stage_dispersal_kernel, dim3(F), dim3(256), 0, io.s, d_i.as<uint8_t>(), c->dev.scrambler, n, d_o.as<uint8_t>());
        HIPCK(hipGetLastError());
        io.down(out, d_o, size_t(F) * n);
    });
}

int mgpu_bit_to_byte(mgpu_ctx* c, const uint8_t* bits, int F, int nbits, uint8_t* bytes) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(bits && bytes && F > 0 && nbits > 0, "bad argument");
        const int nbytes = (nbits + 7) / 8;
        Io io(c);
        DevBuf d_i(size_t(F) * nbits), d_o(size_t(F) * nbytes);
        io.up(d_i, bits, size_t(F) * nbits);
        hipLaunchKernelGGL(mgpu_stage_bit_to_byte_kernel, dim3(F), dim3(256), 0, io.s, d_i.as<uint8_t>(), nbits, d_o.as<uint8_t>());
        HIPCK(hipGetLastError());
        io.down(bytes, d_o, size_t(F) * nbytes);
    });
}

int mgpu_crc16_modbus_rtu(mgpu_ctx* c, const uint8_t* bytes, int F, int n, uint16_t* crc) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(bytes && crc && F > 0 && n >= 0, "bad argument");
        Io io(c);
        DevBuf d_i(size_t(F) * n + 16), d_o(size_t(F) * 2);
        io.up(d_i, bytes, size_t(F) * n);
        hipLaunchKernelGGL(mgpu_stage_crc16_kernel, dim3((F + 63) / 64), dim3(64), 0, io.s, d_i.as<uint8_t>(), F, n, d_o.as<uint16_t>());
        HIPCK(hipGetLastError());
        io.down(crc, d_o, size_t(F) * 2);
    });
}

}  // extern "C"
