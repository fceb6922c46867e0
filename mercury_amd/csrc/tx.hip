// transmit_byte on the GPU: kernels behind include/mercury_tx.h and their host entry points.
//
//   payload --txgen kernel (txgen.hip: CRC .. IFFT+GI, caller's bytes instead of the generator's)--> data baseband
//   preamble carriers --symbol_mod kernel (once per context)--> preamble baseband
//   both --tx_mix kernel: power scaling, x4 linear interpolation, mixer--> passband
//   --peak_clip kernel (preamble part, data part)--> --fir kernel x2 (SINGLE_MESSAGE)--> out
//
// Everything is FP64 in the reference's operation order (no FMA contraction in this TU), so the samples equal the
// reference's bit for bit; the carrier cos/sin and the two pow() constants come from the host libm like the reference's.
#include <cmath>
#include <cstring>
#include <memory>
#include <map>

#include "ctx.hpp"
#include "fft256.h"
#include "../../include/mercury_tx.h"

// cl_ofdm::symbol_mod (ofdm.cc:855-860), one wavefront per symbol
extern "C" __global__ __launch_bounds__(256) void mgpu_symbol_mod_kernel(const double* __restrict__ twiddle, const double* __restrict__ carriers,
                                                                       int n, double* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) c2 fftb[4 * FFT256_STRIDE];
    __shared__ __attribute__((aligned(16))) c2 tw[128];      // 16-byte aligned: fft256_twiddle reads it as ds_read_b128
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128; i += 256) tw[fft256_tw_slot(i)] = {twiddle[2 * i], -twiddle[2 * i + 1]};       // conjugated: IFFT (ofdm.cc:365)
    __syncthreads();
    const int s = blockIdx.x * 4 + wave;
    if (s >= n) return;
    const c2* in = reinterpret_cast<const c2*>(carriers) + size_t(s) * 50;
    auto carrier = [&](int i) -> c2 {                    // zero_padder (ofdm.cc:379-400)
        const int col = carrier_of_bin(i);
        return col < 0 ? c2{0.0, 0.0} : in[col];
    };
    c2 r0 = carrier(lane), r1 = carrier(lane + 64), r2 = carrier(lane + 128), r3 = carrier(lane + 192);
    wave_fft256(r0, r1, r2, r3, fftb + wave * FFT256_STRIDE, tw, lane);
    c2* y = reinterpret_cast<c2*>(out) + size_t(s) * 272;
    auto emit = [&](const c2& x, int p) {                // gi_adder (ofdm.cc:412-422)
        const int k = brev8(p);
        y[k + 16] = x;
        if (k >= 240) y[k - 240] = x;
    };
    emit(r0, 4 * lane); emit(r1, 4 * lane + 1); emit(r2, 4 * lane + 2); emit(r3, 4 * lane + 3);
}

// Power scaling (telecom_system.cc:517-527), rational_resampler INTERPOLATION (ofdm.cc:2279-2291, interpolate_linear
// interpolator.cc:43-50) and the mixer (ofdm.cc:2309-2314), one thread per passband sample. The preamble and the data part
// are interpolated separately (two baseband_to_passband calls, :531-532), each extrapolating its own last sample.
//   pre_bb  [npre]            preamble baseband, shared by every frame
//   data_bb [F][data_stride]  data baseband (the first ndata samples of each row)
//   cs      [..][2]           cos / sin of the carrier phase for sample cs_frame_stride * f + n
extern "C" __global__ __launch_bounds__(256) void mgpu_tx_mix_kernel(
    const double* __restrict__ pre_bb, int npre, const double* __restrict__ data_bb, int data_stride, int ndata, double pn, double m_pre,
    double m_data, double amplitude, const double* __restrict__ cs, int cs_frame_stride, double* __restrict__ out, int total) {
    const int f = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
    if (n >= total) return;
    double* o = out + size_t(f) * total;
    const int used = (npre + ndata) * 4;
    if (n >= used) { o[n] = 0.0; return; }
    const bool in_pre = n < npre * 4;
    const c2* in = in_pre ? reinterpret_cast<const c2*>(pre_bb) : reinterpret_cast<const c2*>(data_bb) + size_t(f) * data_stride;
    const int cnt = in_pre ? npre : ndata, k = in_pre ? n : n - npre * 4, i = k >> 2, j = k & 3;
    const double m = in_pre ? m_pre : m_data;
    const bool last = i == cnt - 1;
    const c2 xa = in[last ? cnt - 2 : i], xb = in[last ? cnt - 1 : i + 1];
    const c2 a = {(xa.re / pn) * m, (xa.im / pn) * m}, b = {(xb.re / pn) * m, (xb.im / pn) * m};
    const double x = double(last ? 4 + j : j);
    const c2 v = {a.re + ((b.re - a.re) * x) / 4.0, a.im + ((b.im - a.im) * x) / 4.0};
    const double* c = cs + 2 * (size_t(cs_frame_stride) * f + n);
    double y = v.re * amplitude * c[0];
    y += v.im * amplitude * c[1];
    o[n] = y;
}

// cl_ofdm::peak_clip (ofdm.cc:1565-1592) on one segment of every frame: blockIdx.y = 0 the preamble part, 1 the data part.
// The mean power is a sum in sample order, i.e. one dependent chain of additions per segment (26,112 for a mode-8 data part,
// 348,160 for ROBUST_0). One lane walks the chain through LDS while the other three wavefronts of the workgroup form the next
// 1024 terms (double buffer), so the chain never waits for memory; the chains of a batch run side by side (8 per CU).
#define PC_CHUNK 1024
extern "C" __global__ __launch_bounds__(256) void mgpu_peak_clip_kernel(double* __restrict__ x, int total, int npre4, int used, double pow_pre,
                                                                      double pow_data) {
    __shared__ double term[2][PC_CHUNK];
    __shared__ double peak_s;
    const int seg = blockIdx.y, tid = threadIdx.x;
    double* p = x + size_t(blockIdx.x) * total + (seg ? npre4 : 0);
    const int n = seg ? used - npre4 : npre4;
    if (n <= 0) return;
    for (int i = tid; i < min(PC_CHUNK, n); i += 256) { const double v = p[i]; term[0][i] = v * v; }
    __syncthreads();
    double acc = 0.0;
    for (int base = 0, k = 0; base < n; base += PC_CHUNK, ++k) {
        const int m = min(PC_CHUNK, n - base);
        if (tid >= 64) {                                  // producers: the chunk after this one
            const int nb = base + PC_CHUNK, mn = min(PC_CHUNK, n - nb);
            for (int i = tid - 64; i < mn; i += 192) { const double v = p[nb + i]; term[(k + 1) & 1][i] = v * v; }
        } else if (tid == 0) {                            // the chain
            const double* t = term[k & 1];
            int q = 0;
            for (; q + 8 <= m; q += 8) {
                const double a0 = t[q], a1 = t[q + 1], a2 = t[q + 2], a3 = t[q + 3];
                const double a4 = t[q + 4], a5 = t[q + 5], a6 = t[q + 6], a7 = t[q + 7];
                acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;
            }
            for (; q < m; ++q) acc += t[q];
        }
        __syncthreads();
    }
    if (tid == 0) peak_s = sqrt((acc / double(n)) * (seg ? pow_data : pow_pre));
    __syncthreads();
    const double peak = peak_s;
    for (int i = tid; i < n; i += 256) {
        double v = p[i];
        if (v > 0 && v > peak) v = peak;
        if (v < 0 && v < -peak) v = -peak;
        p[i] = v;
    }
}

// cl_FIR::apply(double*) (fir_filter.cc:189-210): out[i] = sum_j in[i + h - j] * c[j] over the taps whose sample exists, added in
// tap order — for the tap count the reference's design always yields (97 at 48 kHz with a 1 kHz transition band), shaped for
// the hardware: 1024 outputs per workgroup, 4 consecutive outputs per thread. Output i0+4t+r needs inputs 4t + m, m = r + 96 - j,
// so the tile is stored by (m & 3, m >> 2): at every step all lanes read consecutive doubles (no bank conflicts), each thread
// walks a sliding window of 100 inputs through registers (one LDS read per 4 multiply-adds instead of 8) and the taps arrive
// through scalar loads. Samples outside the frame are zero in the tile: adding their +-0 products changes no sum (the
// accumulators start at +0 and never become -0), so this equals the reference's skipping of those taps bit for bit.
#define FIR97_NT 97
#define FIR97_OUT 1024
#define FIR97_S 281                             // doubles per residue class: (1024 + 96) / 4 = 280, + 1
// Outputs [out_begin, out_begin + out_count) of each row are written, to out[row][i - out_begin] (the batch form filters a
// padded concatenation and keeps its middle).
extern "C" __global__ __launch_bounds__(256) void mgpu_fir97_kernel(const double* __restrict__ in, int n, const double* __restrict__ taps,
                                                                  double* __restrict__ out, int out_begin, int out_count) {
    __shared__ double tile[4 * FIR97_S];
    const int t = threadIdx.x, h = (FIR97_NT - 1) / 2, i0 = out_begin + blockIdx.x * FIR97_OUT, lo = i0 + h - (FIR97_NT - 1);
    const double* x = in + size_t(blockIdx.y) * n;
    for (int e = t; e < FIR97_OUT + FIR97_NT - 1; e += 256) {
        const int q = lo + e;
        tile[(e & 3) * FIR97_S + (e >> 2)] = (q >= 0 && q < n) ? x[q] : 0.0;
    }
    __syncthreads();
    const double* w = tile + t;
#define FIR97_W(m) w[((m) & 3) * FIR97_S + ((m) >> 2)]
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    double w0 = FIR97_W(96), w1 = FIR97_W(97), w2 = FIR97_W(98), w3 = FIR97_W(99);    // step j works on inputs 96-j .. 99-j
    // 4 rounds of 24 taps (a round's taps fit the scalar registers; the window pointer moves back 6 doubles per round), then tap 96
#pragma unroll 1
    for (int jb = 0; jb < FIR97_NT - 1; jb += 24) {
        const double* c = taps + jb;
#pragma unroll
        for (int jj = 0; jj < 24; ++jj) {
            const double cj = c[jj];
            a0 += w0 * cj; a1 += w1 * cj; a2 += w2 * cj; a3 += w3 * cj;
            w3 = w2; w2 = w1; w1 = w0; w0 = FIR97_W(95 - jj);
        }
        w -= 6;
    }
    {
        const double cj = taps[FIR97_NT - 1];
        a0 += w0 * cj; a1 += w1 * cj; a2 += w2 * cj; a3 += w3 * cj;
    }
#undef FIR97_W
    const int i = blockIdx.x * FIR97_OUT + 4 * t;            // relative to out_begin
    double* y = out + size_t(blockIdx.y) * out_count + i;
    if (i + 3 < out_count) { y[0] = a0; y[1] = a1; y[2] = a2; y[3] = a3; }
    else { if (i < out_count) y[0] = a0; if (i + 1 < out_count) y[1] = a1; if (i + 2 < out_count) y[2] = a2; }
}

using namespace mgpu_detail;

namespace {

// per-context transmit state: preamble baseband, filter taps and carrier table for the last carrier / phase origin used
struct TxState {
    double* d_pre_bb = nullptr;
    double carrier = -1;
    double* d_fir[2] = {nullptr, nullptr};
    int ntaps[2] = {0, 0};
    double* d_cs = nullptr;
    size_t cs_cap = 0;
    int pre_eq_version = 0;                             // the context's pre_eq_version d_pre_bb was built with
    double* d_tx_buffer = nullptr;                      // passband_data_tx_buffer: 3 frames of unfiltered audio carried between the
                                                        // FIRST / MIDDLE / FLUSH_MESSAGE calls (telecom_system.cc:559-590); zero at first
    void* d_work[3] = {nullptr, nullptr, nullptr};      // data baseband, clipped passband, first filter's output: grown on demand, kept
    size_t work_cap[3] = {0, 0, 0};
    void* work(int i, size_t bytes, hipStream_t s) {
        if (work_cap[i] < bytes) {
            HIPCK(hipStreamSynchronize(s));
            (void)hipFree(d_work[i]);
            d_work[i] = nullptr; work_cap[i] = 0;
            HIPCK(hipMalloc(&d_work[i], bytes));
            work_cap[i] = bytes;
        }
        return d_work[i];
    }
    double cs_carrier = -1;
    uint64_t cs_start = 0;
    size_t cs_count = 0;
    ~TxState() {
        (void)hipFree(d_pre_bb); (void)hipFree(d_fir[0]); (void)hipFree(d_fir[1]); (void)hipFree(d_cs); (void)hipFree(d_tx_buffer);
        for (void* p : d_work) (void)hipFree(p);
    }
};
void free_tx_state(void* p) { delete static_cast<TxState*>(p); }

void launch_symbol_mod(mgpu_ctx* c, const double* d_carriers, int n, double* d_out, hipStream_t s) {
    hipLaunchKernelGGL(mgpu_symbol_mod_kernel, dim3((n + 3) / 4), dim3(256), 0, s, c->dev.twiddle, d_carriers, n, d_out);
    HIPCK(hipGetLastError());
}

// the preamble's baseband (symbol_mod of the preamble carriers, with the installed pre_equalization_channel: telecom_system.cc:477-484)
void build_preamble_baseband(mgpu_ctx* c, TxState* st, hipStream_t s) {
    const auto& t = c->tab;
    std::vector<mgpu::Cplx> car = t.preamble_carriers;
    if (!c->pre_eq.empty() && t.mfsk_M == 0)
        for (size_t i = 0; i < car.size(); ++i) {
            const double hr = c->pre_eq[2 * (i % t.Nc)], hi = c->pre_eq[2 * (i % t.Nc) + 1], gr = car[i].re, gi = car[i].im;
            car[i] = {gr * hr - gi * hi, gr * hi + gi * hr};
        }
    DevBuf d_car(car.size() * 16);
    HIPCK(hipMemcpyAsync(d_car.p, car.data(), car.size() * 16, hipMemcpyHostToDevice, s));
    launch_symbol_mod(c, d_car.as<double>(), t.preamble, st->d_pre_bb, s);
    HIPCK(hipStreamSynchronize(s));
    st->pre_eq_version = c->pre_eq_version;
}

TxState& tx_state(mgpu_ctx* c, hipStream_t s) {
    if (c->tx_state && static_cast<TxState*>(c->tx_state)->pre_eq_version != c->pre_eq_version)
        build_preamble_baseband(c, static_cast<TxState*>(c->tx_state), s);
    if (!c->tx_state) {
        std::unique_ptr<TxState> st(new TxState);            // published in the context only once it is complete
        const auto& t = c->tab;
        HIPCK(hipMalloc(reinterpret_cast<void**>(&st->d_pre_bb), size_t(t.preamble) * t.Nofdm * 16));
        build_preamble_baseband(c, st.get(), s);
        const size_t total = size_t(t.Nofdm) * (t.Nsymb + t.preamble) * 4;
        HIPCK(hipMalloc(reinterpret_cast<void**>(&st->d_tx_buffer), 3 * total * 8));
        HIPCK(hipMemsetAsync(st->d_tx_buffer, 0, 3 * total * 8, s));
        HIPCK(hipStreamSynchronize(s));
        c->tx_state = st.release();
        c->tx_state_free = free_tx_state;
    }
    return *static_cast<TxState*>(c->tx_state);
}

// carrier table from the host libm, as the reference evaluates it: cos / sin(2*M_PI*fc*(double)n*Ts), n from start_sample.
// Returns the table entry of start_sample. A request inside the cached range is served from it (a streaming caller that
// advances start_sample call by call keeps hitting a table built once with head-room); the table is bounded.
const double* ensure_carrier_table(TxState& st, double carrier_hz, uint64_t start_sample, size_t count, hipStream_t s) {
    need(count <= (size_t(1) << 27), "carrier table too long: split the batch (at most 2^27 samples of carrier phase per call)");
    if (st.cs_carrier == carrier_hz && start_sample >= st.cs_start && start_sample + count <= st.cs_start + st.cs_count)
        return st.d_cs + 2 * (start_sample - st.cs_start);
    const size_t build = std::max(count, std::min<size_t>(size_t(1) << 22, 8 * count));     // head-room for the calls that follow on
    std::vector<double> cs(2 * build);
    const double Ts = 1.0 / kSampleRate;
    for (size_t n = 0; n < build; ++n) {
        const unsigned long k = static_cast<unsigned long>(start_sample + n);
        // the reference's build evaluates cos and sin of this one phase as a single sincos() call (the compiler
        // merges them); glibc's sincos is not bit-for-bit its cos + sin, so the same call is made here
        ::sincos(2 * M_PI * carrier_hz * double(k) * Ts, &cs[2 * n + 1], &cs[2 * n]);
    }
    HIPCK(hipStreamSynchronize(s));                      // nothing in flight still reads the old table
    if (st.cs_cap < cs.size()) {
        (void)hipFree(st.d_cs);
        st.d_cs = nullptr; st.cs_cap = 0; st.cs_count = 0;
        HIPCK(hipMalloc(reinterpret_cast<void**>(&st.d_cs), cs.size() * 8));
        st.cs_cap = cs.size();
    }
    HIPCK(hipMemcpy(st.d_cs, cs.data(), cs.size() * 8, hipMemcpyHostToDevice));
    st.cs_carrier = carrier_hz; st.cs_start = start_sample; st.cs_count = build;
    return st.d_cs;
}

void transmit_dev(mgpu_ctx* c, const uint8_t* d_payload, int payload_stride, const int* d_nbytes, int F, const mgpu_transmit_config& cfg,
                  double* d_out, hipStream_t s) {
    const auto& t = c->tab;
    const int interp = 4, npre = t.preamble * t.Nofdm, ndata = t.active_nsymb * t.Nofdm, total = t.Nofdm * (t.Nsymb + t.preamble) * interp;
    const int used = (npre + ndata) * interp;
    TxState& st = tx_state(c, s);
    const bool batch = cfg.message_location == MGPU_BATCH_MESSAGE;
    // FIRST / MIDDLE / FLUSH_MESSAGE (telecom_system.cc:559-590): call n filters the span that starts half a frame into the 3-frame
    // buffer [B0, B1 (FIRST: x_n), x_n] and returns its middle frame; the buffer then moves on one frame. Over F consecutive calls
    // the spans are windows of ONE concatenation cat = [B0, B1 or x_0, x_0, x_1, ..., x_(F-1)] (frame n, n+1, n+2 for call n), and
    // an output sample sits at least T/2 - 96 samples inside its window, beyond the reach of the two 97-tap filters, so every
    // sum has exactly the terms, in the order, it has in the reference's per-call filtering: out_n = frame n+1 of FIR2(FIR1(cat)).
    const bool stream_mode = cfg.message_location >= MGPU_FIRST_MESSAGE && cfg.message_location <= MGPU_FLUSH_MESSAGE;
    const bool filtered = cfg.message_location == MGPU_SINGLE_MESSAGE || batch || stream_mode;
    const bool continuous = cfg.phase_continuous || batch;
    const double* const d_cs = ensure_carrier_table(st, cfg.carrier_hz, cfg.start_sample, continuous ? size_t(used) * F : size_t(used), s);
    if (filtered && st.carrier != cfg.carrier_hz) {
        HIPCK(hipStreamSynchronize(s));
        for (int w = 0; w < 2; ++w) {
            const std::vector<double> taps = mgpu::design_tx_fir(w, cfg.carrier_hz);
            (void)hipFree(st.d_fir[w]);
            st.d_fir[w] = nullptr;
            HIPCK(hipMalloc(reinterpret_cast<void**>(&st.d_fir[w]), taps.size() * 8));
            HIPCK(hipMemcpy(st.d_fir[w], taps.data(), taps.size() * 8, hipMemcpyHostToDevice));
            st.ntaps[w] = int(taps.size());
            need(st.ntaps[w] == FIR97_NT, "transmit filter design changed: the filter kernel is built for 97 taps");
        }
        st.carrier = cfg.carrier_hz;
    }
    // scaling constants in the reference's types and order (telecom_system.cc:388, :506-527)
    const float power_normalization = std::sqrt(double(t.Nfft * interp));
    const double mfsk_boost = t.mfsk_M > 0 ? std::sqrt(double(t.Nc) / t.mfsk_nstreams) * std::pow(10.0, -2.0 / 20.0) : 1.0;
    const double pw = std::sqrt(cfg.output_power_watt), preamble_boost = std::sqrt(2.0);          // telecom_system.cc:2840
    const double m_pre = pw * preamble_boost * mfsk_boost, m_data = pw * mfsk_boost;
    const double pow_pre = std::pow(10, cfg.preamble_papr_cut / 10.0), pow_data = std::pow(10, cfg.data_papr_cut / 10.0);

    double* const bb = static_cast<double*>(st.work(0, size_t(F) * t.frame_samples * 16, s));
    // batch form: one padding frame in front of and behind the F frames (arq_common.cc:2236-2240)
    const size_t pad = batch || stream_mode ? size_t(total) : 0;
    double* const t0 = filtered ? static_cast<double*>(st.work(1, (size_t(F) * total + 2 * pad) * 8, s)) : nullptr;
    double* const t1_all = filtered ? static_cast<double*>(st.work(2, (size_t(F) * total + 2 * pad) * 8, s)) : nullptr;
    double* clipped = filtered ? t0 + (stream_mode ? 2 * pad : pad) : d_out;     // stream form: two history frames in front
    for (int off = 0; off < F; off += kMaxFramesPerLaunch) {
        const int n = std::min(F - off, kMaxFramesPerLaunch);
        hipLaunchKernelGGL(mgpu_txgen_kernel, dim3(n), dim3(256), c->lds_tx, s, c->dev, uint64_t(0), uint64_t(0), n, 0.0, -1,
                           bb + size_t(off) * t.frame_samples * 2, static_cast<uint8_t*>(nullptr),
                           d_payload + size_t(off) * size_t(payload_stride < 0 ? -payload_stride : payload_stride), payload_stride, at(d_nbytes, size_t(off)), 0, 0);
        HIPCK(hipGetLastError());
    }
    const int kMaxY = 32768;                                   // gridDim.y limit is 65535
    for (int off = 0; off < F; off += kMaxY) {
        const int n = std::min(F - off, kMaxY);
        double* o = clipped + size_t(off) * total;
        hipLaunchKernelGGL(mgpu_tx_mix_kernel, dim3((total + 255) / 256, n), dim3(256), 0, s, st.d_pre_bb, npre,
                           bb + size_t(off) * t.frame_samples * 2, t.frame_samples, ndata, double(power_normalization), m_pre, m_data,
                           cfg.carrier_amplitude, d_cs + (continuous ? 2 * size_t(used) * off : 0), continuous ? used : 0,
                           o, total);
        HIPCK(hipGetLastError());
        hipLaunchKernelGGL(mgpu_peak_clip_kernel, dim3(n, 2), dim3(256), 0, s, o, total, npre * interp, used, pow_pre, pow_data);
        HIPCK(hipGetLastError());
        if (filtered && !batch && !stream_mode) {
            double* t1 = t1_all + size_t(off) * total;
            for (int w = 0; w < 2; ++w) {
                double* dst = w ? d_out + size_t(off) * total : t1;
                hipLaunchKernelGGL(mgpu_fir97_kernel, dim3((total + FIR97_OUT - 1) / FIR97_OUT, n), dim3(256), 0, s, w ? t1 : o, total, st.d_fir[w], dst, 0, total);
                HIPCK(hipGetLastError());
            }
        }
    }
    if (batch || stream_mode) {
        const size_t ncat = size_t(F + 2) * total;
        if (batch) {    // arq_common.cc:2236-2248: pad with the first / last frame, filter the concatenation, keep the middle
            HIPCK(hipMemcpyAsync(t0, t0 + total, size_t(total) * 8, hipMemcpyDeviceToDevice, s));
            HIPCK(hipMemcpyAsync(t0 + size_t(F + 1) * total, t0 + size_t(F) * total, size_t(total) * 8, hipMemcpyDeviceToDevice, s));
        } else {        // history in front: B0, then B1 (FIRST_MESSAGE: the first frame itself, :559-566)
            HIPCK(hipMemcpyAsync(t0, st.d_tx_buffer, size_t(total) * 8, hipMemcpyDeviceToDevice, s));
            HIPCK(hipMemcpyAsync(t0 + total, cfg.message_location == MGPU_FIRST_MESSAGE ? t0 + 2 * size_t(total) : st.d_tx_buffer + total, size_t(total) * 8,
                                 hipMemcpyDeviceToDevice, s));
            // the buffer after the last call's shift_left (:584): [frame F, frame F+1, frame F+1] of the concatenation
            HIPCK(hipMemcpyAsync(st.d_tx_buffer, t0 + size_t(F) * total, 2 * size_t(total) * 8, hipMemcpyDeviceToDevice, s));
            HIPCK(hipMemcpyAsync(st.d_tx_buffer + 2 * size_t(total), t0 + size_t(F + 1) * total, size_t(total) * 8, hipMemcpyDeviceToDevice, s));
        }
        hipLaunchKernelGGL(mgpu_fir97_kernel, dim3(unsigned((ncat + FIR97_OUT - 1) / FIR97_OUT), 1), dim3(256), 0, s, t0, int(ncat), st.d_fir[0], t1_all, 0,
                           int(ncat));
        HIPCK(hipGetLastError());
        const size_t nout = size_t(F) * total;
        hipLaunchKernelGGL(mgpu_fir97_kernel, dim3(unsigned((nout + FIR97_OUT - 1) / FIR97_OUT), 1), dim3(256), 0, s, t1_all, int(ncat), st.d_fir[1], d_out,
                           total, int(nout));
        HIPCK(hipGetLastError());
    }
}

void check_config(const mgpu_ctx* c, const mgpu_transmit_config* cfg, int payload_stride, int F) {
    need(cfg != nullptr && F >= 0, "bad argument");
    const bool stream_mode = cfg->message_location >= MGPU_FIRST_MESSAGE && cfg->message_location <= MGPU_FLUSH_MESSAGE;
    need(cfg->message_location == MGPU_SINGLE_MESSAGE || cfg->message_location == MGPU_NO_FILTER_MESSAGE || cfg->message_location == MGPU_BATCH_MESSAGE ||
             stream_mode, "message_location must be MGPU_FIRST/MIDDLE/FLUSH/SINGLE/NO_FILTER/BATCH_MESSAGE");
    need(!(cfg->message_location == MGPU_BATCH_MESSAGE || stream_mode) || (size_t(F) + 2) * size_t(mgpu_transmit_frame_samples(const_cast<mgpu_ctx*>(c))) < (size_t(1) << 31),
         "batch too long for one filtering pass (2^31 samples)");
    need(payload_stride >= c->tab.payload_bytes || payload_stride <= -c->tab.nReal, "payload_stride is shorter than the frame's payload");     // negative: rows of data bits (transmit_bit)
    need(cfg->carrier_hz > 0 && cfg->carrier_hz < kSampleRate / 2 && cfg->output_power_watt >= 0, "bad carrier or power");
}

}  // namespace

extern "C" {

int mgpu_host_pre_equalization_channel(int cfg, double carrier_hz, double* channel_c128) {
    if (!channel_c128) return MGPU_ERR_ARG;
    try {
        const mgpu::ModeTables m = mgpu::build_mode_tables(cfg, 0, mgpu_ldpc_blob, mgpu_ldpc_blob_size);
        const std::vector<mgpu::Cplx> h = mgpu::pre_equalization_channel(m, carrier_hz);
        std::memcpy(channel_c128, h.data(), h.size() * 16);
        return MGPU_OK;
    } catch (const std::exception&) { return MGPU_ERR_ARG; }
}

int mgpu_context_pre_equalization_channel(mgpu_ctx* c, double carrier_hz, double* channel_c128) {
    if (!c || !channel_c128) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(c->tab.mfsk_M == 0, "pre_equalization_channel: the OFDM modes only (telecom_system.cc:474-494)");
        const std::vector<mgpu::Cplx> h = mgpu::pre_equalization_channel(c->tab, carrier_hz);     // the context's own tables (explicit parameters included)
        std::memcpy(channel_c128, h.data(), h.size() * 16);
    });
}

int mgpu_set_pre_equalization_channel(mgpu_ctx* c, const double* channel_c128) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(c->tab.mfsk_M == 0 || !channel_c128, "pre_equalization_channel: the OFDM modes only (telecom_system.cc:474-494)");
        // No transmit call of this context may still be reading the old table or the preamble baseband made from it: such calls can run on
        // a caller's stream (mgpu_transmit_byte_batch_dev, SINGLE / NO_FILTER / BATCH), which this context cannot name, so the whole device
        // is drained — the table changes when a mode is loaded, not per frame.
        HIPCK(hipDeviceSynchronize());
        if (channel_c128) {
            c->pre_eq.assign(channel_c128, channel_c128 + 2 * size_t(c->tab.Nc));
            if (!c->d_pre_eq_buf) c->d_pre_eq_buf = c->keep(upload(c->pre_eq));
            else HIPCK(hipMemcpy(c->d_pre_eq_buf, c->pre_eq.data(), c->pre_eq.size() * 8, hipMemcpyHostToDevice));
            c->dev.pre_eq = c->d_pre_eq_buf;
        } else {
            c->pre_eq.clear();
            c->dev.pre_eq = nullptr;                              // the device buffer stays with the context for the next table
        }
        ++c->pre_eq_version;
    });
}

int mgpu_transmit_frame_samples(mgpu_ctx* c) {
    return c ? c->tab.Nofdm * (c->tab.Nsymb + c->tab.preamble) * 4 : MGPU_ERR_ARG;
}

int mgpu_transmit_byte_batch_dev(mgpu_ctx* c, const void* d_payload, int payload_stride, const void* d_nbytes, int F,
                                 const mgpu_transmit_config* cfg, void* d_passband, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        check_config(c, cfg, payload_stride, F);
        need(d_payload && d_passband, "bad argument");
        // the FIRST / MIDDLE / FLUSH messages of a stream share the context's three-frame history: calls on different streams would
        // race on it, so a stream call of the overlap-save kind must run on the context's own stream (stream == NULL)
        const bool stream_mode = cfg->message_location == MGPU_FIRST_MESSAGE || cfg->message_location == MGPU_MIDDLE_MESSAGE ||
                                 cfg->message_location == MGPU_FLUSH_MESSAGE;       // SINGLE, NO_FILTER and BATCH keep nothing between calls
        need(!stream || !stream_mode, "FIRST / MIDDLE / FLUSH messages carry state between calls: pass stream = NULL (the context's stream)");
        if (F == 0) return;
        transmit_dev(c, static_cast<const uint8_t*>(d_payload), payload_stride, static_cast<const int*>(d_nbytes), F, *cfg,
                     static_cast<double*>(d_passband), stream ? static_cast<hipStream_t>(stream) : c->stream);
        if (!stream) HIPCK(hipStreamSynchronize(c->stream));
    });
}

int mgpu_transmit_byte_batch(mgpu_ctx* c, const uint8_t* payload, int payload_stride, const int* nbytes, int F, const mgpu_transmit_config* cfg,
                             double* passband) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        check_config(c, cfg, payload_stride, F);
        need(payload && passband, "bad argument");
        if (F == 0) return;
        if (nbytes)
            for (int f = 0; f < F; ++f) need(nbytes[f] >= 0 && nbytes[f] <= c->tab.payload_bytes, "message too long");
        const size_t total = size_t(mgpu_transmit_frame_samples(c));
        DevBuf d_pl(size_t(F) * payload_stride), d_nb(size_t(F) * 4), d_out(size_t(F) * total * 8);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_pl.p, payload, size_t(F) * payload_stride, hipMemcpyHostToDevice, s));
        if (nbytes) HIPCK(hipMemcpyAsync(d_nb.p, nbytes, size_t(F) * 4, hipMemcpyHostToDevice, s));
        transmit_dev(c, d_pl.as<uint8_t>(), payload_stride, nbytes ? d_nb.as<int>() : nullptr, F, *cfg, d_out.as<double>(), s);
        HIPCK(hipMemcpyAsync(passband, d_out.p, size_t(F) * total * 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_transmit_bit_batch(mgpu_ctx* c, const uint8_t* bits, int F, const mgpu_transmit_config* cfg, double* passband) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        const int nReal = c->tab.nReal;
        check_config(c, cfg, -nReal, F);
        need(bits && passband, "bad argument");
        if (F == 0) return;
        const size_t total = size_t(mgpu_transmit_frame_samples(c));
        DevBuf d_bits(size_t(F) * nReal), d_out(size_t(F) * total * 8);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_bits.p, bits, size_t(F) * nReal, hipMemcpyHostToDevice, s));
        transmit_dev(c, d_bits.as<uint8_t>(), -nReal, nullptr, F, *cfg, d_out.as<double>(), s);      // negative stride: rows of data bits
        HIPCK(hipMemcpyAsync(passband, d_out.p, size_t(F) * total * 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_transmit_buffer(mgpu_ctx* c, double* buffer, int set) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(buffer != nullptr, "bad argument");
        hipStream_t s = c->stream;
        TxState& st = tx_state(c, s);
        const size_t bytes = 3 * size_t(mgpu_transmit_frame_samples(c)) * 8;
        if (set) HIPCK(hipMemcpyAsync(st.d_tx_buffer, buffer, bytes, hipMemcpyHostToDevice, s));
        else HIPCK(hipMemcpyAsync(buffer, st.d_tx_buffer, bytes, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_generate_ack_pattern_passband(mgpu_ctx* c, int pattern, const mgpu_transmit_config* cfg, double* out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(cfg && out && (pattern == 1 || pattern == 2), "bad argument (pattern: 1 = ACK, 2 = BREAK)");
        need(cfg->carrier_hz > 0 && cfg->carrier_hz < kSampleRate / 2 && cfg->output_power_watt >= 0, "bad carrier or power");
        const auto& t = c->tab;
        const int interp = 4, nbb = kAckNsymb * t.Nofdm, total = nbb * interp;
        hipStream_t s = c->stream;
        TxState& st = tx_state(c, s);
        // cl_mfsk::generate_ack_pattern / generate_break_pattern (mfsk.cc:195-230): one tone per symbol, hopping
        std::vector<mgpu::Cplx> car(size_t(kAckNsymb) * t.Nc, mgpu::Cplx{0.0, 0.0});
        const int* tones = pattern == 2 ? kBreakTones : kAckTones;
        const double amp = std::sqrt(double(t.Nc) / 1);
        for (int p = 0; p < kAckNsymb; ++p) car[size_t(p) * t.Nc + kAckOffset + (tones[p % kAckLen] + p * kAckHop) % kAckM] = mgpu::Cplx{amp, 0.0};
        DevBuf d_car(car.size() * 16), d_bb(size_t(nbb) * 16), d_out(size_t(total) * 8);
        HIPCK(hipMemcpyAsync(d_car.p, car.data(), car.size() * 16, hipMemcpyHostToDevice, s));
        launch_symbol_mod(c, d_car.as<double>(), kAckNsymb, d_bb.as<double>(), s);
        const double* const d_cs = ensure_carrier_table(st, cfg->carrier_hz, cfg->start_sample, size_t(total), s);
        const float power_normalization = std::sqrt(double(t.Nfft * interp));                       // telecom_system.cc:1595
        const double ack_boost = std::sqrt(double(t.Nc) / 1) * std::pow(10.0, -2.0 / 20.0);         // :1611
        const double m = std::sqrt(cfg->output_power_watt) * ack_boost;
        hipLaunchKernelGGL(mgpu_tx_mix_kernel, dim3((total + 255) / 256, 1), dim3(256), 0, s, d_bb.as<double>(), nbb, static_cast<const double*>(nullptr), 0,
                           0, double(power_normalization), m, 0.0, cfg->carrier_amplitude, d_cs, 0, d_out.as<double>(), total);
        HIPCK(hipGetLastError());
        const double p10 = std::pow(10, cfg->data_papr_cut / 10.0);
        hipLaunchKernelGGL(mgpu_peak_clip_kernel, dim3(1, 1), dim3(256), 0, s, d_out.as<double>(), total, total, total, p10, p10);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(out, d_out.p, size_t(total) * 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_symbol_mod(mgpu_ctx* c, const double* carriers, int n, double* out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(carriers && out && n >= 0, "bad argument");
        if (n == 0) return;
        DevBuf d_in(size_t(n) * c->tab.Nc * 16), d_out(size_t(n) * c->tab.Nofdm * 16);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_in.p, carriers, size_t(n) * c->tab.Nc * 16, hipMemcpyHostToDevice, s));
        launch_symbol_mod(c, d_in.as<double>(), n, d_out.as<double>(), s);
        HIPCK(hipMemcpyAsync(out, d_out.p, size_t(n) * c->tab.Nofdm * 16, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

}  // extern "C"
