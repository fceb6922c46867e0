// cl_telecom_system::receive_byte as a whole (telecom_system.cc:646-1503), batched over W independent capture
// windows: passband samples in, payload + receive_stats out. The DSP runs in the library's kernels
// (passband_to_baseband, Schmidl-Cox / MFSK time sync, energy gates, decimation, Moose, the RX hot path); the
// control flow — bounds / energy / metric gates, silence-skip and SKIP-H recoveries, the multi-trial retry loop with
// its k-th-best-peak and last-good fallbacks — is host logic that advances all windows of the batch in lock-step
// rounds, each round one batched kernel call per DSP step over the windows that still need it.
//
// Parity: every block is checked against the oracle / the compiled reference on its own; the orchestration is checked window for window
// against oracle/mercury_oracle.c:morc_receive_byte and, since round 4, against the reference's own cl_telecom_system::receive_byte
// (telecom_system.cc compiled unmodified: oracle/ref_ts_harness.cc; tests/test_receive_byte_vs_reference.py pins the oracle,
// tests/test_receive_byte.py::test_gpu_receive_byte_equals_the_reference_cl_telecom_system the GPU) — see DESIGN.md section 7.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "ctx.hpp"
#include "../../include/mercury_tx.h"

namespace {

// MERCURY_RXLOOP_TIMING=1: wall-clock per phase on stderr (the stream is synchronised at every mark)
struct PhaseTimer {
    bool on = std::getenv("MERCURY_RXLOOP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(hipStream_t s, const char* what) {
        if (!on) return;
        (void)hipStreamSynchronize(s);
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[rxloop] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

constexpr int kInterp = 4, kCoarseStep = 100;
constexpr int kManySpans = 4096;          // span energies: the lane-per-span kernel from this many spans per launch
constexpr double kEnergyGate = 0.001, kMetricGate = 0.5, kMeanHGate = 0.3, kFreqIgnore = 0.1;   // telecom_system.cc:843, :854, :1269; physical_config.cc:60

struct Win {                       // one capture window's walk through receive_byte
    int delay = 0, pream = 1, sync_trials = 0, skip_h = 0;
    double metric = 0.0, freq = 0.0, coarse_freq_offset = 0.0;
    bool in_loop = false, recovery_attempted = false, decoded = false;
    bool use_last_delay = false, use_last_freq = false;   // decided per trial
};

// Device workspace for W windows; allocated on the first call and kept in the context (hipMalloc / hipFree of several GB
// per call cost more than the kernels)
struct Workspace {
    DevBuf d_pass, d_bbi, d_frames, d_carrier, d_ia, d_ib, d_ic, d_vals, d_sum, d_cnt, d_freq, d_meanh, d_stats_k, d_payload_k, d_snr_k;
    size_t vals_per_window;
    double* h_vals = nullptr;        // page-locked landing area for the synchroniser metrics (tens of MB per call)
    // page-locked arena for the small index / result arrays of the control rounds: a copy from or to pageable memory holds the calling
    // thread for ~20-40 us each while the runtime stages it; from here the copies are queued back to back and the host moves on.
    // Bump-allocated; reset whenever the stream has been synchronised (everything queued before has completed by then).
    char* h_pin = nullptr;
    size_t pin_cap = 0, pin_off = 0;
    struct PendingDown { void* dst; const void* src; size_t bytes; };
    std::vector<PendingDown> pending;
    void* pin_take(size_t bytes) {
        const size_t need = (bytes + 63) & ~size_t(63);
        if (pin_off + need > pin_cap) return nullptr;
        void* p = h_pin + pin_off;
        pin_off += need;
        return p;
    }
    hipStream_t side = nullptr;      // the signal-strength sum (a 92 k-term dependent chain per window) runs beside the synchroniser
    hipStream_t copy = nullptr;      // brings the capture windows in, slice by slice, under the first kernels
    hipStream_t search = nullptr;    // the coarse search of a group of slices, beside the mixer / filter of the next ones
    std::vector<hipEvent_t> group_ev, we_ev;
    hipEvent_t ev_search = nullptr;
    std::vector<hipEvent_t> slice_ev;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    ~Workspace() {
        if (h_vals) (void)hipHostFree(h_vals);
        if (h_pin) (void)hipHostFree(h_pin);
        if (side) (void)hipStreamDestroy(side);
        if (copy) (void)hipStreamDestroy(copy);
        if (search) (void)hipStreamDestroy(search);
        for (hipEvent_t e : group_ev) (void)hipEventDestroy(e);
        for (hipEvent_t e : we_ev) (void)hipEventDestroy(e);
        if (ev_search) (void)hipEventDestroy(ev_search);
        for (hipEvent_t e : slice_ev) (void)hipEventDestroy(e);
        if (ev_ready) (void)hipEventDestroy(ev_ready);
        if (ev_done) (void)hipEventDestroy(ev_done);
    }
    Workspace(int W, int buf, int frame_n, size_t vals_per_window, int payload_stride, int numa_node)
        : d_pass(size_t(W) * buf * 8), d_bbi(size_t(W) * buf * 16), d_frames(size_t(W) * frame_n * 16), d_carrier(size_t(W) * 8),
          d_ia(size_t(W) * 128 * 4), d_ib(size_t(W) * 128 * 4), d_ic(size_t(W) * 4), d_vals(size_t(W) * vals_per_window * 8),
          d_sum(size_t(W) * 128 * 8), d_cnt(size_t(W) * 128 * 4), d_freq(size_t(W) * 16), d_meanh(size_t(W) * 8),
          d_stats_k(size_t(W) * sizeof(MgpuStatsDev)), d_payload_k(size_t(W) * payload_stride), d_snr_k(size_t(W) * 8), vals_per_window(vals_per_window) {
        HIPCK(host_alloc_on_node(reinterpret_cast<void**>(&h_vals), size_t(W) * vals_per_window * 8, numa_node));      // control rounds' results: on the GPU's NUMA node
        pin_cap = std::max<size_t>(size_t(1) << 20, size_t(W) * 128 * 8 * 6);             // a few rounds of the largest index / result arrays
        HIPCK(host_alloc_on_node(reinterpret_cast<void**>(&h_pin), pin_cap, numa_node));
        HIPCK(hipStreamCreate(&side));
        HIPCK(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
        HIPCK(hipEventCreateWithFlags(&ev_search, hipEventDisableTiming));
    }
};
void free_workspace(void* p) { delete static_cast<Workspace*>(p); }

struct Loop {
    mgpu_ctx* c;
    const mgpu::ModeTables& t;
    int W, buf, sym, pre, frame_i, frame_n, lower, upper, ngi_i, nfft_i, L;
    bool mfsk;
    hipStream_t s;
    mgpu_receive_config rc;
    Workspace& ws;
    DevBuf &d_pass, &d_bbi, &d_frames, &d_carrier, &d_ia, &d_ib, &d_ic, &d_vals, &d_sum, &d_cnt, &d_freq, &d_meanh;
    std::vector<double> carrier;   // per window, as currently applied (carrier + fine offset of the running trial)
    const double* pass = nullptr;  // the capture windows on the device: the workspace copy, or the caller's buffer when it already lies in HBM

    static Workspace& workspace(mgpu_ctx* ctx, int W, int buffer_nsymb) {
        const auto& t = ctx->tab;
        if (!ctx->rxloop_ws || ctx->rxloop_ws_windows < W) {
            if (ctx->rxloop_ws) free_workspace(ctx->rxloop_ws);
            ctx->rxloop_ws = nullptr;
            const int buf = t.Nofdm * buffer_nsymb * kInterp, sym = t.Nofdm * kInterp;
            const size_t vals = size_t(std::max(buf / kCoarseStep + 2, 4 * sym + 2));       // coarse / fine candidate counts
            ctx->rxloop_ws = new Workspace(W, buf, t.Nofdm * (t.Nsymb + t.preamble), vals, t.payload_stride, ctx->numa_node);
            ctx->rxloop_ws_windows = W;
            ctx->rxloop_ws_free = free_workspace;
        }
        return *static_cast<Workspace*>(ctx->rxloop_ws);
    }

    Loop(mgpu_ctx* ctx, int W_, const mgpu_receive_config& rc_, int buffer_nsymb)
        : c(ctx), t(ctx->tab), W(W_), buf(t.Nofdm * buffer_nsymb * kInterp), sym(t.Nofdm * kInterp), pre(t.preamble),
          frame_i(t.Nofdm * (t.Nsymb + t.preamble) * kInterp), frame_n(t.Nofdm * (t.Nsymb + t.preamble)), lower(t.preamble),
          upper(buffer_nsymb - (t.Nsymb + t.preamble)), ngi_i(t.Ngi * kInterp), nfft_i(t.Nfft * kInterp), L(t.preamble * t.Nofdm * kInterp),
          mfsk(t.mfsk_M > 0), s(ctx->stream), rc(rc_), ws(workspace(ctx, W_, buffer_nsymb)),
          d_pass(ws.d_pass), d_bbi(ws.d_bbi), d_frames(ws.d_frames), d_carrier(ws.d_carrier), d_ia(ws.d_ia), d_ib(ws.d_ib), d_ic(ws.d_ic),
          d_vals(ws.d_vals), d_sum(ws.d_sum), d_cnt(ws.d_cnt), d_freq(ws.d_freq), d_meanh(ws.d_meanh), carrier(W_, rc_.carrier_hz) {
        // a call that threw between down_async() and settle() leaves entries whose destinations were its stack vectors: never replay them
        ws.pending.clear();
        ws.pin_off = 0;
        d_ia.view = d_ib.view = d_ic.view = d_carrier.view = nullptr;
    }
    // Nothing of a call may outlive it: when the call ends - by return or by an exception unwinding it - with slots of the staging ring handed
    // out and no settle() since, the stream is drained before the ring can be reset by the next call (the error path skipped the wait).
    ~Loop() { if (unsettled) (void)hipStreamSynchronize(s); }
    const bool zero_copy = []{ const char* e = std::getenv("MERCURY_RB_ZEROCOPY"); return !e || atoi(e) != 0; }();

    bool in_bounds(int p) const { return p > lower && p < upper; }

    // Host -> device for the control rounds' small index arrays (window lists, start offsets, carriers: a few KB). They are placed in the
    // page-locked staging area and the kernels read them THERE (the buffer's view): a 4 KB hipMemcpyAsync is a 5 us blit kernel plus the
    // gaps around it on the stream, and a call of 1024 windows made 400 of them (a fifth of its device time). A staging slot is reused
    // only after settle() has waited for the stream, so every kernel launched with a view has read it by then.
    // INVARIANT: a buffer with a view may only be consumed by kernels launched on stream `s` — settle() waits for `s` alone, so a kernel on
    // the side stream reading a view could still be running when its slot is handed out again. The side stream's kernels (signal level,
    // upload slices) take device buffers only; keep it that way or give them their own settle().
    // true from the first staging slot handed out / copy queued until settle() has waited for the stream: while it is set, kernels in flight may
    // still read this call's slots of the staging ring, and the call must not return (receive_byte_impl's Drain).
    bool unsettled = false;
    void up(DevBuf& d, const void* h, size_t bytes) {
        d.view = nullptr;
        unsettled = true;
        if (void* p = ws.pin_take(bytes)) {
            std::memcpy(p, h, bytes);
            if (zero_copy) { d.view = p; return; }
            h = p;
        }
        HIPCK(hipMemcpyAsync(d.p, h, bytes, hipMemcpyHostToDevice, s));
    }
    // device -> host without waiting: the bytes are in h after the next down() / settle()
    void down_async(void* h, const void* d_src, size_t bytes) {
        unsettled = true;
        if (void* p = ws.pin_take(bytes)) {
            HIPCK(hipMemcpyAsync(p, d_src, bytes, hipMemcpyDeviceToHost, s));
            ws.pending.push_back({h, p, bytes});
        } else {
            HIPCK(hipMemcpyAsync(h, d_src, bytes, hipMemcpyDeviceToHost, s));
        }
    }
    void down_async(void* h, DevBuf& d, size_t bytes) { down_async(h, static_cast<const void*>(d.p), bytes); }
    void settle() {
        HIPCK(hipStreamSynchronize(s));
        for (const auto& q : ws.pending) std::memcpy(q.dst, q.src, q.bytes);
        ws.pending.clear();
        ws.pin_off = 0;
        unsettled = false;
    }
    void down(void* h, DevBuf& d, size_t bytes) { down_async(h, d, bytes); settle(); }

    // passband_to_baseband of the whole buffer for the listed windows, overwriting their interpolated baseband
    void p2b(const std::vector<int>& wins, int filter) {
        if (wins.empty()) return;
        up(d_ia, wins.data(), wins.size() * 4);
        up(d_carrier, carrier.data(), size_t(W) * 8);
        const auto& taps = filter ? t.fir_data : t.fir_time_sync;
        const int ntaps = int(taps.size());
        // windows that all mix with the call's own carrier (always so before a frequency offset has been measured) use the host-libm
        // table: the reference's own cos / sin values, and no trigonometry on the device. A window re-mixed at its measured offset
        // has a carrier of its own — never the table's, which would have to be rebuilt (3 ms of host time) for every such launch.
        bool shared = true;
        for (int w : wins) shared = shared && carrier[w] == rc.carrier_hz;
        const double* cs = shared ? mixer_table(c, carrier[wins[0]], size_t(buf), s) : nullptr;
        launch_p2b(pass, buf, d_carrier.as<double>(), nullptr, 0, buf, 1, c->d_fir[filter], ntaps, d_bbi.as<double>(), d_ia.as<int>(), cs, nullptr, 0,
                   int(wins.size()), s);
    }

    // FIR_rx_data baseband of the frame at `delay` only, decimated, straight into d_frames (row = slot[j] or j): passband_to_baseband
    // (ofdm.cc:2316-2339) followed by rational_resampler(DECIMATION) at the delay (:2267-2278) keeps every kInterp-th sample of the
    // (preamble + Nsymb) * Nofdm * kInterp the frame spans — a tenth of the capture window; each kept sample is the same 33-term sum.
    void p2b_frames(const std::vector<int>& wins, const std::vector<int>& delay, const int* slot) {
        if (wins.empty()) return;
        std::vector<int> st(W, 0);
        for (size_t j = 0; j < wins.size(); ++j) st[wins[j]] = delay[j];
        up(d_ia, wins.data(), wins.size() * 4);
        up(d_ib, st.data(), size_t(W) * 4);
        if (slot) up(d_ic, slot, wins.size() * 4);
        up(d_carrier, carrier.data(), size_t(W) * 8);
        const int ntaps = int(t.fir_data.size());
        bool shared = true;
        for (int w : wins) shared = shared && carrier[w] == rc.carrier_hz;
        const double* cs = shared ? mixer_table(c, carrier[wins[0]], size_t(buf), s) : nullptr;
        launch_p2b(pass, buf, d_carrier.as<double>(), d_ib.as<int>(), 0, frame_n, kInterp, c->d_fir[1], ntaps, d_frames.as<double>(), d_ia.as<int>(), cs,
                   slot ? d_ic.as<int>() : nullptr, 1, int(wins.size()), s);
    }

    // time_sync_preamble[_with_metric] on a sub-window [start, start + size) of each listed window
    void tsync(const std::vector<int>& wins, const std::vector<int>& start, const std::vector<int>& size, int step,
               const std::vector<int>& loc, int ntrials, std::vector<int>& delay, std::vector<double>& corr) {
        const int n = int(wins.size());
        delay.assign(n, 0);
        corr.assign(n, 0.0);
        if (!n) return;
        std::vector<int> nc(n);
        int ncmax = 1;
        for (int k = 0; k < n; ++k) { nc[k] = size[k] > L ? (size[k] - L + step - 1) / step : 0; ncmax = std::max(ncmax, nc[k]); }
        need(size_t(ncmax) <= ws.vals_per_window, "search window larger than the workspace");
        up(d_ia, wins.data(), size_t(n) * 4);
        up(d_ib, start.data(), size_t(n) * 4);
        up(d_ic, nc.data(), size_t(n) * 4);
        launch_tsync_metric(d_bbi.as<double>(), buf, d_ib.as<int>(), d_ia.as<int>(), d_ic.as<int>(), ncmax, n, step, pre, ngi_i, nfft_i,
                            d_vals.as<double>(), s);
        select(n, ncmax, nc, size, loc, step, ntrials, delay, corr);
    }

    // the reference's peak selection (ofdm.cc:1943-1964) over the candidate metrics of n windows lying in d_vals ([n][ncmax])
    void select(int n, int ncmax, const std::vector<int>& nc, const std::vector<int>& size, const std::vector<int>& loc, int step, int ntrials,
                std::vector<int>& delay, std::vector<double>& corr) {
        delay.assign(n, 0);
        corr.assign(n, 0.0);
        if (!n) return;
        if (n >= 32) {   // peak selection where the metrics lie; only (delay, correlation) per window come back
            up(d_ic, nc.data(), size_t(n) * 4);
            up(d_ia, size.data(), size_t(n) * 4);                    // wins / start are consumed: reuse their index buffers
            up(d_ib, loc.data(), size_t(n) * 4);
            hipLaunchKernelGGL(mgpu_select_peak_kernel, dim3(n), dim3(64), 0, s, d_vals.as<double>(), d_ic.as<int>(), ncmax, step,
                               d_ia.as<int>(), d_ib.as<int>(), ntrials, n, d_cnt.as<int>(), d_sum.as<double>());
            HIPCK(hipGetLastError());
            down_async(delay.data(), d_cnt, size_t(n) * 4);
            down(corr.data(), d_sum, size_t(n) * 8);
        } else {         // a few windows: one lane per window would crawl through its candidates; the host is quicker
            const double* vals = ws.h_vals;
            down(ws.h_vals, d_vals, size_t(n) * ncmax * 8);
            for (int k = 0; k < n; ++k) select_peak(&vals[size_t(k) * ncmax], nc[k], step, size[k], loc[k], ntrials, &delay[k], &corr[k]);
        }
    }

    // sum and count of |x|^2 over [off, off + len) (clipped at the buffer end) for every (window, offset) pair
    void energies(const std::vector<int>& wv, const std::vector<int>& off, std::vector<double>& sum, std::vector<int>& cnt, int len = 0) {
        if (len <= 0) len = sym;
        const int n = int(wv.size());
        sum.assign(n, 0.0);
        cnt.assign(n, 0);
        for (int base = 0; base < n; base += W * 128) {              // the index buffers hold W * 128 entries
            const int m = std::min(n - base, W * 128);
            up(d_ia, wv.data() + base, size_t(m) * 4);
            up(d_ib, off.data() + base, size_t(m) * 4);
            // a few spans: a wavefront each (short latency); thousands: a lane each (sync.hip)
            if (m >= kManySpans)
                hipLaunchKernelGGL(mgpu_span_energy_many_kernel, dim3((m + 255) / 256), dim3(256), 0, s, d_bbi.as<double>(), buf, d_ia.as<int>(), d_ib.as<int>(), m,
                                   len, d_sum.as<double>(), d_cnt.as<int>());
            else
                hipLaunchKernelGGL(mgpu_span_energy_kernel, dim3((m + 3) / 4), dim3(256), 0, s, d_bbi.as<double>(), buf, d_ia.as<int>(), d_ib.as<int>(), m,
                                   len, d_sum.as<double>(), d_cnt.as<int>());
            HIPCK(hipGetLastError());
            down_async(sum.data() + base, d_sum, size_t(m) * 8);
            down(cnt.data() + base, d_cnt, size_t(m) * 4);
        }
    }
    static double mean(double sum, int cnt) { return cnt > 0 ? sum / cnt : 0.0; }

    // The "scan forward for signal energy, re-run Schmidl-Cox from there" recovery shared by the bounds check
    // (telecom_system.cc:733-806), the silence skip (:862-925) and, with a fixed start and size, SKIP-H (:1436-1497)
    void recover(std::vector<Win>& win, const std::vector<int>& wins, const std::vector<int>& scan_from, bool fixed_start, bool need_metric,
                 std::vector<char>& ok) {
        const int n = int(wins.size());
        ok.assign(n, 0);
        if (!n) return;
        std::vector<int> search_start(n, -1);
        if (fixed_start) {
            for (int k = 0; k < n; ++k) if (scan_from[k] < upper) search_start[k] = scan_from[k] * sym;      // :1447-1456
        } else {
            std::vector<int> wv, off, first(n + 1, 0);
            for (int k = 0; k < n; ++k) {
                for (int q = scan_from[k]; q < upper; ++q) { wv.push_back(wins[k]); off.push_back(q * sym); }
                first[k + 1] = int(wv.size());
            }
            std::vector<double> sum;
            std::vector<int> cnt;
            energies(wv, off, sum, cnt);
            for (int k = 0; k < n; ++k)
                for (int j = first[k]; j < first[k + 1]; ++j)
                    if (mean(sum[j], cnt[j]) > kEnergyGate) { search_start[k] = off[j]; break; }
        }
        std::vector<int> sel, sw, ss, sz, loc;
        for (int k = 0; k < n; ++k) {
            if (search_start[k] < 0) continue;
            int available = buf - search_start[k];
            if (fixed_start) available = std::min(available, t.Nofdm * (2 * pre + t.Nsymb) * kInterp);
            if (available <= pre * sym) continue;
            sel.push_back(k); sw.push_back(wins[k]); ss.push_back(search_start[k]); sz.push_back(available); loc.push_back(0);
        }
        std::vector<int> d;
        std::vector<double> corr;
        tsync(sw, ss, sz, kCoarseStep, loc, 1, d, corr);
        std::vector<int> off(sel.size());
        for (size_t j = 0; j < sel.size(); ++j) { d[j] += ss[j]; off[j] = d[j]; }
        std::vector<double> sum;
        std::vector<int> cnt;
        energies(sw, off, sum, cnt);
        for (size_t j = 0; j < sel.size(); ++j) {
            int rsym = d[j] / sym;
            if (rsym < 1) rsym = 1;
            if (mean(sum[j], cnt[j]) >= kEnergyGate && (!need_metric || corr[j] >= kMetricGate) && in_bounds(rsym)) {
                Win& x = win[wins[sel[j]]];
                x.delay = d[j]; x.metric = corr[j]; x.pream = rsym;
                ok[sel[j]] = 1;
            }
        }
    }
};

}  // namespace

extern "C" {

int mgpu_receive_buffer_nsymb(mgpu_ctx* c) {
    if (!c) return -1;
    const auto& t = c->tab;                                      // data_container.cc:133-143
    const double sym_time_ms = 1000.0 * t.Nofdm * kInterp / 48000.0;
    const int turnaround_symb = int(std::ceil(1200.0 / sym_time_ms)) + 4;
    const int frame_symb = t.preamble + t.Nsymb;
    int min_buf = frame_symb * 2;
    if (frame_symb + turnaround_symb > min_buf) min_buf = frame_symb + turnaround_symb;
    if (min_buf < 32) min_buf = 32;
    return min_buf;
}

// cl_telecom_system::measure_signal_only (telecom_system.cc:1520-1541): time-sync filter + whole-window power, nothing else
int mgpu_measure_signal_only(mgpu_ctx* c, const double* passband, int W, double carrier_hz, double* signal_strength_dbm) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(passband && signal_strength_dbm && W > 0 && W <= c->max_batch, "bad argument (W must be 1..max_batch)");
        const mgpu_receive_config rc = {carrier_hz, 1, 0, 0, 0};
        Loop lp(c, W, rc, mgpu_receive_buffer_nsymb(c));
        hipStream_t s = lp.s;
        std::vector<int> all(W);
        for (int w = 0; w < W; ++w) all[w] = w;
        HIPCK(hipMemcpyAsync(lp.d_pass.p, passband, size_t(W) * lp.buf * 8, hipMemcpyDefault, s));
        lp.pass = lp.d_pass.as<double>();
        lp.p2b(all, 0);
        hipLaunchKernelGGL(mgpu_window_energy_kernel, dim3(W), dim3(64), 0, s, lp.d_bbi.as<double>(), lp.buf, lp.buf, lp.d_freq.as<double>());
        HIPCK(hipGetLastError());
        std::vector<double> sum(W);
        lp.down(sum.data(), lp.d_freq, size_t(W) * 8);
        for (int w = 0; w < W; ++w) signal_strength_dbm[w] = 10.0 * std::log10((sum[w] / lp.buf) / 0.001);     // ofdm.cc:1523-1539
    });
}

}  // extern "C"

namespace {
// receive_byte for W windows that lie in host or device memory, on the context's stream
void receive_byte_impl(mgpu_ctx* c, const double* passband, int W, const mgpu_receive_config* rcp, mgpu_link_state* state, uint8_t* payload,
                       mgpu_receive_stats* stats) {
    {
        const auto& t = c->tab;
        const int T = rcp->time_sync_trials_max;
        PhaseTimer pt;
        Loop lp(c, W, *rcp, mgpu_receive_buffer_nsymb(c));
        hipStream_t s = lp.s;
        pt.mark(s, "device buffers");
        ensure_workspaces(c, WS_FRONTEND | WS_LLR | WS_OUT);
        std::vector<Win> win(W);
        std::vector<int> all(W);
        for (int w = 0; w < W; ++w) all[w] = w;
        // receive_stats as init() leaves it (telecom_system.cc:1968-1981) + the per-call resets (:653-655)
        for (int w = 0; w < W; ++w) {
            mgpu_receive_stats& r = stats[w];
            r.iterations_done = -1; r.crc = 0; r.all_zeros = 0; r.message_decoded = 0; r.snr_db = -99.9;
            r.delay = 0; r.sync_trials = 0; r.freq_offset = 0; r.coarse_metric = 0; r.frame_overflow_symbols = 0; r.mean_H = -1.0;
            r.signal_strength_dbm = -999;
        }
        std::memset(payload, 0, size_t(W) * t.payload_stride);
        // ---- upload + :676-696 coarse synchronisation on the FIR_rx_time_sync baseband, pipelined over slices of windows ----
        // The windows may lie in host memory (the reference's capture buffer; 740 KB each in mode 8, i.e. 13 ms of PCIe time per
        // 1024) or already in HBM (hipMemcpyDefault). A copy stream brings them over slice by slice; the mixer + time-sync filter
        // and the Schmidl-Cox metric of a slice run as soon as it has landed, under the copies of the following slices (a copy
        // from pageable memory holds the host thread, but the kernels of the slices before it are already queued).
        const int kSlice = 64, kCoarseGroup = 2;
        const int nsl = (W + kSlice - 1) / kSlice;
        hipPointerAttribute_t pattr{};
        const bool on_device = hipPointerGetAttributes(&pattr, passband) == hipSuccess && pattr.type == hipMemoryTypeDevice;
        if (!on_device) (void)hipGetLastError();
        while (int(lp.ws.slice_ev.size()) < nsl) {
            hipEvent_t e = nullptr;
            HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            lp.ws.slice_ev.push_back(e);
        }
        if (!lp.ws.copy) HIPCK(hipStreamCreateWithFlags(&lp.ws.copy, hipStreamNonBlocking));
        lp.up(lp.d_ia, all.data(), size_t(W) * 4);
        lp.up(lp.d_carrier, lp.carrier.data(), size_t(W) * 8);
        const double* mix_cs = mixer_table(c, rcp->carrier_hz, size_t(lp.buf), s);        // every window mixes with the call's carrier here
        HIPCK(hipEventRecord(lp.ws.ev_ready, s));
        HIPCK(hipStreamWaitEvent(lp.ws.copy, lp.ws.ev_ready, 0));                         // the previous call is done with d_pass
        const int ntaps_ts = int(t.fir_time_sync.size());
        const int ncand0 = lp.buf > lp.L ? (lp.buf - lp.L + kCoarseStep - 1) / kCoarseStep : 0;
        need(size_t(std::max(ncand0, 1)) <= lp.ws.vals_per_window, "search window larger than the workspace");
        // Windows that already lie in HBM are read where they are. The coarse search (one wavefront per SIMD, issue-limited: sync.hip)
        // runs on a stream of its own, group by group, beside the mixer / filter launches of the following slices; from host memory a
        // group is kCoarseGroup slices (its search runs under the next group's copies), from HBM all of them.
        lp.pass = on_device ? passband : lp.d_pass.as<double>();
        if (!lp.ws.search) HIPCK(hipStreamCreateWithFlags(&lp.ws.search, hipStreamNonBlocking));
        const int group = on_device ? nsl : kCoarseGroup;           // from HBM: one launch (groups of 4 slices beside the filter launches measured 5 % slower)
        while (int(lp.ws.group_ev.size()) < nsl) {
            hipEvent_t e = nullptr;
            HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            lp.ws.group_ev.push_back(e);
        }
        while (int(lp.ws.we_ev.size()) < nsl) {
            hipEvent_t e = nullptr;
            HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            lp.ws.we_ev.push_back(e);
        }
        // the whole-window signal level (and the MFSK search) is skipped when every window comes with a known delay (the MFSK BER loop)
        bool need_level = !lp.mfsk || !state;
        if (!need_level) for (int w = 0; w < W; ++w) if (state[w].fixed_delay_plus_one <= 0) { need_level = true; break; }
        // Windows already in HBM (OFDM): the signal-level chains (one wavefront per window, 0.6 ms of dependent additions) are launched
        // behind the coarse search instead of ahead of it: they then run beside the gates, their host round trips and the recovery search,
        // where the GPU has room; beside the coarse search — one latency-bound wavefront per SIMD — they cost it 0.37 ms. (From host memory
        // the search waits for PCIe anyway; MFSK windows have the longest chain and nothing but the whole call to hide it behind.)
        const bool defer_level = need_level && on_device && !lp.mfsk;
        auto launch_level = [&](int g0, int gn, hipEvent_t ev) {
            HIPCK(hipEventRecord(ev, s));
            HIPCK(hipStreamWaitEvent(lp.ws.side, ev, 0));
            hipLaunchKernelGGL(mgpu_window_energy_kernel, dim3(gn), dim3(64), 0, lp.ws.side, lp.d_bbi.as<double>() + size_t(g0) * lp.buf * 2, lp.buf, lp.buf,
                               lp.d_freq.as<double>() + g0);
            HIPCK(hipGetLastError());
        };
        for (int k = 0; k < nsl; ++k) {
            const int off = k * kSlice, n = std::min(kSlice, W - off);
            if (!on_device) {
                HIPCK(hipMemcpyAsync(lp.d_pass.as<double>() + size_t(off) * lp.buf, passband + size_t(off) * lp.buf, size_t(n) * lp.buf * 8, hipMemcpyDefault, lp.ws.copy));
                HIPCK(hipEventRecord(lp.ws.slice_ev[k], lp.ws.copy));
                HIPCK(hipStreamWaitEvent(s, lp.ws.slice_ev[k], 0));
            }
            launch_p2b(lp.pass, lp.buf, lp.d_carrier.as<double>(), nullptr, 0, lp.buf, 1, c->d_fir[0], ntaps_ts, lp.d_bbi.as<double>(), lp.d_ia.as<int>() + off,
                       mix_cs, nullptr, 0, n, s);
            const bool group_end = (k + 1) % group == 0 || k == nsl - 1;
            if (group_end && need_level && !defer_level) {
                // :678 measure_signal_stregth (ofdm.cc:1523-1539): the whole window's |x|^2 added in sample order, a 92 k-term dependent
                // chain per window. One wavefront and 4 KB of LDS per window on a side stream, launched ahead of the group's coarse search
                // so that the two share the compute units (behind the search it added its full latency to the call).
                const int g0 = (k / group) * group * kSlice;
                launch_level(g0, off + n - g0, lp.ws.we_ev[k]);
            }
            if (!lp.mfsk && ncand0 > 0 && group_end) {
                const int g0 = (k / group) * group * kSlice, gn = off + n - g0;
                HIPCK(hipEventRecord(lp.ws.group_ev[k], s));
                HIPCK(hipStreamWaitEvent(lp.ws.search, lp.ws.group_ev[k], 0));
                launch_tsync_metric(lp.d_bbi.as<double>() + size_t(g0) * lp.buf * 2, lp.buf, nullptr, nullptr, nullptr, ncand0, gn, kCoarseStep, lp.pre, lp.ngi_i,
                                    lp.nfft_i, lp.d_vals.as<double>() + size_t(g0) * ncand0, lp.ws.search);
            }
        }
        if (!lp.mfsk && ncand0 > 0) {                                // the main stream continues when the last group's search is done
            HIPCK(hipEventRecord(lp.ws.ev_search, lp.ws.search));
            HIPCK(hipStreamWaitEvent(s, lp.ws.ev_search, 0));
        }
        pt.mark(s, "upload + p2b + coarse metric");
        if (defer_level) launch_level(0, W, lp.ws.we_ev[0]);        // behind the coarse search (the event is recorded after the main stream's wait for it)
        HIPCK(hipEventRecord(lp.ws.ev_done, lp.ws.side));            // signal strength: the main stream waits for it before the trial loop overwrites the baseband
        pt.mark(s, "signal strength");
        std::vector<char> live(W, 1);                             // still on the way to the trial loop
        std::vector<char> fixed_delay(W, 0);
        if (lp.mfsk) {
            // cl_ofdm::time_sync_mfsk (ofdm.cc:2011-2061): slot energies and the preamble-tone search, both where the baseband lies; only the
            // delays come back. Windows with a known delay (:663-672 mfsk_fixed_delay, used once, no signal level) need neither.
            std::vector<int> d(W, 0);
            if (need_level) {
                const int nslots = lp.buf / lp.sym;
                DevBuf d_e(size_t(W) * nslots * t.Nc * 8);
                HIPCK(hipMemsetAsync(d_e.p, 0, size_t(W) * nslots * t.Nc * 8, s));
                hipLaunchKernelGGL(mgpu_slot_energy_kernel, dim3((nslots + 3) / 4, W), dim3(256), 0, s, lp.d_bbi.as<double>(), lp.buf, nslots, kInterp,
                                   c->dev.twiddle, d_e.as<double>());
                HIPCK(hipGetLastError());
                std::vector<int> ss(W, 0);
                if (state) for (int w = 0; w < W; ++w) ss[w] = state[w].mfsk_search_start;
                lp.up(lp.d_ib, ss.data(), size_t(W) * 4);
                launch_mfsk_sync(c, d_e.as<double>(), W, nslots, lp.buf, lp.d_ib.as<int>(), lp.d_cnt.as<int>(), s);
                lp.down(d.data(), lp.d_cnt, size_t(W) * 4);
            }
            for (int w = 0; w < W; ++w) {
                if (state && state[w].fixed_delay_plus_one > 0) {
                    win[w].delay = state[w].fixed_delay_plus_one - 1;
                    state[w].fixed_delay_plus_one = 0;
                    fixed_delay[w] = 1;
                    continue;
                }
                win[w].delay = d[w];
            }
        } else {
            std::vector<int> zero(W, 0), full(W, lp.buf), d;
            std::vector<double> corr;
            lp.select(W, ncand0, std::vector<int>(W, ncand0), full, zero, kCoarseStep, 1, d, corr);     // the metrics are already in d_vals
            for (int w = 0; w < W; ++w) { win[w].delay = d[w]; win[w].metric = corr[w]; }
        }
        pt.mark(s, "coarse time sync");
        for (int w = 0; w < W; ++w) { win[w].pream = std::max(1, win[w].delay / lp.sym); }
        // ---- :702-718 MFSK frame completeness ----
        if (lp.mfsk)
            for (int w = 0; w < W; ++w) {
                const int frame_end = win[w].delay + (lp.pre + t.active_nsymb) * lp.sym;
                if (frame_end > lp.buf) { stats[w].frame_overflow_symbols = (frame_end - lp.buf + lp.sym - 1) / lp.sym; live[w] = 0; }
            }
        if (!lp.mfsk) {
            // ---- :733-806 preamble outside the valid bounds: scan the buffer for signal, search again from there ----
            std::vector<int> wins, from;
            for (int w = 0; w < W; ++w) if (!lp.in_bounds(win[w].pream)) { wins.push_back(w); from.push_back(lp.lower + 1); }
            std::vector<char> ok;
            lp.recover(win, wins, from, false, true, ok);
        }
        for (int w = 0; w < W; ++w) if (live[w] && !lp.in_bounds(win[w].pream)) live[w] = 0;
        if (!lp.mfsk) {
            // ---- :808-928 energy / metric gates and the silence-skip recovery ----
            std::vector<int> wv, off;
            for (int w = 0; w < W; ++w) if (live[w]) { wv.push_back(w); off.push_back(win[w].delay); }
            std::vector<double> sum;
            std::vector<int> cnt;
            pt.mark(s, "gates: bounds recovery");
            lp.energies(wv, off, sum, cnt);
            pt.mark(s, "gates: energy at delay");
            std::vector<int> wins, from;
            for (size_t j = 0; j < wv.size(); ++j) {
                bool energy_ok = !(Loop::mean(sum[j], cnt[j]) < kEnergyGate);
                if (energy_ok && win[wv[j]].metric < kMetricGate) energy_ok = false;
                if (!energy_ok) { wins.push_back(wv[j]); from.push_back(win[wv[j]].pream + 1); }
            }
            if (pt.on) std::fprintf(stderr, "[rxloop] %zu of %zu windows fail the energy / metric gate\n", wins.size(), wv.size());
            std::vector<char> ok;
            lp.recover(win, wins, from, false, true, ok);
            for (size_t j = 0; j < wins.size(); ++j) if (!ok[j]) live[wins[j]] = 0;
        }
        for (int w = 0; w < W; ++w) { win[w].in_loop = live[w] != 0; }
        pt.mark(s, "bounds / energy gates");

        {   // the signal-strength sums must be out of the baseband before the trial loop re-filters it
            HIPCK(hipStreamWaitEvent(s, lp.ws.ev_done, 0));
            std::vector<double> sum(W);
            lp.down(sum.data(), lp.d_freq, size_t(W) * 8);
            for (int w = 0; w < W; ++w) stats[w].signal_strength_dbm = fixed_delay[w] ? 0.0 : 10.0 * std::log10((sum[w] / lp.buf) / 0.001);
        }
        // ---- :931-1431 the trial loop, one round per trial over the windows still in it ----
        DevBuf& d_stats_k = lp.ws.d_stats_k;
        DevBuf& d_payload_k = lp.ws.d_payload_k;
        std::vector<MgpuStatsDev> st_k(W);
        std::vector<uint8_t> pay_k(size_t(W) * t.payload_stride);
        for (;;) {
            std::vector<int> act;
            for (int w = 0; w < W; ++w) {
                Win& x = win[w];
                if (!x.in_loop) continue;
                if (x.sync_trials > T || (lp.mfsk && x.sync_trials > 0)) { x.in_loop = false; continue; }   // :931, :939-944
                act.push_back(w);
            }
            if (act.empty()) {
                // ---- :1436-1497 SKIP-H recovery: every trial died on a low channel estimate -> look for a later preamble ----
                std::vector<int> wins, from;
                for (int w = 0; w < W; ++w) {
                    Win& x = win[w];
                    if (!lp.mfsk && live[w] && !x.decoded && x.skip_h >= T + 1 && !x.recovery_attempted) {
                        x.recovery_attempted = true;
                        wins.push_back(w); from.push_back(x.pream + 2);
                    }
                }
                if (wins.empty()) break;
                std::vector<int> searchable;
                for (size_t j = 0; j < wins.size(); ++j) {
                    const int start = from[j] * lp.sym;
                    const int available = std::min(lp.buf - start, t.Nofdm * (2 * lp.pre + t.Nsymb) * kInterp);
                    if (from[j] < lp.upper && available > lp.pre * lp.sym) { searchable.push_back(wins[j]); lp.carrier[wins[j]] = rcp->carrier_hz; }
                }
                lp.p2b(searchable, 0);                                // fresh FIR_rx_time_sync baseband for the search (:1458-1463)
                std::vector<char> ok;
                lp.recover(win, wins, from, true, false, ok);
                bool any = false;
                for (size_t j = 0; j < wins.size(); ++j)
                    if (ok[j]) { Win& x = win[wins[j]]; x.sync_trials = 0; x.skip_h = 0; x.coarse_freq_offset = 0.0; x.in_loop = true; any = true; }
                if (!any) break;
                continue;
            }
            const int n = int(act.size());
            // -- delay for this trial: last good one on the final trial (:945-948), else the k-th best fine-search peak (:1014-1018)
            std::vector<int> fw, fstart, fsize, floc, cw;
            for (int w : act) {
                Win& x = win[w];
                x.use_last_delay = !lp.mfsk && x.sync_trials == T && rcp->use_last_good_time_sync && state && state[w].delay_of_last_decoded_message != -1;
                if (lp.mfsk) continue;
                if (x.use_last_delay) { x.delay = state[w].delay_of_last_decoded_message; continue; }
                if (x.sync_trials == 1 && rcp->coarse_freq_sync_enabled) { cw.push_back(w); continue; }
                fw.push_back(w); fstart.push_back((x.pream - 1) * lp.sym); fsize.push_back((lp.pre + 4) * lp.sym); floc.push_back(x.sync_trials);
            }
            if (!cw.empty()) {   // :949-1012 coarse frequency search before trial 1: Schmidl-Cox at carrier -30 / 0 / +30 Hz
                const double freq_search[3] = {-30.0, 0.0, 30.0};
                const int nc = int(cw.size());
                std::vector<double> best_corr(nc, 0.0), best_off(nc, 0.0), zero_corr(nc, 0.0);
                std::vector<int> best_delay(nc), zero(nc, 0), ssize(nc, t.Nofdm * (2 * lp.pre + t.Nsymb) * kInterp);
                for (int j = 0; j < nc; ++j) best_delay[j] = win[cw[j]].delay;
                for (int i = 0; i < 3; ++i) {
                    for (int w : cw) lp.carrier[w] = rcp->carrier_hz + freq_search[i];
                    lp.p2b(cw, 0);
                    std::vector<int> d;
                    std::vector<double> corr;
                    lp.tsync(cw, zero, ssize, kCoarseStep, zero, 1, d, corr);
                    for (int j = 0; j < nc; ++j) {
                        if (std::fabs(freq_search[i]) < 0.1) zero_corr[j] = corr[j];
                        if (corr[j] > best_corr[j]) { best_corr[j] = corr[j]; best_off[j] = freq_search[i]; best_delay[j] = d[j]; }
                    }
                }
                for (int j = 0; j < nc; ++j) {
                    Win& x = win[cw[j]];
                    if (std::fabs(best_off[j]) > 1.0 && best_corr[j] > 0.5 && best_corr[j] > zero_corr[j] + 0.1) {
                        x.coarse_freq_offset = best_off[j];
                        x.delay = best_delay[j];
                        x.pream = std::max(1, x.delay / lp.sym);
                    }
                    lp.carrier[cw[j]] = rcp->carrier_hz + x.coarse_freq_offset;
                }
                lp.p2b(cw, 0);                                       // time-sync baseband at the (possibly corrected) carrier
                for (int w : cw) { fw.push_back(w); fstart.push_back((win[w].pream - 1) * lp.sym); fsize.push_back((lp.pre + 4) * lp.sym); floc.push_back(win[w].sync_trials); }
            }
            {
                std::vector<int> d;
                std::vector<double> corr;
                lp.tsync(fw, fstart, fsize, 1, floc, T, d, corr);
                for (size_t j = 0; j < fw.size(); ++j) win[fw[j]].delay = fstart[j] + d[j];
            }
            pt.mark(s, "trial: fine time sync");
            for (int w : act) {                                      // :1020-1031
                Win& x = win[w];
                if (x.delay < 0) x.delay = 0;
                if (x.delay > lp.buf - lp.frame_i) x.delay = lp.buf - lp.frame_i;
            }
            if (!lp.mfsk) {                                          // :1039-1071 post-fine-sync energy fix
                std::vector<int> wv, off;
                for (int w : act) for (int q = 0; q <= 3; ++q) { wv.push_back(w); off.push_back(std::min(win[w].delay + q * lp.sym, lp.buf)); }
                std::vector<double> sum;
                std::vector<int> cnt;
                lp.energies(wv, off, sum, cnt);
                for (int k = 0; k < n; ++k) {
                    Win& x = win[act[k]];
                    if (sum[size_t(k) * 4] / lp.sym < kEnergyGate) {
                        const int orig = x.delay;
                        for (int q = 1; q <= 3; ++q) {
                            const int cand = orig + q * lp.sym;
                            if (cand + lp.sym > lp.buf) break;
                            if (sum[size_t(k) * 4 + q] / lp.sym >= kEnergyGate) { x.delay = cand; break; }
                        }
                    }
                }
            }
            pt.mark(s, "trial: energy fix");
            // -- :1083-1105 FIR_rx_data baseband at the (coarse-corrected) carrier, frame cut out at `delay`, decimated
            for (int w : act) lp.carrier[w] = rcp->carrier_hz + win[w].coarse_freq_offset;   // effective_carrier_freq, :1074
            auto frames = [&](const std::vector<int>& wins, const int* slot) {     // data-filter baseband of the frame at `delay`, decimated
                std::vector<int> dl(wins.size());
                for (size_t j = 0; j < wins.size(); ++j) dl[j] = win[wins[j]].delay;
                lp.p2b_frames(wins, dl, slot);
            };
            frames(act, nullptr);
            pt.mark(s, "trial: p2b data filter + cut");
            // -- :1108-1131 fine frequency offset (Moose) or the last good one on the final trial; re-mix if it matters
            std::vector<double> f(n, 0.0);
            {
                const int pre_half = lp.pre / 2 == 0 ? 1 : lp.pre / 2;
                hipLaunchKernelGGL(mgpu_fsync_kernel, dim3(n), dim3(256), 0, s, lp.d_frames.as<double>() + size_t(t.Ngi) * 2, lp.frame_n, pre_half,
                                   c->dev.twiddle, lp.d_freq.as<double>());
                HIPCK(hipGetLastError());
                std::vector<double> mul(size_t(n) * 2);
                lp.down(mul.data(), lp.d_freq, size_t(n) * 16);
                for (int k = 0; k < n; ++k) f[k] = moose_hz(mul[2 * k], mul[2 * k + 1], (48000.0 * 50.0 / 256 / 4) / double(t.Nc));
            }
            std::vector<int> rw, rslot;
            for (int k = 0; k < n; ++k) {
                Win& x = win[act[k]];
                if (x.sync_trials == T && rcp->use_last_good_freq_offset && state && state[act[k]].freq_offset_of_last_decoded_message != 0)
                    f[k] = state[act[k]].freq_offset_of_last_decoded_message;
                x.freq = f[k];
                if (!lp.mfsk && std::fabs(f[k]) > kFreqIgnore) { lp.carrier[act[k]] = rcp->carrier_hz + x.coarse_freq_offset + f[k]; rw.push_back(act[k]); rslot.push_back(k); }
            }
            frames(rw, rslot.data());
            pt.mark(s, "trial: Moose + re-mix");
            // -- :1132-1345 the hot path on the data symbols (they start `preamble` symbols into each extracted frame)
            MgpuTapsDev taps{};
            if (!lp.mfsk) taps.mean_H = lp.d_meanh.as<double>();
            launch_frontend(c, lp.d_frames.as<double>() + size_t(lp.pre) * t.Nofdm * 2, n, c->d_llr, c->d_variance, c->d_snrvar, taps, s, lp.frame_n);
            launch_decoder(c, c->d_llr, n, nullptr, nullptr, d_payload_k.as<uint8_t>(), d_stats_k.as<MgpuStatsDev>(), c->d_variance, c->d_snrvar, s);
            // receive_stats.SNR is a double (telecom_system.cc:1343-1396): 10 log10(1 / variance) of the float variance (LS modes), -10 log10 of
            // the re-encoded symbols' error power (ZF modes). The kernels' records carry it as a float (the mgpu_frame_stats ABI); here the
            // argument of the logarithm comes back and the host takes it with the libm the reference calls: the double equals the reference's.
            const bool zf = c->tab.estimator == MGPU_EST_ZF;
            launch_zf_snr(c, n, d_payload_k.as<uint8_t>(), d_stats_k.as<MgpuStatsDev>(), s, 0, zf ? lp.ws.d_snr_k.as<double>() : nullptr);
            std::vector<double> zf_var(zf ? n : 0, 1.0);
            std::vector<float> snr_var(!zf && !lp.mfsk ? n : 0, 1.0f);
            if (zf) lp.down_async(zf_var.data(), lp.ws.d_snr_k, size_t(n) * 8);
            else if (!lp.mfsk) lp.down_async(snr_var.data(), static_cast<const void*>(c->d_snrvar), size_t(n) * 4);
            std::vector<double> mh(n, 1.0);
            if (!lp.mfsk) lp.down_async(mh.data(), lp.d_meanh, size_t(n) * 8);
            lp.down_async(st_k.data(), d_stats_k, size_t(n) * sizeof(MgpuStatsDev));
            lp.down(pay_k.data(), d_payload_k, size_t(n) * t.payload_stride);
            pt.mark(s, "trial: RX path + results");
            for (int k = 0; k < n; ++k) {
                const int w = act[k];
                Win& x = win[w];
                mgpu_receive_stats& r = stats[w];
                r.delay = x.delay;
                if (!lp.mfsk) {
                    r.mean_H = mh[k];
                    if (mh[k] < kMeanHGate) { ++x.skip_h; ++x.sync_trials; r.sync_trials = x.sync_trials; continue; }   // :1269-1280: no decode this trial
                }
                const MgpuStatsDev& d = st_k[k];
                r.iterations_done = d.iterations_done; r.crc = d.crc; r.all_zeros = d.all_zeros;
                std::memcpy(payload + size_t(w) * t.payload_stride, &pay_k[size_t(k) * t.payload_stride], t.payload_stride);
                if (!d.message_decoded) {                            // :1343-1360
                    r.snr_db = -99.9; r.message_decoded = 0;
                    ++x.sync_trials;
                } else {                                             // :1361-1430
                    r.snr_db = double(d.snr_db); r.message_decoded = 1;                       // MFSK: 0.0 (:1362-1367)
                    if (zf) r.snr_db = -10.0 * std::log10(zf_var[k]);                              // ofdm.cc:1622-1635
                    else if (!lp.mfsk) r.snr_db = 10.0 * std::log10(1.0 / double(snr_var[k]));        // :1369-1376
                    x.decoded = true; x.in_loop = false;
                    if (!lp.mfsk) { r.freq_offset = x.freq; if (state) state[w].freq_offset_of_last_decoded_message = x.freq; }
                    if (state) state[w].delay_of_last_decoded_message = x.delay;
                }
                r.sync_trials = x.sync_trials;
            }
            {   // Windows that go on to another trial: in the reference the baseband buffer now holds the FIR_rx_data output of the whole
                // capture window (:1083-1105 wrote it) and a later trial may read it before refreshing it. p2b_frames computed only the
                // samples the RX path reads, so the full buffer is produced here — for the windows that did not decode only.
                // Only for the windows that WILL run another trial (the test at the top of the loop, :931 / :939-944): a window that has used up
                // its trials leaves the loop there and nothing reads its baseband again (the SKIP-H recovery mixes afresh, :1458-1463). Round 6:
                // until then the last round of a call re-mixed every failing window once more for nobody - and that kernel, still in flight when
                // the call returned, was the use-after-return of round 5 (it read its window list in the staging ring). Now the last kernel a
                // call launches is always followed by a round's down() / settle().
                std::vector<int> again;
                for (int w : act) if (win[w].in_loop && !(win[w].sync_trials > T || (lp.mfsk && win[w].sync_trials > 0))) again.push_back(w);
                lp.p2b(again, 1);
            }
        }
        for (int w = 0; w < W; ++w) { stats[w].delay = win[w].delay; stats[w].coarse_metric = win[w].metric; stats[w].sync_trials = win[w].sync_trials; }
        // The call is blocking, to the last kernel (round 5's fault: a kernel still in flight read its window list in the staging ring after the
        // next call had reset it). Every path above ends on a settle(); should one ever not, the stream is waited for here (and by `drain`
        // when an exception unwinds the call).
        if (lp.unsettled) HIPCK(hipStreamSynchronize(s));
        lp.unsettled = false;
    }
}
}  // namespace

// The capture thread's widening of the audio device's samples to the doubles receive_byte works on (radio_capture_thread,
// audioio.c:893-936), on the device: INT32 / INT_MAX (:909), INT16 / 32768.0 (:907), FLOAT32 widened (:905). int -> double is exact and the
// division is the IEEE-754 correctly rounded one on both sides, so the doubles are the CPU's bit for bit; what crosses PCIe is 4 (2) bytes
// per sample instead of 8.
template <typename T>
__global__ __launch_bounds__(256) void mgpu_widen_capture_kernel(const T* __restrict__ in, size_t n, double divisor, double* __restrict__ out) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const double x = double(in[i]);
        out[i] = divisor == 1.0 ? x : x / divisor;
    }
}
namespace {
size_t sample_bytes(int fmt) { return fmt == MGPU_SAMPLES_F64 ? 8 : fmt == MGPU_SAMPLES_INT16 ? 2 : 4; }
void launch_widen(const void* d_in, int fmt, size_t n, double* d_out, hipStream_t s) {
    const dim3 grid(unsigned(std::min<size_t>((n + 255) / 256, 65535u * 4))), block(256);
    if (fmt == MGPU_SAMPLES_INT32) hipLaunchKernelGGL(mgpu_widen_capture_kernel<int32_t>, grid, block, 0, s, static_cast<const int32_t*>(d_in), n, 2147483647.0, d_out);
    else if (fmt == MGPU_SAMPLES_INT16) hipLaunchKernelGGL(mgpu_widen_capture_kernel<int16_t>, grid, block, 0, s, static_cast<const int16_t*>(d_in), n, 32768.0, d_out);
    else hipLaunchKernelGGL(mgpu_widen_capture_kernel<float>, grid, block, 0, s, static_cast<const float*>(d_in), n, 1.0, d_out);
    HIPCK(hipGetLastError());
}
void ensure_stage(mgpu_ctx* c, size_t bytes) {
    if (c->rb_stage_cap >= bytes) return;
    (void)hipFree(c->rb_stage);
    c->rb_stage = nullptr; c->rb_stage_cap = 0;
    HIPCK(hipMalloc(&c->rb_stage, bytes));
    c->rb_stage_cap = bytes;
}
void ensure_compact(mgpu_ctx* c, size_t bytes) {
    if (c->rb_compact_cap >= bytes) return;
    (void)hipFree(c->rb_compact);
    c->rb_compact = nullptr; c->rb_compact_cap = 0;
    HIPCK(hipMalloc(&c->rb_compact, bytes));
    c->rb_compact_cap = bytes;
}

// mgpu_receive_byte_batch / _samples: fmt = the sample format of `capture`
void receive_byte_any(mgpu_ctx* c, const void* capture, int fmt, int W, const mgpu_receive_config* rcp, mgpu_link_state* state, uint8_t* payload,
                      mgpu_receive_stats* stats) {
    need(capture && rcp && payload && stats && W > 0 && W <= c->max_batch, "bad argument (W must be 1..max_batch)");
    need(fmt == MGPU_SAMPLES_F64 || fmt == MGPU_SAMPLES_INT32 || fmt == MGPU_SAMPLES_INT16 || fmt == MGPU_SAMPLES_F32, "unknown sample format");
    need(rcp->time_sync_trials_max >= 1 && rcp->time_sync_trials_max < 64,
         "time_sync_trials_max must be 1..63 (0 makes the reference index its peak table at -1)");
    // every argument is judged before the first copy or kernel is queued: an error return leaves nothing in flight and no state[] entry touched
    if (state && c->tab.mfsk_M == 0) for (int w = 0; w < W; ++w) need(state[w].fixed_delay_plus_one <= 0, "fixed_delay_plus_one: MFSK modes only");
    // Windows in host memory: bringing 1024 mode-8 windows over PCIe takes 13.5 ms as doubles (half that as INT32 / FLOAT32 samples, a quarter
    // as INT16) and the synchroniser + decoder another 17 ms. The windows are independent, so the call is cut into sub-batches: a helper
    // thread uploads them one after another into a staging buffer (a copy from pageable memory holds its calling thread; compact samples are
    // widened there by a kernel on the upload stream), this thread runs the whole receive_byte on each sub-batch as soon as it has landed.
    // The call then lasts the upload plus the receive_byte of the last sub-batch, so small sub-batches win until the fixed cost of the
    // control rounds takes over: 1024 windows in 19.0 ms with sub-batches of 512, 17.1 ms with 256, 19.4 ms with 128 (doubles, end of round
    // 2; the upload alone is 13.5 ms). MERCURY_RB_SUB overrides, MERCURY_NO_PIPELINE=1 disables it.
    static const bool no_pipe = getenv("MERCURY_NO_PIPELINE") != nullptr;
    hipPointerAttribute_t pattr{};
    const bool on_device = hipPointerGetAttributes(&pattr, capture) == hipSuccess && pattr.type == hipMemoryTypeDevice;
    if (!on_device) (void)hipGetLastError();
    const auto& t = c->tab;
    const size_t buf = size_t(t.Nofdm) * mgpu_receive_buffer_nsymb(c) * kInterp;
    const size_t sb = sample_bytes(fmt);
    const char* src = static_cast<const char*>(capture);
    if (!c->rb_stream) HIPCK(hipStreamCreateWithFlags(&c->rb_stream, hipStreamNonBlocking));
    const int kMinSub = 256;
    if (on_device || no_pipe || W < 2 * kMinSub) {
        if (fmt == MGPU_SAMPLES_F64) { receive_byte_impl(c, static_cast<const double*>(capture), W, rcp, state, payload, stats); return; }
        // compact samples, one piece: (upload,) widen into the staging buffer, then the doubles path on device memory
        ensure_stage(c, size_t(W) * buf * 8);
        const void* d_in = capture;
        if (!on_device) {
            ensure_compact(c, size_t(W) * buf * sb);
            HIPCK(hipMemcpyAsync(c->rb_compact, capture, size_t(W) * buf * sb, hipMemcpyHostToDevice, c->rb_stream));
            d_in = c->rb_compact;
        }
        launch_widen(d_in, fmt, size_t(W) * buf, static_cast<double*>(c->rb_stage), c->rb_stream);
        HIPCK(hipStreamSynchronize(c->rb_stream));
        receive_byte_impl(c, static_cast<const double*>(c->rb_stage), W, rcp, state, payload, stats);
        return;
    }
    static const int sub_env = getenv("MERCURY_RB_SUB") ? atoi(getenv("MERCURY_RB_SUB")) : 0;
    // The sub-batches: [offset, count). Doubles are upload-bound (the synchroniser of a sub-batch is over before the next one has landed): equal
    // pieces of 256, so that little is left to do behind the last byte. Compact samples (4 or 2 bytes each) land two to four times faster than
    // they are processed, and every receive_byte_impl call pays its control rounds' fixed ~1.5 ms whatever its size: a short first piece gets the
    // device started, then the pieces grow (1/8, 3/8, 1/2 of the call; INT16: 128 windows, then the rest) - fewer calls, each one's upload still
    // hidden behind its predecessor. Measured on 1024 mode-8 windows (tools/bench_rb_sched.py, profiles/r06_rb_sched.txt), k windows/s, equal pieces of
    // 256 -> these schedules: INT32 66.7 -> 72.4, INT16 71.7 -> 85.3; doubles stay at equal pieces (60.0; 128,384,512 gives 52.9).
    // MERCURY_RB_SUB=<n> forces equal pieces of n, MERCURY_RB_SCHED=<a,b,c,...> a list of piece sizes (the last one repeats).
    std::vector<std::pair<int, int>> pieces;
    {
        std::vector<int> sched;
        if (const char* e = getenv("MERCURY_RB_SCHED")) {
            for (const char* q = e; *q;) { const int v = atoi(q); if (v >= 32) sched.push_back(v); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
        }
        if (sched.empty()) {
            if (sub_env >= 64) sched.push_back(std::min(sub_env, W));
            else if (fmt == MGPU_SAMPLES_F64) sched.push_back(256);
            else if (sb == 2) sched = {128, std::max(256, W - 128)};          // INT16: the whole call lands in the time one piece is processed
            else { const int a = std::max(128, (W / 8 + 63) & ~63); sched = {a, 3 * a, std::max(256, W - 4 * a)}; }
        }
        size_t k = 0;
        for (int off = 0; off < W;) {
            const int n = std::min(sched[std::min(k, sched.size() - 1)], W - off);
            pieces.emplace_back(off, n);
            off += n; ++k;
        }
    }
    const int nsub = int(pieces.size());
    ensure_stage(c, size_t(W) * buf * 8);
    if (fmt != MGPU_SAMPLES_F64) ensure_compact(c, size_t(W) * buf * sb);
    double* stage = static_cast<double*>(c->rb_stage);
    std::mutex m;
    std::condition_variable cv;
    int landed = 0;
    hipError_t failed = hipSuccess;
    std::thread uploader([&] {
        hipError_t e = hipSetDevice(c->cfg.device);
        for (int j = 0; j < nsub; ++j) {
            const int off = pieces[j].first, n = pieces[j].second;
            if (fmt == MGPU_SAMPLES_F64) {
                if (e == hipSuccess) e = hipMemcpyAsync(stage + size_t(off) * buf, src + size_t(off) * buf * 8, size_t(n) * buf * 8, hipMemcpyHostToDevice, c->rb_stream);
            } else {
                char* d_c = static_cast<char*>(c->rb_compact) + size_t(off) * buf * sb;
                if (e == hipSuccess) e = hipMemcpyAsync(d_c, src + size_t(off) * buf * sb, size_t(n) * buf * sb, hipMemcpyHostToDevice, c->rb_stream);
                if (e == hipSuccess) {
                    try { launch_widen(d_c, fmt, size_t(n) * buf, stage + size_t(off) * buf, c->rb_stream); }
                    catch (...) { e = hipErrorLaunchFailure; }
                }
            }
            if (e == hipSuccess) e = hipStreamSynchronize(c->rb_stream);
            {
                std::lock_guard<std::mutex> lk(m);
                failed = e;
                landed = j + 1;
            }
            cv.notify_all();
            if (e != hipSuccess) break;
        }
    });
    try {
        for (int j = 0; j < nsub; ++j) {
            const int off = pieces[j].first, n = pieces[j].second;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return landed > j || failed != hipSuccess; });
                if (failed != hipSuccess) break;
            }
            receive_byte_impl(c, stage + size_t(off) * buf, n, rcp, state ? state + off : nullptr, payload + size_t(off) * t.payload_stride, stats + off);
        }
    } catch (...) {
        uploader.join();
        throw;
    }
    uploader.join();
    if (failed != hipSuccess) throw std::runtime_error(std::string("upload of the capture windows: ") + hipGetErrorString(failed));
}
}  // namespace

extern "C" int mgpu_receive_byte_batch(mgpu_ctx* c, const double* passband, int W, const mgpu_receive_config* rcp, mgpu_link_state* state,
                                       uint8_t* payload, mgpu_receive_stats* stats) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] { receive_byte_any(c, passband, MGPU_SAMPLES_F64, W, rcp, state, payload, stats); });
}

extern "C" int mgpu_receive_byte_batch_samples(mgpu_ctx* c, const void* capture, int sample_format, int W, const mgpu_receive_config* rcp,
                                               mgpu_link_state* state, uint8_t* payload, mgpu_receive_stats* stats) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] { receive_byte_any(c, capture, sample_format, W, rcp, state, payload, stats); });
}

// cl_telecom_system::passband_test_EsN0 (telecom_system.cc:231-330) per Es/N0 point, batched: random payloads -> transmit_byte
// (SINGLE_MESSAGE) -> apply_with_delay (AWGN on the audio, the frame `delay` samples into the capture window) -> receive_byte ->
// cl_error_rate::check over the payload bits. Everything stays on the device except the per-window results receive_byte returns.
extern "C" int mgpu_passband_test_esn0(mgpu_ctx* c, const double* esn0_db, int npoints, long long frames_per_point, uint64_t seed, uint64_t frame0,
                                       double carrier_hz, double output_power_watt, mgpu_error_rate* out, double* windows_out, uint8_t* sent_out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(esn0_db && out && npoints > 0 && frames_per_point > 0 && output_power_watt > 0, "bad argument");
        const auto& t = c->tab;
        const bool mfsk = t.mfsk_M > 0;
        const int total = mgpu_transmit_frame_samples(c);
        const int window = t.Nofdm * mgpu_receive_buffer_nsymb(c) * kInterp;
        const int delay = ((t.preamble + 2) * t.Nofdm + (t.Nfft == 1024 ? 100 : 50)) * kInterp;           // :242-249, :292
        need(delay + total <= window, "the frame does not fit the capture window behind the test delay");
        const int B = int(std::min<long long>(frames_per_point, std::min(c->max_batch, 1024)));            // 1024 windows = 0.76 GB of audio
        const int stride = t.payload_stride, nbytes = t.payload_bytes;
        DevBuf d_pl(size_t(B) * stride), d_audio(size_t(B) * total * 8), d_win(size_t(B) * window * 8);
        std::vector<uint8_t> sent(size_t(B) * stride), got(size_t(B) * stride);
        std::vector<mgpu_receive_stats> st(B);
        std::vector<mgpu_link_state> ls(B);
        const mgpu_transmit_config txc = {carrier_hz, 1.4142135623730951, output_power_watt, 7.0, 10.0, 0, MGPU_SINGLE_MESSAGE, 0};   // physical_config.cc defaults
        const mgpu_receive_config rxc = {carrier_hz, 2, 1, 1, 0};
        hipStream_t s = c->stream;
        for (int p = 0; p < npoints; ++p) {
            // sigma: :236-239 for OFDM; :266-279 calibrates it once per call from the first frame's power for MFSK
            float sigma = mfsk ? 0.0f : 1.0f / float(std::sqrt(std::pow(10.0f, float(esn0_db[p]) / 10.0f)));
            bool calibrated = !mfsk;
            long long be = 0, fe = 0, ok = 0;
            double iters = 0;
            for (long long done = 0; done < frames_per_point; done += B) {
                const int n = int(std::min<long long>(B, frames_per_point - done));
                const uint64_t first = frame0 + uint64_t(p) * uint64_t(frames_per_point) + uint64_t(done);
                hipLaunchKernelGGL(mgpu_gen_payload_kernel, dim3(n), dim3(256), 0, s, seed, first, n, nbytes, stride, d_pl.as<uint8_t>());
                HIPCK(hipGetLastError());
                if (mgpu_transmit_byte_batch_dev(c, d_pl.p, stride, nullptr, n, &txc, d_audio.p, s) != MGPU_OK) throw std::runtime_error(std::string(c->err));
                if (!calibrated) {
                    std::vector<double> a0(total);
                    HIPCK(hipMemcpyAsync(a0.data(), d_audio.p, size_t(total) * 8, hipMemcpyDeviceToHost, s));
                    HIPCK(hipStreamSynchronize(s));
                    double psig = 0;
                    for (int i = 0; i < total; ++i) psig += a0[i] * a0[i];
                    psig /= total;
                    const double bandwidth = 48000.0 * 50.0 / 256 / 4;
                    sigma = float(std::sqrt(2.0 * psig * (48000.0 / 2.0) / (std::pow(10.0, double(float(esn0_db[p])) / 10.0) * bandwidth)));
                    calibrated = true;
                }
                const double ampl = double(sigma / std::sqrt(2.0f));                                   // awgn.cc:68
                hipLaunchKernelGGL(mgpu_passband_channel_kernel, dim3((window + 255) / 256, n), dim3(256), 0, s, d_audio.as<double>(), total, delay, window,
                                   ampl, seed, first, n, d_win.as<double>());
                HIPCK(hipGetLastError());
                HIPCK(hipMemcpyAsync(sent.data(), d_pl.p, size_t(n) * stride, hipMemcpyDeviceToHost, s));
                if (windows_out) HIPCK(hipMemcpyAsync(windows_out + (size_t(p) * frames_per_point + done) * window, d_win.p, size_t(n) * window * 8, hipMemcpyDeviceToHost, s));
                HIPCK(hipStreamSynchronize(s));
                if (sent_out) std::memcpy(sent_out + (size_t(p) * frames_per_point + done) * stride, sent.data(), size_t(n) * stride);
                for (int w = 0; w < n; ++w) ls[w] = mgpu_link_state{-1, 0.0, 0, mfsk ? delay + 1 : 0};      // :293-296 mfsk_fixed_delay
                receive_byte_impl(c, d_win.as<double>(), n, &rxc, ls.data(), got.data(), st.data());
                for (int w = 0; w < n; ++w) {
                    int e = 0;
                    for (int b = 0; b < nbytes; ++b) e += __builtin_popcount(unsigned(sent[size_t(w) * stride + b] ^ got[size_t(w) * stride + b]));
                    be += e; fe += e != 0; ok += st[w].message_decoded != 0;
                    iters += st[w].iterations_done > 0 ? st[w].iterations_done : 0;
                }
            }
            mgpu_error_rate& r = out[p];
            r.esn0_db = esn0_db[p];
            r.Frames_total = frames_per_point; r.Error_frames_total = fe;
            r.Bits_total = frames_per_point * nbytes * 8; r.Error_bits_total = be;
            r.BER = double(be) / double(r.Bits_total); r.FER = double(fe) / double(frames_per_point);
            r.avg_iterations = iters / double(frames_per_point);
            r.crc_ok_frames = ok;
        }
    });
}

