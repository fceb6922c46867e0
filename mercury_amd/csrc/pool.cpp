// Multi-GPU driver (include/mercury_pool.h): one context + one host worker thread per device, frame-range sharding, counters
// merged on the host, no collectives. Host C++ on top of the C-ABI only — the same calls an application would make, so a pool
// behaves exactly like G independent contexts driven from G threads.
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mercury_pool.h"
#include "numa.hpp"

namespace {

thread_local std::string g_pool_create_error;     // what mgpu_pool_last_error(NULL) reports: the calling thread's last failed create

struct Worker {
    mgpu_ctx* ctx = nullptr;
    int device = 0;
    int numa_node = -1;             // the device's NUMA node (sysfs); the worker thread runs on that node's CPUs when there is one
    bool bound = false;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;       // set by the caller, cleared by the worker
    bool has_job = false, done = false, quit = false;
    int rc = 0;
    double ms = 0;

    void loop() {
        // launches, waits and — on the host-buffer entry points — the staging copies of this device happen on this thread: keep it next to
        // the device's root complex (MERCURY_POOL_AFFINITY=0 leaves the thread where the scheduler puts it)
        const char* e = std::getenv("MERCURY_POOL_AFFINITY");
        if (!(e && e[0] == '0')) bound = mgpu_numa::bind_thread_to_node(numa_node);
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit) return;
            std::function<int()> fn = std::move(job);
            has_job = false;
            lk.unlock();
            const auto t0 = std::chrono::steady_clock::now();
            const int r = fn();
            const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            lk.lock();
            rc = r; ms = dt; done = true;
            cv.notify_all();
        }
    }
    void submit(std::function<int()> fn) {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(fn); has_job = true; done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
};

}  // namespace

struct mgpu_pool {
    std::vector<Worker*> w;
    mgpu_config cfg{};
    std::string err;
    mgpu_pool_counters last{};
    std::mutex call;                // one pool call at a time
};

namespace {

// run shard(ctx, g, first, count) on every worker that has frames; merge return codes (first failure wins) and timings.
// counts == nullptr: the contiguous split of F frames (mgpu_pool_shard); otherwise the caller's own per-device counts.
int run_sharded(mgpu_pool* p, int F, const std::function<int(mgpu_ctx*, int, int, int)>& shard, const int* counts = nullptr) {
    const int G = int(p->w.size());
    const auto t0 = std::chrono::steady_clock::now();
    p->err.clear();                 // mgpu_pool_last_error describes this call, not an earlier one
    std::vector<int> first(G), count(G);
    int at = 0;
    for (int g = 0; g < G; ++g) {
        if (counts) { first[g] = at; count[g] = counts[g]; at += counts[g]; }
        else mgpu_pool_shard(F, G, g, &first[g], &count[g]);
        if (count[g] > 0) {
            mgpu_ctx* ctx = p->w[g]->ctx;
            const int f0 = first[g], n = count[g];
            p->w[g]->submit([=, &shard] { return shard(ctx, g, f0, n); });
        }
    }
    int rc = MGPU_OK;
    p->last = mgpu_pool_counters{};
    p->last.n_devices = G;
    for (int g = 0; g < G; ++g) {
        if (count[g] <= 0) continue;
        const int r = p->w[g]->wait();
        p->last.device_frames[g] = count[g];
        p->last.device_ms[g] = p->w[g]->ms;
        if (r != MGPU_OK && rc == MGPU_OK) {
            rc = r;
            p->err = "device " + std::to_string(p->w[g]->device) + " (context " + std::to_string(g) + "): " + mgpu_last_error(p->w[g]->ctx);
        }
    }
    p->last.frames = F;
    p->last.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

}  // namespace

extern "C" {

void mgpu_pool_shard(int F, int G, int g, int* first, int* count) {
    const long long a = (long long)F * g / G, b = (long long)F * (g + 1) / G;
    if (first) *first = int(a);
    if (count) *count = int(b - a);
}

int mgpu_pool_create(const mgpu_config* cfg, const int* devices, int n_devices, mgpu_pool** out) {
    if (!cfg || !devices || !out || n_devices < 1 || n_devices > MGPU_POOL_MAX_DEVICES) {
        g_pool_create_error = "bad argument (1..16 devices)";
        return MGPU_ERR_ARG;
    }
    *out = nullptr;
    mgpu_pool* p = new mgpu_pool();
    p->cfg = *cfg;
    for (int g = 0; g < n_devices; ++g) {
        mgpu_config c = *cfg;
        c.device = devices[g];
        mgpu_ctx* ctx = nullptr;
        const int rc = mgpu_create(&c, &ctx);
        if (rc != MGPU_OK) {
            g_pool_create_error = "device " + std::to_string(devices[g]) + ": " + mgpu_last_error(nullptr);
            mgpu_pool_destroy(p);
            return rc;
        }
        Worker* w = new Worker();
        w->ctx = ctx;
        w->device = devices[g];
        {
            mgpu_device_props dp;
            w->numa_node = mgpu_device_props_get(devices[g], &dp) == MGPU_OK ? dp.numa_node : -1;
        }
        w->th = std::thread([w] { w->loop(); });
        p->w.push_back(w);
    }
    *out = p;
    return MGPU_OK;
}

void mgpu_pool_destroy(mgpu_pool* p) {
    if (!p) return;
    for (Worker* w : p->w) {
        { std::lock_guard<std::mutex> lk(w->m); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
        mgpu_destroy(w->ctx);
        delete w;
    }
    delete p;
}

int mgpu_pool_size(const mgpu_pool* p) { return p ? int(p->w.size()) : 0; }
int mgpu_pool_device_numa_node(mgpu_pool* p, int i) { return (p && i >= 0 && i < int(p->w.size())) ? p->w[i]->numa_node : -1; }
mgpu_ctx* mgpu_pool_context(mgpu_pool* p, int i) { return (p && i >= 0 && i < int(p->w.size())) ? p->w[i]->ctx : nullptr; }
const char* mgpu_pool_last_error(mgpu_pool* p) { return p ? p->err.c_str() : g_pool_create_error.c_str(); }

int mgpu_pool_last_counters(mgpu_pool* p, mgpu_pool_counters* out) {
    if (!p || !out) return MGPU_ERR_ARG;
    *out = p->last;
    return MGPU_OK;
}

int mgpu_pool_rx_batch(mgpu_pool* p, const double* bb, int F, uint8_t* payload, mgpu_frame_stats* stats) {
    if (!p || !bb || F < 0) return MGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->call);
    mgpu_info info{};
    mgpu_get_info(p->w[0]->ctx, &info);
    if ((long long)F > (long long)p->cfg.max_batch * (long long)p->w.size()) { p->err = "F exceeds n_devices * max_batch"; return MGPU_ERR_ARG; }
    const size_t frame = size_t(info.frame_samples) * 2, stride = size_t(info.payload_stride);
    const int rc = run_sharded(p, F, [&](mgpu_ctx* ctx, int, int f0, int n) {
        return mgpu_rx_batch(ctx, bb + size_t(f0) * frame, n, payload ? payload + size_t(f0) * stride : nullptr, stats ? stats + f0 : nullptr, nullptr);
    });
    if (rc == MGPU_OK && stats)
        for (int f = 0; f < F; ++f) {
            p->last.decoded += stats[f].message_decoded != 0;
            p->last.ldpc_iterations += stats[f].iterations_done > p->cfg.max_iters ? p->cfg.max_iters : stats[f].iterations_done;
        }
    return rc;
}

int mgpu_pool_ldpc_batch(mgpu_pool* p, const float* llr, int F, uint8_t* bits, int* iters) {
    if (!p || !llr || F < 0) return MGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->call);
    mgpu_info info{};
    mgpu_get_info(p->w[0]->ctx, &info);
    if ((long long)F > (long long)p->cfg.max_batch * (long long)p->w.size()) { p->err = "F exceeds n_devices * max_batch"; return MGPU_ERR_ARG; }
    const int rc = run_sharded(p, F, [&](mgpu_ctx* ctx, int, int f0, int n) {
        return mgpu_ldpc_batch(ctx, llr + size_t(f0) * info.N, n, bits ? bits + size_t(f0) * info.K : nullptr, iters ? iters + f0 : nullptr);
    });
    if (rc == MGPU_OK && iters)
        for (int f = 0; f < F; ++f) p->last.ldpc_iterations += iters[f] > p->cfg.max_iters ? p->cfg.max_iters : iters[f];
    return rc;
}

static bool counts_ok(mgpu_pool* p, const int* counts, long long* total) {
    long long F = 0;
    for (size_t g = 0; g < p->w.size(); ++g) {
        if (counts[g] < 0 || counts[g] > p->cfg.max_batch) { p->err = "counts[g] must be 0..max_batch"; return false; }
        F += counts[g];
    }
    *total = F;
    return true;
}

int mgpu_pool_rx_batch_dev(mgpu_pool* p, const void* const* d_bb, const int* counts, void* const* d_payload, void* const* d_stats) {
    if (!p || !d_bb || !counts || !d_payload || !d_stats) return MGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->call);
    long long F = 0;
    if (!counts_ok(p, counts, &F)) return MGPU_ERR_ARG;
    const int G = int(p->w.size());
    std::vector<long long> dec(G, 0), its(G, 0);
    const int max_iters = p->cfg.max_iters;
    const int rc = run_sharded(p, int(F), [&](mgpu_ctx* ctx, int g, int f0, int n) {
        (void)f0;
        void* s = mgpu_context_stream(ctx);
        int r = mgpu_rx_batch_dev(ctx, d_bb[g], n, d_payload[g], d_stats[g], nullptr, s);
        if (r != MGPU_OK) return r;
        std::vector<mgpu_frame_stats> st(n);
        r = mgpu_copy_to_host(ctx, st.data(), d_stats[g], size_t(n) * sizeof(mgpu_frame_stats), s);       // waits for the shard, too
        if (r != MGPU_OK) return r;
        for (int f = 0; f < n; ++f) {
            dec[g] += st[f].message_decoded != 0;
            its[g] += st[f].iterations_done > max_iters ? max_iters : st[f].iterations_done;
        }
        return int(MGPU_OK);
    }, counts);
    for (int g = 0; g < G; ++g) { p->last.decoded += dec[g]; p->last.ldpc_iterations += its[g]; }
    return rc;
}

int mgpu_pool_ldpc_batch_dev(mgpu_pool* p, const void* const* d_llr, const int* counts, void* const* d_bits_opt, void* const* d_iters) {
    if (!p || !d_llr || !counts || !d_iters) return MGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->call);
    long long F = 0;
    if (!counts_ok(p, counts, &F)) return MGPU_ERR_ARG;
    const int G = int(p->w.size());
    std::vector<long long> its(G, 0);
    const int max_iters = p->cfg.max_iters;
    const int rc = run_sharded(p, int(F), [&](mgpu_ctx* ctx, int g, int f0, int n) {
        (void)f0;
        void* s = mgpu_context_stream(ctx);
        int r = mgpu_ldpc_batch_dev(ctx, d_llr[g], n, d_bits_opt ? d_bits_opt[g] : nullptr, d_iters[g], nullptr, nullptr, nullptr, s);
        if (r != MGPU_OK) return r;
        std::vector<int> it(n);
        r = mgpu_copy_to_host(ctx, it.data(), d_iters[g], size_t(n) * sizeof(int), s);
        if (r != MGPU_OK) return r;
        for (int f = 0; f < n; ++f) its[g] += it[f] > max_iters ? max_iters : it[f];
        return int(MGPU_OK);
    }, counts);
    for (int g = 0; g < G; ++g) p->last.ldpc_iterations += its[g];
    return rc;
}

int mgpu_pool_txgen_dev(mgpu_pool* p, uint64_t seed, uint64_t frame0, const int* counts, double noise_amp, int channel,
                        void* const* d_bb, void* const* d_payload_opt) {
    if (!p || !counts || !d_bb) return MGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->call);
    long long F = 0;
    if (!counts_ok(p, counts, &F)) return MGPU_ERR_ARG;
    const int G = int(p->w.size());
    return run_sharded(p, int(F), [&](mgpu_ctx* ctx, int g, int f0, int n) {
        (void)f0;
        void* s = mgpu_context_stream(ctx);
        const int r = mgpu_txgen_dev(ctx, seed, frame0 + uint64_t(f0), n, noise_amp, channel, d_bb[g], d_payload_opt ? d_payload_opt[g] : nullptr, s);
        return r != MGPU_OK ? r : mgpu_synchronize(ctx, s);
    }, counts);
}

int mgpu_pool_receive_byte_batch(mgpu_pool* p, const double* passband, int W, const mgpu_receive_config* config, mgpu_link_state* state,
                                 uint8_t* payload, mgpu_receive_stats* stats) {
    if (!p || !passband || !config || W < 0) return MGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(p->call);
    mgpu_info info{};
    mgpu_get_info(p->w[0]->ctx, &info);
    if ((long long)W > (long long)p->cfg.max_batch * (long long)p->w.size()) { p->err = "W exceeds n_devices * max_batch"; return MGPU_ERR_ARG; }
    const size_t window = size_t(mgpu_receive_buffer_nsymb(p->w[0]->ctx)) * info.Nofdm * 4, stride = size_t(info.payload_stride);
    const int rc = run_sharded(p, W, [&](mgpu_ctx* ctx, int, int f0, int n) {
        return mgpu_receive_byte_batch(ctx, passband + size_t(f0) * window, n, config, state ? state + f0 : nullptr,
                                       payload ? payload + size_t(f0) * stride : nullptr, stats ? stats + f0 : nullptr);
    });
    if (rc == MGPU_OK && stats)
        for (int f = 0; f < W; ++f) {
            p->last.decoded += stats[f].message_decoded != 0;
            if (stats[f].iterations_done > 0) p->last.ldpc_iterations += stats[f].iterations_done > p->cfg.max_iters ? p->cfg.max_iters : stats[f].iterations_done;
        }
    return rc;
}

}  // extern "C"
