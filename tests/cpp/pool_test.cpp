// A C++14 host program (the reference's language level) driving the multi-GPU pool the way a batched
// RX_SHM_process_main (telecom_system.cc:2266-2390) would: F frames in, payload + stats out. Two contexts on device 0
// must return byte-identical output to one context, for ragged F.
//   pool_test <cfg> <F> <bb.bin> <out.bin> [n_contexts=2]
// out.bin: [F][payload_stride] payload, then [F] mgpu_frame_stats, first from the pool, then from a single context.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mercury_pool.h"

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage\n"); return 2; }
    const int cfg = std::atoi(argv[1]), F = std::atoi(argv[2]), n_ctx = argc > 5 ? std::atoi(argv[5]) : 2;
    mgpu_config c;
    std::memset(&c, 0, sizeof c);
    c.cfg = cfg; c.max_iters = 50; c.decoder = MGPU_DEC_SPA; c.agc = 1; c.variance_source = 1; c.device = 0; c.max_batch = F;
    std::vector<int> devices(n_ctx, 0);
    mgpu_pool* pool = nullptr;
    if (mgpu_pool_create(&c, devices.data(), n_ctx, &pool) != MGPU_OK) { std::fprintf(stderr, "pool: %s\n", mgpu_pool_last_error(nullptr)); return 1; }
    mgpu_info info;
    mgpu_get_info(mgpu_pool_context(pool, 0), &info);
    std::vector<double> bb(size_t(F) * info.frame_samples * 2);
    FILE* f = std::fopen(argv[3], "rb");
    if (!f || std::fread(bb.data(), 8, bb.size(), f) != bb.size()) { std::fprintf(stderr, "input\n"); return 1; }
    std::fclose(f);
    std::vector<uint8_t> pay(size_t(F) * info.payload_stride), pay1(pay.size());
    std::vector<mgpu_frame_stats> st(F), st1(F);
    if (mgpu_pool_rx_batch(pool, bb.data(), F, pay.data(), st.data()) != MGPU_OK) { std::fprintf(stderr, "rx: %s\n", mgpu_pool_last_error(pool)); return 1; }
    mgpu_pool_counters k;
    mgpu_pool_last_counters(pool, &k);
    long long frames = 0;
    for (int g = 0; g < k.n_devices; ++g) {
        int a, n;
        mgpu_pool_shard(F, k.n_devices, g, &a, &n);
        if (n != k.device_frames[g]) { std::fprintf(stderr, "shard size\n"); return 1; }
        frames += n;
    }
    if (frames != F || k.frames != F) { std::fprintf(stderr, "counters\n"); return 1; }
    // the same frames again as device-resident shards (mgpu_pool_rx_batch_dev): this program does not link HIP - memory, copies and
    // ordering all go through the C-ABI - and the bytes that come back must be the ones the host-buffer call returned
    {
        const int G = k.n_devices;
        std::vector<const void*> d_in(G);
        std::vector<void*> d_pay(G), d_st(G);
        std::vector<int> cnt(G), first(G);
        for (int g = 0; g < G; ++g) {
            mgpu_pool_shard(F, G, g, &first[g], &cnt[g]);
            mgpu_ctx* ctx = mgpu_pool_context(pool, g);
            const size_t n = size_t(cnt[g] > 0 ? cnt[g] : 1);
            void* in = mgpu_device_malloc(ctx, n * info.frame_samples * 16);
            d_pay[g] = mgpu_device_malloc(ctx, n * info.payload_stride);
            d_st[g] = mgpu_device_malloc(ctx, n * sizeof(mgpu_frame_stats));
            if (!in || !d_pay[g] || !d_st[g]) { std::fprintf(stderr, "device_malloc: %s\n", mgpu_last_error(ctx)); return 1; }
            if (mgpu_copy_to_device(ctx, in, bb.data() + size_t(first[g]) * info.frame_samples * 2, size_t(cnt[g]) * info.frame_samples * 16, nullptr) != MGPU_OK) return 1;
            d_in[g] = in;
        }
        if (mgpu_pool_rx_batch_dev(pool, d_in.data(), cnt.data(), d_pay.data(), d_st.data()) != MGPU_OK) { std::fprintf(stderr, "rx_dev: %s\n", mgpu_pool_last_error(pool)); return 1; }
        mgpu_pool_counters kd;
        mgpu_pool_last_counters(pool, &kd);
        if (kd.decoded != k.decoded || kd.ldpc_iterations != k.ldpc_iterations || kd.frames != F) { std::fprintf(stderr, "dev counters\n"); return 1; }
        std::vector<uint8_t> pd(pay.size());
        std::vector<mgpu_frame_stats> sd(F);
        for (int g = 0; g < G; ++g) {
            mgpu_ctx* ctx = mgpu_pool_context(pool, g);
            if (mgpu_copy_to_host(ctx, pd.data() + size_t(first[g]) * info.payload_stride, d_pay[g], size_t(cnt[g]) * info.payload_stride, nullptr) != MGPU_OK) return 1;
            if (mgpu_copy_to_host(ctx, sd.data() + first[g], d_st[g], size_t(cnt[g]) * sizeof(mgpu_frame_stats), nullptr) != MGPU_OK) return 1;
            mgpu_device_free(ctx, const_cast<void*>(d_in[g])); mgpu_device_free(ctx, d_pay[g]); mgpu_device_free(ctx, d_st[g]);
        }
        if (std::memcmp(pd.data(), pay.data(), pay.size()) != 0 || std::memcmp(sd.data(), st.data(), sizeof(mgpu_frame_stats) * F) != 0) {
            std::fprintf(stderr, "device-resident shards differ from the host-buffer call\n");
            return 1;
        }
    }
    mgpu_pool_destroy(pool);
    mgpu_ctx* one = nullptr;
    if (mgpu_create(&c, &one) != MGPU_OK) { std::fprintf(stderr, "create: %s\n", mgpu_last_error(nullptr)); return 1; }
    if (mgpu_rx_batch(one, bb.data(), F, pay1.data(), st1.data(), nullptr) != MGPU_OK) { std::fprintf(stderr, "rx1: %s\n", mgpu_last_error(one)); return 1; }
    mgpu_destroy(one);
    f = std::fopen(argv[4], "wb");
    std::fwrite(pay.data(), 1, pay.size(), f); std::fwrite(st.data(), sizeof(mgpu_frame_stats), F, f);
    std::fwrite(pay1.data(), 1, pay1.size(), f); std::fwrite(st1.data(), sizeof(mgpu_frame_stats), F, f);
    std::fclose(f);
    std::printf("decoded %lld iterations %lld wall_ms %.3f\n", k.decoded, k.ldpc_iterations, k.wall_ms);
    return 0;
}
