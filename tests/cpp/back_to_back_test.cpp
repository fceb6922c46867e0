// Calls of the C-ABI straight after one another, no pause between them (a C / C++ host, unlike the Python tests, comes back within
// microseconds): every entry point must be finished with the context's staging areas and workspaces when it returns. Round 5 found
// mgpu_receive_byte_batch returning with its last kernel in flight (profiles/NOTES.md R5.4); this runs the situations that exposed it and
// their neighbours and demands that a repeated call on the same input returns the same bytes.
//   usage: back_to_back_test <cfg> <windows.bin (double [n][window])> <n>       exit 0 = every repetition identical
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mercury_gpu.h"
#include "mercury_rxloop.h"

#define CK(x) do { int rc_ = (x); if (rc_ != MGPU_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mgpu_last_error(ctx)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    const int cfg = atoi(argv[1]), n = atoi(argv[3]);
    mgpu_config gc;
    memset(&gc, 0, sizeof gc);
    gc.cfg = cfg; gc.max_iters = 50; gc.decoder = MGPU_DEC_SPA; gc.agc = 1; gc.variance_source = 1; gc.max_batch = 1024;
    mgpu_ctx* ctx = nullptr;
    if (mgpu_create(&gc, &ctx) != MGPU_OK) { fprintf(stderr, "create: %s\n", mgpu_last_error(nullptr)); return 3; }
    mgpu_info info;
    CK(mgpu_get_info(ctx, &info));
    const size_t win = size_t(mgpu_receive_buffer_nsymb(ctx)) * info.Nofdm * 4;
    std::vector<double> base(size_t(n) * win);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(base.data(), 8, base.size(), f) != base.size()) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
    fclose(f);
    const int Wmax = 600;
    std::vector<double> host(size_t(Wmax) * win);
    for (int w = 0; w < Wmax; ++w) memcpy(&host[size_t(w) * win], &base[size_t(w % n) * win], win * 8);
    double* dev = static_cast<double*>(mgpu_device_malloc(ctx, host.size() * 8));
    if (!dev) { fprintf(stderr, "device malloc: %s\n", mgpu_last_error(ctx)); return 3; }
    CK(mgpu_copy_to_device(ctx, dev, host.data(), host.size() * 8, nullptr));
    const mgpu_receive_config rc = {48000.0 * 50.0 / 256 / 4 / 2 + 300 + 1.5, 2, 1, 1, 0};

    // the reference result of every distinct window: one window per call, a pause (synchronize) between calls
    std::vector<mgpu_receive_stats> ref_stats(n);
    std::vector<uint8_t> ref_payload(size_t(n) * info.payload_stride);
    memset(ref_stats.data(), 0, ref_stats.size() * sizeof(mgpu_receive_stats));          // padding bytes compare equal
    for (int w = 0; w < n; ++w) {
        CK(mgpu_receive_byte_batch(ctx, &base[size_t(w) * win], 1, &rc, nullptr, &ref_payload[size_t(w) * info.payload_stride], &ref_stats[w]));
        CK(mgpu_synchronize(ctx, nullptr));
    }
    std::vector<mgpu_receive_stats> stats(Wmax);
    std::vector<uint8_t> payload(size_t(Wmax) * info.payload_stride);
    long bad = 0, calls = 0;
    auto check = [&](int first, int W, const char* what) {
        for (int w = 0; w < W; ++w) {
            const int k = (first + w) % n;
            if (memcmp(&stats[w], &ref_stats[k], sizeof(mgpu_receive_stats)) != 0 ||
                memcmp(&payload[size_t(w) * info.payload_stride], &ref_payload[size_t(k) * info.payload_stride], info.payload_stride) != 0) {
                if (bad < 5) fprintf(stderr, "%s: call %ld window %d (source %d) differs: decoded %d/%d delay %d/%d iters %d/%d\n", what, calls, w, k,
                                     stats[w].message_decoded, ref_stats[k].message_decoded, stats[w].delay, ref_stats[k].delay, stats[w].iterations_done,
                                     ref_stats[k].iterations_done);
                ++bad;
            }
        }
        ++calls;
    };
    // device-resident windows, sizes that alternate between large and tiny (the tiny call follows the large one's last kernel immediately)
    const int sizes[] = {256, 8, 256, 3, 64, 1, 512, 8, 130, 256, 256, 8};
    for (int rep = 0; rep < 3; ++rep) {
        int first = 0;
        for (int W : sizes) {
            if (first + W > Wmax) first = 0;
            memset(stats.data(), 0, stats.size() * sizeof(mgpu_receive_stats));
            CK(mgpu_receive_byte_batch(ctx, dev + size_t(first) * win, W, &rc, nullptr, payload.data(), stats.data()));
            check(first, W, "device");
            first += W;
        }
    }
    // host windows: one piece (< 512) and the pipelined path (>= 512: sub-batches run back to back behind the uploads)
    const int hsizes[] = {520, 8, 600, 256, 512, 3};
    for (int rep = 0; rep < 2; ++rep)
        for (int W : hsizes) {
            memset(stats.data(), 0, stats.size() * sizeof(mgpu_receive_stats));
            CK(mgpu_receive_byte_batch(ctx, host.data(), W, &rc, nullptr, payload.data(), stats.data()));
            check(0, W, "host");
        }
    // other entry points that use the same workspaces, interleaved without pauses
    std::vector<double> dbm(64), dbm0(64);
    CK(mgpu_measure_signal_only(ctx, host.data(), 64, rc.carrier_hz, dbm0.data()));
    for (int rep = 0; rep < 6; ++rep) {
        memset(stats.data(), 0, stats.size() * sizeof(mgpu_receive_stats));
        CK(mgpu_receive_byte_batch(ctx, dev, 64, &rc, nullptr, payload.data(), stats.data()));
        check(0, 64, "interleaved receive_byte");
        CK(mgpu_measure_signal_only(ctx, host.data(), 64, rc.carrier_hz, dbm.data()));
        if (memcmp(dbm.data(), dbm0.data(), 64 * 8) != 0) { if (bad < 5) fprintf(stderr, "measure_signal_only differs (rep %d)\n", rep); ++bad; }
    }
    mgpu_device_free(ctx, dev);
    mgpu_destroy(ctx);
    printf("cfg %d: %ld calls, %ld differing windows\n", cfg, calls, bad);
    return bad ? 1 : 0;
}
