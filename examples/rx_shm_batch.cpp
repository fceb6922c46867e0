// RX_SHM_process_main, batched: the reference's receive loop
// (source/physical_layer/telecom_system.cc:2266-2390) takes one capture window of passband audio at a time, calls
// receive_byte on it and writes the decoded payload to the "/mercury-comm" ring that client programs read
// (examples/receiver.c). This program does the same for B windows per step through the library's C-ABI only:
//
//     capture windows (here: a file of raw doubles, W windows back to back)
//       -> mgpu_receive_byte_batch   (synchronise, gate, retry, demodulate, decode: include/mercury_rxloop.h)
//       -> mgpu_shm_publish_decoded  (the ring of include/mercury_shm.h; same objects / protocol as the reference)
//
// Each window slot keeps its own link state (last good delay / frequency offset) from one step to the next, the way
// the reference keeps it in receive_stats between calls — think of B radios, or B channels of one wide-band capture.
//
// The capture file holds the windows as doubles (.f64) or as the audio device delivers them - INT32 (.i32: what the reference's capture
// thread asks for, audioio.c:744), INT16 (.i16) or FLOAT32 (.f32) - in which case they are widened on the device as audioio.c:893-936 widens
// them on the host (mgpu_receive_byte_batch_samples): same results, half or a quarter of the bytes over PCIe.
//
//   usage: rx_shm_batch <cfg> <windows.f64|.i32|.i16|.f32> <batch> [carrier_hz] [ring_name]
//   build: g++ -O2 -std=c++14 -I include examples/rx_shm_batch.cpp -L mercury_amd -lmercury_gpu -Wl,-rpath,$PWD/mercury_amd -o rx_shm_batch
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mercury_gpu.h"
#include "mercury_rxloop.h"
#include "mercury_shm.h"

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <cfg> <windows.f64> <batch> [carrier_hz] [ring_name]\n", argv[0]); return 2; }
    const int cfg = atoi(argv[1]), batch = atoi(argv[3]);
    const double carrier = argc > 4 ? atof(argv[4]) : 48000.0 * 50.0 / 256 / 4 / 2 + 300;      // physical_config.cc:84
    const char* ring_name = argc > 5 ? argv[5] : MGPU_SHM_PAYLOAD_NAME;

    mgpu_config gc = {};
    gc.cfg = cfg; gc.max_iters = 50; gc.decoder = MGPU_DEC_SPA; gc.agc = 1; gc.variance_source = 1; gc.device = 0; gc.max_batch = batch;
    mgpu_ctx* rx = nullptr;
    if (mgpu_create(&gc, &rx) != MGPU_OK) { fprintf(stderr, "mgpu_create: %s\n", mgpu_last_error(nullptr)); return 1; }
    mgpu_info info;
    mgpu_get_info(rx, &info);
    const size_t window = size_t(mgpu_receive_buffer_nsymb(rx)) * info.Nofdm * 4;

    // the ring: attach to Mercury's if it is running, otherwise create it (circular_buf_connect_shm / _init_shm)
    mgpu_shm* ring = nullptr;
    if (mgpu_shm_connect(ring_name, MGPU_SHM_PAYLOAD_BUFFER_SIZE, &ring) != MGPU_OK &&
        mgpu_shm_create(ring_name, MGPU_SHM_PAYLOAD_BUFFER_SIZE, &ring) != MGPU_OK) { fprintf(stderr, "cannot open the payload ring\n"); return 1; }

    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 1; }
    const char* ext = strrchr(argv[2], '.');
    const int fmt = ext && !strcmp(ext, ".i32") ? MGPU_SAMPLES_INT32 : ext && !strcmp(ext, ".i16") ? MGPU_SAMPLES_INT16 :
                    ext && !strcmp(ext, ".f32") ? MGPU_SAMPLES_F32 : MGPU_SAMPLES_F64;
    const size_t sample = fmt == MGPU_SAMPLES_F64 ? 8 : fmt == MGPU_SAMPLES_INT16 ? 2 : 4;
    void* windows = mgpu_alloc_host(window * batch * sample);                                     // page-locked capture buffer
    std::vector<uint8_t> payload(size_t(batch) * info.payload_stride);
    std::vector<mgpu_receive_stats> stats(batch);
    std::vector<mgpu_frame_stats> fstats(batch);
    std::vector<mgpu_link_state> link(batch);
    for (auto& l : link) { l.delay_of_last_decoded_message = -1; l.freq_offset_of_last_decoded_message = 0; l.mfsk_search_start = 0; }
    const mgpu_receive_config rc = {carrier, 2, 1, 1, 0};                                      // physical_config.cc:85-87 defaults

    long total = 0, decoded = 0, lost = 0;
    for (int step = 0;; ++step) {
        const size_t got = fread(windows, sample * window, batch, f);
        if (got == 0) break;
        const int W = int(got);
        if (mgpu_receive_byte_batch_samples(rx, windows, fmt, W, &rc, link.data(), payload.data(), stats.data()) != MGPU_OK) {
            fprintf(stderr, "receive_byte_batch: %s\n", mgpu_last_error(rx));
            return 1;
        }
        for (int w = 0; w < W; ++w) { fstats[w] = mgpu_frame_stats(); fstats[w].message_decoded = stats[w].message_decoded; }
        int pub = 0, drop = 0;
        mgpu_shm_publish_decoded(ring, payload.data(), fstats.data(), W, info.payload_stride, info.payload_bytes, &pub, &drop);   // :2326-2333
        total += W; decoded += pub + drop; lost += drop;
        for (int w = 0; w < W; ++w)     // the status line of :2336-2345
            printf("step %d window %d: %s  SNR %5.1f dB  level %6.1f dBm  delay %d  trials %d  iterations %d\n", step, w,
                   stats[w].message_decoded ? "decoded" : "-      ", stats[w].snr_db, stats[w].signal_strength_dbm, stats[w].delay,
                   stats[w].sync_trials, stats[w].iterations_done);
    }
    printf("%ld windows, %ld decoded, %ld lost to a full ring\n", total, decoded, lost);
    fclose(f);
    mgpu_free_host(windows);
    mgpu_shm_close(ring);
    mgpu_destroy(rx);
    return 0;
}
