// Private to the library: the context behind the opaque mgpu_ctx handle, the kernel declarations and the helpers
// that the host-side translation units (api.hip, rxloop.hip) share. Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include <cmath>
#include "../../include/mercury_gpu.h"
#include "../../include/mercury_rxloop.h"
#include "device_tables.h"
#include "tables.hpp"

extern "C" const unsigned char mgpu_ldpc_blob[];
extern "C" const unsigned long mgpu_ldpc_blob_size;

extern "C" size_t mgpu_frontend_lds_bytes(int G, int nPilots, int nBits, int threads);
extern "C" size_t mgpu_spa_lds_bytes(int E, int N);
extern "C" size_t mgpu_gbf_lds_bytes(int N);
extern "C" size_t mgpu_spa_fast_lds_bytes(int Sg, int N);
extern "C" size_t mgpu_txgen_lds_bytes(int G);

extern "C" __global__ void mgpu_frontend_kernel(MgpuDev, const double*, int, float*, float*, float*, double*, MgpuTapsDev);
extern "C" __global__ void mgpu_frontend_kernel_t1024(MgpuDev, const double*, int, float*, float*, float*, double*, MgpuTapsDev);
extern "C" __global__ void mgpu_mfsk_frontend_kernel_m32(MgpuDev, const double*, int, int, float*, float*, float*, MgpuTapsDev);
extern "C" __global__ void mgpu_mfsk_frontend_kernel_m16x2(MgpuDev, const double*, int, int, float*, float*, float*, MgpuTapsDev);
extern "C" int mgpu_mfsk_syms_per_block();
extern "C" __global__ void mgpu_slot_energy_kernel(const double*, int, int, int, const double*, double*);
extern "C" __global__ void mgpu_mfsk_sync_kernel(const double*, int, int, MgpuMfskSync, const int*, int*);
extern "C" __global__ void mgpu_zf_snr_kernel(MgpuDev, const uint8_t*, const double*, int, MgpuStatsDev*, double*);
extern "C" size_t mgpu_zfsnr_lds_bytes(int nData);
extern "C" __global__ void mgpu_p2b_kernel(const double*, int, const double*, const int*, int, int, int, const double*, int, double, double, double*, const int*, const double*, const int*, int);
extern "C" __global__ void mgpu_p2b_slide_d1_kernel(const double*, int, const double*, const int*, int, int, const double*, double, double, double*, const int*, const double*, const int*, int);
extern "C" __global__ void mgpu_p2b_slide_d4_kernel(const double*, int, const double*, const int*, int, int, const double*, double, double, double*, const int*, const double*, const int*, int);
extern "C" __global__ void mgpu_p2b_slide_d1_sincos_kernel(const double*, int, const double*, const int*, int, int, const double*, double, double, double*, const int*, const double*, const int*, int);
extern "C" __global__ void mgpu_p2b_slide_d4_sincos_kernel(const double*, int, const double*, const int*, int, int, const double*, double, double, double*, const int*, const double*, const int*, int);
extern "C" int mgpu_p2b_slide_geometry(int*);
extern "C" __global__ void mgpu_tsync_metric_kernel(const double*, int, const int*, const int*, const int*, int, int, int, int, int, double*);
extern "C" __global__ void mgpu_tsync_metric_dense_kernel(const double*, int, const int*, const int*, const int*, int, int, int, int, int, double*);
extern "C" int mgpu_tsync_coarse_threads();
extern "C" __global__ void mgpu_tsync_metric_stream_kernel(const double*, int, const int*, const int*, const int*, int, double*, int, int, int);
extern "C" void mgpu_tsync_stream_geometry(int*);
extern "C" __global__ void mgpu_tsync_metric_fine_kernel_r4(const double*, int, const int*, const int*, const int*, int, int, int, int, double*);
extern "C" __global__ void mgpu_tsync_metric_fine_kernel_r8(const double*, int, const int*, const int*, const int*, int, int, int, int, double*);
extern "C" int mgpu_tsync_fine_geometry(int*);
extern "C" __global__ void mgpu_tsync_metric_generic_kernel(const double*, int, const int*, const int*, const int*, int, int, int, int, int, double*);
extern "C" __global__ void mgpu_fsync_kernel(const double*, int, int, const double*, double*);
extern "C" __global__ void mgpu_span_energy_kernel(const double*, int, const int*, const int*, int, int, double*, int*);
extern "C" __global__ void mgpu_span_energy_many_kernel(const double*, int, const int*, const int*, int, int, double*, int*);
extern "C" __global__ void mgpu_window_energy_kernel(const double*, int, int, double*);
extern "C" __global__ void mgpu_select_peak_kernel(const double*, const int*, int, int, const int*, const int*, int, int, int*, double*);
extern "C" __global__ void mgpu_decimate_kernel(const double*, int, const int*, const int*, const int*, int, int, double*);
#define DECL_SPA(NE) extern "C" __global__ void mgpu_ldpc_spa_kernel_ne##NE(LdpcDev, const float*, int, uint8_t*, int*, uint8_t*, MgpuStatsDev*, const float*, const float*);
extern "C" int mgpu_spa_max_degree(int ne);     // largest check degree the ne-round instance's unrolled product walk covers
DECL_SPA(4) DECL_SPA(5) DECL_SPA(6) DECL_SPA(7) DECL_SPA(8)
extern "C" __global__ void mgpu_spa_math_probe_kernel(const double*, double*, double*, int);
extern "C" __global__ void mgpu_glibc_trig_probe_kernel(const double*, double*, double*, double*, int);
using DecoderKernel = void (*)(LdpcDev, const float*, int, uint8_t*, int*, uint8_t*, MgpuStatsDev*, const float*, const float*);
extern "C" __global__ void mgpu_gen_payload_kernel(uint64_t, uint64_t, int, int, int, uint8_t*);
extern "C" __global__ void mgpu_passband_channel_kernel(const double*, int, int, int, double, uint64_t, uint64_t, int, double*);
extern "C" __global__ void mgpu_error_count_kernel(const uint8_t*, const uint8_t*, const MgpuStatsDev*, int, int, int, unsigned long long*);
extern "C" __global__ void mgpu_ldpc_gbf_kernel(LdpcDev, const float*, int, uint8_t*, int*, uint8_t*, MgpuStatsDev*, const float*, const float*);
#define DECL_MS(T) extern "C" __global__ void mgpu_ldpc_minsum_kernel_t##T(LdpcDev, const float*, int, uint8_t*, int*, uint8_t*, MgpuStatsDev*, const float*, const float*);
DECL_MS(512)
#define DECL_SF(T) extern "C" __global__ void mgpu_ldpc_spa_fast_kernel_t##T(LdpcDev, const float*, int, uint8_t*, int*, uint8_t*, MgpuStatsDev*, const float*, const float*);
DECL_SF(512)
extern "C" __global__ void mgpu_ldpc_encode_kernel(MgpuDev, const uint8_t*, int, uint8_t*);
extern "C" __global__ void mgpu_txgen_kernel(MgpuDev, uint64_t, uint64_t, int, double, int, double*, uint8_t*, const uint8_t*, int, const int*, int, int);

static_assert(sizeof(MgpuStatsDev) == sizeof(mgpu_frame_stats), "stats layout");

namespace mgpu_detail {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
#define HIPCK(expr)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) throw HipError(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
T* upload(const std::vector<T>& v) {
    T* d = nullptr;
    HIPCK(hipMalloc(&d, v.size() * sizeof(T) + 16));
    HIPCK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

hipError_t host_alloc_on_node(void** p, size_t bytes, int node);   // api.hip
int device_numa_node(int device);

}  // namespace mgpu_detail
using namespace mgpu_detail;

struct mgpu_ctx {
    mgpu_config cfg{};
    mgpu::ModeTables tab;
    MgpuDev dev{};
    LdpcDev ldev{};
    std::vector<void*> owned;       // device allocations freed in destroy
    std::string err;
    int max_batch = 0;
    int numa_node = -1;             // the device's NUMA node (sysfs), -1 when the platform names none: page-locked staging is allocated there
    // workspaces (device)
    double* d_baseband = nullptr;   // lazily sized for the host-buffer entry points
    size_t baseband_cap = 0;
    float* d_llr = nullptr;
    float* d_variance = nullptr;
    float* d_snrvar = nullptr;
    uint8_t* d_payload = nullptr;
    MgpuStatsDev* d_stats = nullptr;
    uint8_t* d_bits = nullptr;
    double* d_eqdata = nullptr;
    double* d_fir[2] = {nullptr, nullptr};   // FIR_rx_time_sync, FIR_rx_data taps
    hipEvent_t sync_ev[2]{};        // around the most recent synchroniser kernel
    float last_sync_ms = -1.f;     // [max_batch][nData] c128, zero-forcing modes only (post-decode SNR)
    int* d_iters = nullptr;
    // single-frame fast path of mgpu_rx_batch: the copy-in / front-end / decoder / copy-out sequence as one hipGraph
    hipGraphExec_t one_frame_graph = nullptr;
    double* d_one_in = nullptr;     // the graph's own one-frame device buffer (never reallocated: the graph holds its address)
    void* h_one_in = nullptr;       // page-locked staging for one frame of samples
    void* h_one_out = nullptr;      // page-locked staging for its payload + stats
    void* rxloop_ws = nullptr;      // device workspace of mgpu_receive_byte_batch, kept between calls (rxloop.hip)
    int rxloop_ws_windows = 0;
    void (*rxloop_ws_free)(void*) = nullptr;
    void* rb_stage = nullptr;       // mgpu_receive_byte_batch from host memory: landing area of the whole call's windows (uploaded by a helper thread)
    size_t rb_stage_cap = 0;
    void* rb_compact = nullptr;     // mgpu_receive_byte_batch_samples: the windows as INT32 / INT16 / FLOAT32 samples, before the widening kernel
    size_t rb_compact_cap = 0;
    hipStream_t rb_stream = nullptr;
    double* d_mix_cs = nullptr;     // receive mixer: cos / sin of the carrier phase per sample index (host libm) for mix_carrier
    double mix_carrier = -1;
    size_t mix_count = 0, mix_cap = 0;
    void* tx_state = nullptr;       // transmit path: preamble baseband, filter taps, carrier table (tx.hip)
    std::vector<double> pre_eq;     // [Nc][2] installed pre_equalization_channel (empty: none); dev.pre_eq is its device copy
    double* d_pre_eq_buf = nullptr; // device copy of pre_eq (owned through keep())
    int pre_eq_version = 0;         // bumped by mgpu_set_pre_equalization_channel: the transmit state rebuilds its preamble
    void (*tx_state_free)(void*) = nullptr;
    hipStream_t stream = nullptr;   // private stream for the host-buffer entry points
    struct Pipe { hipStream_t stream = nullptr; hipEvent_t done = nullptr, copied = nullptr; double* d_in = nullptr; size_t cap = 0; };
    void* h_out = nullptr;          // page-locked staging for the payloads + stats of a pipelined call ([max_batch])
    static constexpr int kPipes = 2;
    Pipe pipe[kPipes];                   // the two chunk pipelines of the blocking host-buffer entry points (api.hip rx_batch_pipelined)
    hipEvent_t hp_ev[4]{};               // call start, first chunk copied, last chunk copied, all done (host-path profile)
    int hp_chunk = 0, hp_nchunks = 0;    // the last pipelined call: frames per chunk, chunks
    float hp_fill_ms = 0, hp_drain_ms = 0, hp_total_ms = 0;
    static constexpr int kEvRing = 64;
    hipEvent_t ev[kEvRing][4]{};    // per launch: front-end start/stop, decoder start/stop
    bool timing = false;
    int ev_count = 0;               // launches recorded since timing was enabled (ring of kEvRing)
    bool ev_fe[kEvRing]{};          // whether the front-end ran in that slot
    size_t lds_fe = 0, lds_dec = 0, lds_tx = 0;
    int fe_threads = 512;           // front-end workgroup size: 1024 when only one workgroup fits a compute unit's LDS anyway (long BPSK frames)
    DecoderKernel spa_kernel = nullptr;
    int dec_threads = 1024;         // workgroup size of the decoder kernel
    int wave_of_wgs = 0;            // decoder workgroups that fill the device once (2 per compute unit); 0 = not asked yet

    template <typename T>
    T* keep(T* p) { owned.push_back(p); return p; }
};


namespace mgpu_detail {

// Workspaces sized by max_batch are created on first use (api.hip)
enum : unsigned { WS_FRONTEND = 1, WS_LLR = 2, WS_OUT = 4, WS_BITS = 8 };
void ensure_workspaces(mgpu_ctx* c, unsigned what);

// HIP caps gridDim*blockDim below 2^32 threads, so very large batches go out in chunks of frames.
constexpr int kMaxFramesPerLaunch = 1 << 21;
template <typename T> T* at(T* p, size_t off) { return p ? p + off : nullptr; }

// frame_stride (complex samples between consecutive frames of d_bb) defaults to the mode's frame_samples
// frame0: index of the call's first frame inside the context's max_batch-sized workspaces (the ZF modes keep their equalised symbols there)
void launch_frontend(mgpu_ctx* c, const double* d_bb, int F, float* d_llr, float* d_var, float* d_snrvar, const MgpuTapsDev& taps,
                     hipStream_t s, int frame_stride = 0, int frame0 = 0);
void launch_zf_snr(mgpu_ctx* c, int F, const uint8_t* d_payload, MgpuStatsDev* d_stats, hipStream_t s, int frame0 = 0, double* d_var_out = nullptr);
void launch_decoder(mgpu_ctx* c, const float* d_llr, int F, uint8_t* d_bits, int* d_iters, uint8_t* d_payload, MgpuStatsDev* d_stats,
                    const float* d_var, const float* d_snrvar, hipStream_t s);

// the reference's peak selection (ofdm.cc:1943-1964): overwrite-not-swap partial sort over an array of `size` entries that
// holds the metric of candidate k at index k*step and 0 elsewhere; returns the index (delay) and value of entry
// `location_to_return` after nTrials_max passes
// the scalar half of cl_ofdm::time_sync_mfsk (ofdm.cc:2004-2060) on the slot energies of one window ([nslots][Nc])
int mfsk_sync_from_energies(const mgpu::ModeTables& t, const double* E, int nslots, int size, int search_start_symb);
void launch_mfsk_sync(mgpu_ctx* c, const double* d_energy, int W, int nslots, int size, const int* d_search_start, int* d_delay, hipStream_t s);
// passband_to_baseband launch for nwin windows (grid y) of `count` outputs each: the sliding-tap kernels for the reference's 33-tap filters
// at decimation 1 / 4, the generic kernel otherwise (api.hip)
void launch_p2b(const double* passband, int in_size, const double* d_carrier, const int* d_start, int start_all, int count, int decim, const double* d_taps,
                int ntaps, double* out, const int* widx, const double* cs, const int* out_row, int row_by_launch, int nwin, hipStream_t s);

void select_peak(const double* cand_vals, int ncand, int step, int size, int location_to_return, int nTrials_max, int* delay, double* corr);

// carrier_sampling_frequency_sync's last step (ofdm.cc:594 with get_angle, misc.cc:34-56) on the sum the Moose kernel returns
inline double moose_hz(double re, double im, double carrier_freq_width) {
    double theta = 0;
    if (re == 0) theta = M_PI / 2;
    else if (re > 0) theta = std::atan(im / re);
    else if (re < 0 && im >= 0) theta = std::atan(im / re) + M_PI;
    else if (re < 0 && im < 0) theta = std::atan(im / re) - M_PI;
    return (theta / M_PI) * carrier_freq_width;
}

// cos / sin table of the receive mixer for `carrier_hz`, at least `count` samples long (built on the host with the reference's libm call,
// cached in the context); the stream is synchronised when the table has to be rebuilt
const double* mixer_table(mgpu_ctx* c, double carrier_hz, size_t count, hipStream_t s);

// Schmidl-Cox metrics of n windows (sync.hip): picks the kernel for the step and the segment lengths.
// d_start / d_widx / d_ncand may be null (search from sample 0, window k = k, ncand_max candidates each).
void launch_tsync_metric(const double* d_bb, int stride, const int* d_start, const int* d_widx, const int* d_ncand, int ncand_max, int n, int step,
                         int pre_nsymb, int ngi_i, int nfft_i, double* d_vals, hipStream_t s, int variant = -1);

// Every entry point that takes a context runs with the context's device current and puts the caller's device back
// afterwards, so one host thread can hold contexts on several GPUs (lazy workspaces, per-call buffers, page-locked
// staging and launches on c->stream all land on cfg.device, whatever the thread's current device was).
struct DeviceScope {
    int prev = -1, want = -1;
    explicit DeviceScope(int device) : want(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) HIPCK(hipSetDevice(want));
    }
    ~DeviceScope() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

inline int guard(mgpu_ctx* c, const std::function<void()>& fn) {
    try {
        if (c) {
            DeviceScope on(c->cfg.device);
            fn();
        } else {
            fn();
        }
        return MGPU_OK;
    } catch (const HipError& e) {
        if (c) c->err = e.what();
        return MGPU_ERR_DEVICE;
    } catch (const std::invalid_argument& e) {
        if (c) c->err = e.what();
        return MGPU_ERR_ARG;
    } catch (const std::exception& e) {
        if (c) c->err = e.what();
        return MGPU_ERR_DEVICE;
    }
}
inline void need(bool ok, const char* what) { if (!ok) throw std::invalid_argument(what); }
struct DevBuf {
    void* p = nullptr;
    void* view = nullptr;       // when set: page-locked host memory holding the buffer's current content, which kernels read in place
    explicit DevBuf(size_t bytes) { HIPCK(hipMalloc(&p, bytes ? bytes : 16)); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    template <typename T> T* as() { return static_cast<T*>(view ? view : p); }
};
// mfsk.cc:82-95, :120-126, :149-155; the universal ACK/BREAK patterns use M = 16, one stream centred in Nc = 50
// (telecom_system.cc:3006), hop step 7, 8 tones sent twice.
constexpr int kAckTones[8] = {4, 7, 5, 12, 13, 1, 9, 15}, kBreakTones[8] = {6, 14, 2, 3, 10, 8, 11, 15};
constexpr int kAckM = 16, kAckNsymb = 16, kAckLen = 8, kAckHop = 7, kAckOffset = 17;
constexpr double kSampleRate = 48000.0;          // telecom_system.cc:1569
const double kCarrierAmplitude = 1.4142135623730951;   // sqrt(2.0), telecom_system.cc:69

}  // namespace mgpu_detail
