/* mercury_pool.h — the multi-GPU driver of the RX path (SURVEY.md §8 row e, §7.1-7; BASELINE.json north_star: "host code stays
 * C/C++ ... shard across the 8 GPUs of one node by frame batch only, no RCCL collectives").
 *
 * A pool is one mgpu context per device, each with its own host worker thread, stream(s) and page-locked staging. A call
 * hands device g of G the contiguous frame range [floor(F*g/G), floor(F*(g+1)/G)) (frame f -> device
 * floor(f*G/F)); the devices run their shards concurrently, results land in the caller's arrays at their frame positions and
 * the per-device counters (frames, decoded frames, LDPC iterations, wall time) are merged on the host. Nothing crosses between
 * devices: frames are independent once synchronised (the reference's cross-frame state lives in the caller's mgpu_link_state).
 *
 * The caller this mirrors is the reference's single-threaded RX loop, cl_telecom_system::RX_SHM_process_main
 * (source/physical_layer/telecom_system.cc:2266-2390): capture windows in, receive_byte, decoded payloads out — batched here over
 * windows and devices (examples/rx_shm_batch.cpp shows the loop on one context; swap its context for a pool).
 *
 * `devices` may name the same device more than once (two contexts time-sharing one GPU): the outputs do not depend on how a
 * batch is split, which is what the tests check (byte-identical to the single-context call for ragged F).
 */
#ifndef MERCURY_POOL_H
#define MERCURY_POOL_H

#include <stdint.h>

#include "mercury_gpu.h"
#include "mercury_rxloop.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MGPU_POOL_MAX_DEVICES 16

typedef struct mgpu_pool mgpu_pool;

/* merged counters of the most recent pool call */
typedef struct mgpu_pool_counters {
    int n_devices;
    long long frames;                 /* frames (or windows) processed */
    long long decoded;                /* message_decoded != 0; counted from the stats array: 0 when the call was given stats == NULL */
    long long ldpc_iterations;        /* iterations executed, max_iters for frames that never converged (iterations_done clipped);
                                         counted from the stats / iters array: 0 when the call was given none */
    double wall_ms;                   /* the call, host clock */
    int device_frames[MGPU_POOL_MAX_DEVICES];
    double device_ms[MGPU_POOL_MAX_DEVICES];   /* each worker's own wall time for its shard */
} mgpu_pool_counters;

/* cfg->device is ignored; cfg->max_batch is the capacity PER DEVICE (a call may carry up to n_devices * max_batch frames).
 * Fails like mgpu_create does (no device, bad mode ...); mgpu_pool_last_error(NULL) then describes the failure. */
int mgpu_pool_create(const mgpu_config* cfg, const int* devices, int n_devices, mgpu_pool** out);
void mgpu_pool_destroy(mgpu_pool* pool);
int mgpu_pool_size(const mgpu_pool* pool);
/* Placement: every worker thread binds itself to the CPUs of its device's NUMA node (sysfs numa_node of the device's PCI address) and
 * its context allocates its page-locked staging there, so that on a two-socket host eight devices' DMA streams read memory next to
 * their own root complex (set MERCURY_POOL_AFFINITY=0 to leave the threads unbound). Returns that node, -1 when the platform names none.
 * Caller-owned input arrays are the caller's to place: mgpu_alloc_host_near(device, bytes) (mercury_gpu.h). */
int mgpu_pool_device_numa_node(mgpu_pool* pool, int i);
mgpu_ctx* mgpu_pool_context(mgpu_pool* pool, int i);      /* the i-th device's context (for mgpu_get_info and the like) */
const char* mgpu_pool_last_error(mgpu_pool* pool);
int mgpu_pool_last_counters(mgpu_pool* pool, mgpu_pool_counters* out);

/* which frames device g of G gets: [*first, *first + *count) */
void mgpu_pool_shard(int F, int G, int g, int* first, int* count);

/* mgpu_rx_batch / mgpu_ldpc_batch / mgpu_receive_byte_batch over the pool: same arguments, same outputs, blocking. */
int mgpu_pool_rx_batch(mgpu_pool* pool, const double* baseband_c128, int F, uint8_t* payload, mgpu_frame_stats* stats);
int mgpu_pool_ldpc_batch(mgpu_pool* pool, const float* llr, int F, uint8_t* bits, int* iters);
int mgpu_pool_receive_byte_batch(mgpu_pool* pool, const double* passband, int W, const mgpu_receive_config* config,
                                 mgpu_link_state* state, uint8_t* payload, mgpu_receive_stats* stats);

/* ---- device-resident shards: every device's part of the batch already lies in that device's own memory ----------------------
 * The arrays have one entry per pool device, in pool order: counts[g] frames (<= max_batch; mgpu_pool_shard gives the split the
 * host-buffer calls use, any other split is as good), d_in[g] / d_out[g] device pointers ON pool device g (mgpu_device_malloc on
 * mgpu_pool_context(pool, g), or the caller's own HIP allocations). Every device runs its shard on its context's stream; the call
 * returns when all of them have finished. Outputs stay on the devices; only what the merged counters need (24 bytes of stats
 * resp. 4 bytes of iteration count per frame) is read back. Layouts are those of mgpu_rx_batch_dev / mgpu_ldpc_batch_dev.
 * mgpu_pool_txgen_dev fills the shards with the synthetic generator's frames frame0, frame0 + 1, ... numbered across the devices
 * in pool order (device g starts at frame0 + counts[0] + ... + counts[g-1]), so that a pool of any size sees the same frames. */
int mgpu_pool_rx_batch_dev(mgpu_pool* pool, const void* const* d_baseband_c128, const int* counts, void* const* d_payload,
                           void* const* d_stats);
int mgpu_pool_ldpc_batch_dev(mgpu_pool* pool, const void* const* d_llr, const int* counts, void* const* d_bits_opt, void* const* d_iters);
int mgpu_pool_txgen_dev(mgpu_pool* pool, uint64_t seed, uint64_t frame0, const int* counts, double noise_amp, int channel,
                        void* const* d_baseband_c128, void* const* d_payload_opt);

#ifdef __cplusplus
}
#endif
#endif /* MERCURY_POOL_H */
