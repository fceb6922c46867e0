/* mercury_gpu.h — C-ABI of the MI355X-native Mercury RX physical layer.
 *
 * This is the drop-in boundary for the reference's physical-layer receive path: batches of
 * independent OFDM frames go in, decoded payload bytes + per-frame receive statistics come
 * out. Plain C types only, no exceptions cross the ABI, every entry point returns an int
 * status (MGPU_OK == 0) and never calls exit() (the reference does: ldpc.cc:246-256).
 *
 * Reference interfaces each entry point replaces (paths relative to Rhizomatica/mercury):
 *
 *   mgpu_rx_batch*      the span of cl_telecom_system::receive_byte that runs per synchronised
 *                       frame — source/physical_layer/telecom_system.cc:1132-1345 (agc=1,
 *                       variance_source=1) — and the RX half of baseband_test_EsN0 —
 *                       telecom_system.cc:155-198 (agc=0, variance_source=0). I.e. the calls
 *                       cl_ofdm::symbol_demod / automatic_gain_control / LS_|ZF_channel_estimator /
 *                       restore_channel_amplitude / channel_equalizer / measure_variance / deframer
 *                       (include/physical_layer/ofdm.h:133-146), deinterleaver
 *                       (include/physical_layer/interleaver.h:28-34), cl_psk::demod
 *                       (include/physical_layer/psk.h:55), cl_ldpc::decode
 *                       (include/physical_layer/ldpc.h:90), bit_energy_dispersal, bit_to_byte,
 *                       CRC16_MODBUS_RTU_calc. For cfg 100..102 (ROBUST_0..2) the front half is the
 *                       M == MOD_MFSK branch instead: symbol_demod + cl_mfsk::demod
 *                       (include/physical_layer/mfsk.h:80 -> source/physical_layer/mfsk.cc:288-390),
 *                       telecom_system.cc:1132-1192; agc / variance_source are ignored.
 *   mgpu_ldpc_batch*    int cl_ldpc::decode(const float* data, int* decoded_data)
 *                       (include/physical_layer/ldpc.h:90 -> source/physical_layer/ldpc.cc:266-278)
 *   mgpu_frame_stats    st_receive_stats fields iterations_done, crc, all_zeros, SNR,
 *                       message_decoded (include/physical_layer/telecom_system.h:63-82)
 *   mgpu_config.cfg     cl_telecom_system::load_configuration(int) (telecom_system.cc:2487)
 *   mgpu_txgen_dev      not part of the RX path: the synthetic-workload generator (TX chain of
 *                       telecom_system.cc:384-470 + AWGN at the scale of :141-153) used by bench.py
 *
 * Threading: one context per (device, host thread); calls on one context must be serialised by
 * the caller (the reference DSP objects are non-reentrant as well). *_dev entry points enqueue
 * on the given HIP stream and return without synchronising.
 */
#ifndef MERCURY_GPU_H
#define MERCURY_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGPU_OK 0
#define MGPU_ERR_ARG 1        /* bad argument (cfg out of range, null pointer, F > max_batch ...) */
#define MGPU_ERR_DEVICE 2     /* HIP runtime error; text in mgpu_last_error() */
#define MGPU_ERR_TABLES 3     /* LDPC table blob missing / corrupt */
#define MGPU_ERR_UNSUPPORTED 4

/* decoder selection — reference: cl_ldpc::decoding_algorithm (physical_defines.h:44-45) */
#define MGPU_DEC_GBF 0        /* gradient bit flipping, ldpc_decoder_GBF.cc:25-117 */
#define MGPU_DEC_SPA 1        /* sum-product, double messages, ldpc_decoder_SPA.cc:25-218 (reference default) */
#define MGPU_DEC_MINSUM 2     /* normalised min-sum, fp32 (not in the reference; fast variant) */
#define MGPU_DEC_SPA_FAST 3   /* sum-product in fp32 with hardware exp/log (not the reference's arithmetic; the fast variant for every mode) */

#define MGPU_EST_ZF 0
#define MGPU_EST_LS 1

typedef struct mgpu_ctx mgpu_ctx;

/* Explicit configuration: any (constellation, LDPC rate, preamble length, channel estimator) combination the
 * reference's classes accept, not only the 17 rows load_configuration pairs (telecom_system.cc:2506-2624), as a cfg id:
 *   M: 2, 4, 8, 16 or 32 (MOD_BPSK..MOD_32QAM); rate16: 1, 2, 3, 4, 5, 6, 8 or 14 (sixteenths, ldpc.cc:140-251);
 *   preamble_nsymb: 1..8; estimator: MGPU_EST_ZF / MGPU_EST_LS.
 * Everything else is what physical_config.cc:30-122 and init() (telecom_system.cc:1804-1982) set for every mode: Nc 50, Nfft 256,
 * gi 1/16, pilots Dx 1 / Dy 3 with boost 1.33, LS window 21 x 21, scrambler / pilot seed 0, preamble seed 1 (boost, LS window and
 * the seeds can be overridden per context: mgpu_create_explicit below); amplitude
 * restoration for the PSK constellations (:2647-2654). MOD_64QAM is not accepted: the reference sizes its frame at 8 symbols =
 * 1602 interleaved bits (:1826), which does not fit the N = 1600 codeword (nVirtual = -2), so it cannot run there either.
 * Evaluates to -1 for an unsupported combination. */
#define MGPU_CFG_RATE_INDEX_(r) ((r) == 1 ? 0 : (r) == 2 ? 1 : (r) == 3 ? 2 : (r) == 4 ? 3 : (r) == 5 ? 4 : (r) == 6 ? 5 : (r) == 8 ? 6 : (r) == 14 ? 7 : -1)
#define MGPU_CFG_MOD_INDEX_(m) ((m) == 2 ? 0 : (m) == 4 ? 1 : (m) == 8 ? 2 : (m) == 16 ? 3 : (m) == 32 ? 4 : -1)
#define MGPU_CFG_EXPLICIT(M, rate16, preamble_nsymb, estimator)                                                                   \
    ((MGPU_CFG_MOD_INDEX_(M) < 0 || MGPU_CFG_RATE_INDEX_(rate16) < 0 || (preamble_nsymb) < 1 || (preamble_nsymb) > 8 ||           \
      ((estimator) != 0 && (estimator) != 1))                                                                                       \
         ? -1                                                                                                                     \
         : 1000 + (((MGPU_CFG_MOD_INDEX_(M) * 8 + MGPU_CFG_RATE_INDEX_(rate16)) * 8 + ((preamble_nsymb)-1)) * 2 + (estimator)))

typedef struct mgpu_config {
    int cfg;              /* Mercury CONFIG_0..CONFIG_16, 100..102 = ROBUST_0..2 (MFSK, common_defines.h:63-65), or MGPU_CFG_EXPLICIT(...) */
    int max_iters;        /* nIteration_max, reference default 50 (physical_config.cc:74), CLI 5..50 */
    int decoder;          /* MGPU_DEC_* */
    int agc;              /* 1 = automatic_gain_control before the estimator (receive_byte) */
    int variance_source;  /* 0 = un-equalised grid (baseband_test_EsN0:178), 1 = equalised grid (receive_byte:1291) */
    int device;           /* HIP device ordinal */
    int max_batch;        /* largest F any call will pass; sizes the device workspaces */
    float minsum_alpha;   /* normalisation factor for MGPU_DEC_MINSUM (0 -> 0.8) */
    int mfsk_ctrl_mode;   /* 1 = short MFSK control frames (cl_telecom_system::set_mfsk_ctrl_mode, telecom_system.cc:1572);
                             ignored unless cfg is ROBUST_0 / ROBUST_1 */
    int test_puncture_nBits; /* cl_telecom_system::test_puncture_nBits (telecom_system.h:111, the punctured-LDPC BER-test hook of
                             main.cc:775): > 0 = demodulated MFSK LLRs from this position on are erasures (telecom_system.cc:1186-1192);
                             0 = disabled. MFSK modes only, like the reference */
} mgpu_config;

typedef struct mgpu_info {
    int cfg, M, bits_per_symbol, K, P, N;
    int Nsymb, Nc, Nfft, Ngi, Nofdm;
    int nData, nBits, nPilots, nVirtual, nReal;
    int bit_blk, tf_blk, preamble_nsymb;
    int estimator, amp_restore, ls_window;
    int Cwidth, Vwidth, E;
    int payload_bytes;    /* (nReal-16)/8, telecom_system.cc:332-335 */
    int payload_stride;   /* bytes between consecutive frames in payload arrays = ceil(nReal/8) */
    int frame_samples;    /* complex samples per frame = active_nsymb*Nofdm */
    int mfsk_M, mfsk_nStreams;       /* tones per stream / parallel streams (mfsk.h:36-39); 0 for the OFDM modes */
    int active_nsymb, active_nbits;  /* get_active_nsymb / get_active_nbits (telecom_system.cc:1577-1585) */
} mgpu_info;

/* mirrors st_receive_stats (telecom_system.h:63-82) for the fields this path produces */
typedef struct mgpu_frame_stats {
    int iterations_done;  /* 0 = input already a codeword; max_iters+1 = never converged */
    int crc;              /* CRC16 over nReal/8 bytes; 0 = pass (only computed when !all_zeros) */
    int all_zeros;
    int message_decoded;  /* !(all_zeros || crc != 0), telecom_system.cc:1343-1345 */
    float variance;       /* measure_variance narrowed to float as the callers hold it */
    float snr_db;         /* 10 log10(1/variance) when decoded, else -99.9 (telecom_system.cc:1347,1430) */
} mgpu_frame_stats;

/* optional per-stage taps (host pointers, any may be NULL) for parity testing */
typedef struct mgpu_stage_taps {
    double* grid;       /* [F][Nsymb*Nc][2]  after symbol_demod (+AGC); MFSK: first active_nsymb rows */
    double* H;          /* [F][Nsymb*Nc][2]  channel after estimate/interp/amp-restore */
    double* eq;         /* [F][Nsymb*Nc][2]  equalised grid */
    double* syms;       /* [F][nData][2]     de-framed, time/freq de-interleaved */
    float* llr_demod;   /* [F][nBits]        after cl_psk::demod */
    float* llr_ldpc;    /* [F][1600]         decoder input (bit de-interleaved, re-packed) */
    double* variance;   /* [F]               measure_variance as double */
    double* agc_gain;   /* [F] */
    long long* cycles;  /* [16] shader-clock stamps at the front-end's phase boundaries for a frame from the middle of the batch (profiling aid) */
} mgpu_stage_taps;

int mgpu_create(const mgpu_config* cfg, mgpu_ctx** out);

/* The rest of SURVEY.md §8b's explicit configuration: the parameters physical_config.cc:30-65 gives every mode and
 * cl_telecom_system::load_configuration copies into the DSP objects (telecom_system.cc:2772-2811). A zero field keeps the reference's
 * value. Honoured: pilot_boost (ofdm_pilot_configurator_pilot_boost, a float: 1.33), ls_window (ofdm_LS_window_width = _hight: 20; an
 * even value is incremented as telecom_system.cc:2802-2809 does; 1..21 cells — the front-end reads at most 7 pilots of a window
 * row) and, when seeds_set != 0, the three PRNG seeds (ofdm_pilot_configurator_seed 0, bit_energy_dispersal_seed 0,
 * ofdm_preamble_configurator_seed 1; seed 0 means 1 to __srandom, os_interop.cc:251). They act on the RX path, the synthetic
 * generator and the transmit chain alike.
 * Frame geometry (round 6): Dy (ofdm_pilot_configurator_Dy, the pilot lattice's row period: 3 for every mode's HIGH_DENSITY default;
 * the reference's LOW_DENSITY option is 5, telecom_system.cc:1848-1869) and Nsymb (ofdm_Nsymb, OFDM symbols per frame; 0 = what
 * cl_telecom_system::init selects from the modulation, telecom_system.cc:1810-1826; LOW_DENSITY: 40 BPSK / 20 QPSK / 10 16QAM) are
 * honoured for the OFDM modes as load_configuration copies them (telecom_system.cc:2775-2778): the pilot lattice
 * (cl_pilot_configurator::configure, ofdm.cc:976-1064), every size derived from it (data_container.cc:90-99) and the whole RX / TX /
 * generator path follow. With Dy != 3 the LS estimator takes the kernel's general window walk (slower than the lattice-specialised
 * one; same sums in the same order). Refused (MGPU_ERR_TABLES): a geometry whose data cells hold more bits than a codeword or fewer than
 * its parity plus one payload byte, one with a column of fewer than two pilots, one that does not fit the front-end's LDS carve, the
 * MFSK modes.
 * FIXED, not parameters: Nc = 50, Nfft = 256 (Ngi = 16, Nofdm = 272) and the pilot lattice's column step Dx = 1 — physical_config.cc:35-65
 * gives all 17 modes these values and the kernels are specialised for them. The three fields exist only so that a caller holding the
 * reference's configuration can have it confirmed: 0 or exactly 50 / 256 / 1 is accepted, anything else returns MGPU_ERR_UNSUPPORTED
 * before any device work. params == NULL: mgpu_create. */
typedef struct mgpu_explicit_params {
    float pilot_boost;
    int ls_window;
    int seeds_set;
    unsigned pilot_seed, scrambler_seed, preamble_seed;
    int Nc, Nfft, Dx;          /* fixed at 50 / 256 / 1 (0 = unspecified); see above */
    int Dy;                    /* pilot row period; 0 = the reference's 3 */
    int Nsymb;                 /* OFDM symbols per frame; 0 = the reference's choice for the modulation */
} mgpu_explicit_params;
int mgpu_create_explicit(const mgpu_config* cfg, const mgpu_explicit_params* params, mgpu_ctx** out);
void mgpu_destroy(mgpu_ctx* ctx);
const char* mgpu_last_error(mgpu_ctx* ctx);   /* ctx may be NULL: error of the last failed mgpu_create */
int mgpu_get_info(mgpu_ctx* ctx, mgpu_info* info);

/* Optional page-locked host memory for the arrays handed to the host-buffer entry points: transfers from / to it run at
 * the full PCIe rate (about twice that of pageable memory). Plays the role of the reference's new[]/delete[] for
 * caller-owned buffers (data_container.cc:90-172); any other host memory works too. */
void* mgpu_alloc_host(size_t bytes);
void mgpu_free_host(void* p);

/* ---- placement on a multi-GPU host (SURVEY.md §8 row e) ---------------------------------------------------------------
 * mgpu_device_props_get: what the benchmark's roofline arithmetic and the pool's placement need to know about device `device`
 *   (HIP runtime + sysfs): compute units, engine / memory clock, LDS per compute unit, HBM size, PCI address, and the NUMA node the
 *   device hangs off (-1: the platform reports none). MGPU_ERR_DEVICE without a visible device.
 * mgpu_alloc_host_near: mgpu_alloc_host with the pages taken from `device`'s NUMA node when the platform names one (the calling
 *   thread's memory policy is set to prefer that node for the duration of the call); a context's own staging buffers are
 *   allocated the same way. Free with mgpu_free_host.
 * mgpu_host_numa_node_of_pci / mgpu_host_numa_cpus: the sysfs lookups behind it (no GPU needed): node of a PCI address
 *   ("0000:c1:00.0"), CPUs of a node (returns how many there are; writes at most `max`). */
typedef struct mgpu_device_props {
    int compute_units, clock_khz, memory_clock_khz, lds_bytes_per_cu, wavefront_size, numa_node;
    unsigned long long hbm_bytes;
    char name[64];
    char gcn_arch[32];
    char pci_bus_id[32];
} mgpu_device_props;
int mgpu_device_props_get(int device, mgpu_device_props* out);
void* mgpu_alloc_host_near(int device, size_t bytes);
int mgpu_host_numa_node_of_pci(const char* pci_bus_id);
int mgpu_host_numa_cpus(int node, int* cpus, int max);

/* ---- host-buffer entry points (blocking; copy in, run, copy out) -------------------- */
/* baseband_c128: [F][Nsymb*Nofdm] complex<double>, data symbols only (preamble already
 * stripped, i.e. the pointer receive_byte passes to symbol_demod at telecom_system.cc:1137).
 * payload: [F][payload_stride] bytes (first payload_bytes are the user payload, then CRC lo/hi).
 * stats: [F]. llr_opt: NULL or [F][1600] decoder-input LLRs. */
int mgpu_rx_batch(mgpu_ctx* ctx, const double* baseband_c128, int F, uint8_t* payload,
                  mgpu_frame_stats* stats, float* llr_opt);
/* same, additionally copying out per-stage intermediates */
int mgpu_rx_batch_taps(mgpu_ctx* ctx, const double* baseband_c128, int F, uint8_t* payload,
                       mgpu_frame_stats* stats, const mgpu_stage_taps* taps);
/* llr: [F][1600] float. bits: [F][K] one byte per hard decision (0/1). iters: [F]. */
int mgpu_ldpc_batch(mgpu_ctx* ctx, const float* llr, int F, uint8_t* bits, int* iters);
/* ---- host-side pieces of the library, callable without a GPU (the CPU test suite checks them against the oracle) ----
 * mgpu_host_select_peak: the reference's peak selection (ofdm.cc:1943-1964, overwrite-not-swap partial sort over `size` entries of
 *   which only every `step`-th is a candidate metric) on the candidate metrics alone, as the synchroniser entry points use it.
 * mgpu_host_fir_taps: the filters the library designs (cl_FIR::design, fir_filter.cc:45-162): which 0 = FIR_rx_time_sync,
 *   1 = FIR_rx_data, 2 = FIR_tx1, 3 = FIR_tx2 (the transmit filters depend on the carrier). taps: room for 128 doubles.
 * mgpu_host_preamble_carriers: the mode's preamble symbols in the carrier domain, [n_symbols][50] complex128.
 * mgpu_host_mode_info: what mgpu_get_info reports for a mode — cl_telecom_system::load_configuration's row and the sizes init()
 *   derives from it (telecom_system.cc:2487-3025, :1818-1826, :2910-2911) — from the same host-side table builder mgpu_create runs. */
int mgpu_host_select_peak(const double* cand_vals, int ncand, int step, int size, int location_to_return, int nTrials_max, int* delay,
                          double* correlation);
int mgpu_host_fir_taps(int which, double carrier_hz, double* taps, int* ntaps);
int mgpu_host_preamble_carriers(int cfg, double* carriers_c128, int* n_symbols);
int mgpu_host_mode_info(int cfg, int mfsk_ctrl_mode, mgpu_info* info);
/* test / documentation hook: the fp32 decoders' LDS placement as modelled on the host — out[0], out[1]: LDS cycles per 32-lane gather
 * group (1.0 = conflict-free) of the check pass's posterior reads and the variable update's message reads; out[2] bins; out[3] occupancy */
int mgpu_host_layout_stats(int cfg, double out[4]);

/* mgpu_host_libm_selfcheck (host-only, no GPU): the device code restates the libm of the reference platform the parity was pinned on -
 * x86-64 glibc 2.35: tanh / atanh of the sum-product decoder (ldpc_decoder_SPA.cc:145,156), atan / sincos of restore_channel_amplitude and the
 * receive mixer (misc.cc:34-71, ofdm.cc:2331-2332). This evaluates the HOST's libm and the same restatement compiled for the host on the
 * branch-boundary sets of tests/test_spa_math.py / tests/test_glibc_trig.py plus a fixed pseudo-random sample and reports, per function
 * (index 0 tanh(0.5 q), 1 2 atanh(x), 2 atan, 3 sincos), how many arguments were evaluated and how many results differed in any bit.
 * Returns the number of functions that differ (0: a reference built on this host computes what the device computes). The result is
 * cached (the first call takes 60-90 ms: about a million libm calls). Opt-in at create time: with MERCURY_GPU_LIBM_CHECK=1 in the environment the
 * process's first mgpu_create runs it and writes one line to stderr if anything differs; by default mgpu_create does not run it. */
typedef struct mgpu_libm_report {
    long long evaluated[4], differed[4];
    double first_differing_argument[4];
    int differing;
    char libc_version[32];
} mgpu_libm_report;
int mgpu_host_libm_selfcheck(mgpu_libm_report* out /* may be NULL */);

/* void cl_ldpc::encode(const int* data, int* encoded_data) (ldpc.h:82, ldpc.cc:111-132) for F words: bits [F][K], one byte per bit
 * -> encoded [F][N] = the data followed by the P parity bits. */
int mgpu_ldpc_encode_batch(mgpu_ctx* ctx, const uint8_t* bits, int F, uint8_t* encoded);

/* ---- device-buffer entry points (asynchronous on `stream`, a hipStream_t) ------------ */
int mgpu_rx_batch_dev(mgpu_ctx* ctx, const void* d_baseband_c128, int F, void* d_payload,
                      void* d_stats, void* d_llr_opt, void* stream);
/* The two halves of mgpu_rx_batch_dev on their own: the front-end leaves float LLRs [F][1600] and the float variance that scaled them [F];
 * the decoder half takes any LLRs. Bits, iteration counts, payload bytes, crc / all_zeros / message_decoded / variance of the stats records
 * equal the fused call's. stats.snr_db of the decoder half is 10 log10(1 / d_variance_f) (-99.9 without a variance or a decoded message):
 * the fused call alone knows the PSK modes' variance before amplitude restoration and the zero-forcing modes' re-encoded symbols
 * (telecom_system.cc:1343-1396), so use it when receive_stats.SNR matters. */
int mgpu_frontend_dev(mgpu_ctx* ctx, const void* d_baseband_c128, int F, void* d_llr, void* d_variance_f,
                      void* stream);
int mgpu_ldpc_batch_dev(mgpu_ctx* ctx, const void* d_llr, int F, void* d_bits_opt, void* d_iters,
                        void* d_payload_opt, void* d_stats_opt, const void* d_variance_f_opt, void* stream);
/* For host programs that do not link the HIP runtime themselves (plain C / C++ callers of this C-ABI, the pool in
 * mercury_pool.h): device memory on the context's device, the context's own stream, and blocking copies ordered behind it. */
void* mgpu_device_malloc(mgpu_ctx* ctx, size_t bytes);            /* NULL on failure (mgpu_last_error) */
void mgpu_device_free(mgpu_ctx* ctx, void* d_ptr);
void* mgpu_context_stream(mgpu_ctx* ctx);                          /* the hipStream_t the host-buffer entry points work on */
int mgpu_synchronize(mgpu_ctx* ctx, void* stream_or_null);         /* waits for `stream` (NULL: the context's own stream) */
int mgpu_copy_to_host(mgpu_ctx* ctx, void* dst, const void* d_src, size_t bytes, void* stream_or_null);      /* ordered behind the stream, blocking */
int mgpu_copy_to_device(mgpu_ctx* ctx, void* d_dst, const void* src, size_t bytes, void* stream_or_null);    /* ordered behind the stream, blocking */
/* synthetic workload: frames frame0..frame0+F-1 of the Philox-keyed generator */
int mgpu_txgen_dev(mgpu_ctx* ctx, uint64_t seed, uint64_t frame0, int F, double noise_amp, int channel,
                   void* d_baseband_c128, void* d_payload_opt, void* stream);

/* ---- the reference's self-simulation: cl_telecom_system::baseband_test_EsN0 (telecom_system.cc:96-229) as BER_PLOT_baseband_process_main
 * drives it (:2393-2480: one call per Es/N0 point, cl_error_rate::check per frame, error_rate.cc:48-70) ------------------------
 * For every Es/N0 point: frames_per_point frames of random data bits -> encode ... IFFT (the TX chain of the synthetic generator,
 * DESIGN.md §6) -> AWGN at that Es/N0 (sigma = 10^(-EsN0/20) at the reference's 1/sqrt(Nfft) scale) -> this context's RX path ->
 * decoded bits compared with the sent ones. All on the device, in batches of at most max_batch frames; blocking. The context's
 * agc / variance_source choose the RX variant (baseband_test_EsN0 itself is agc = 0, variance_source = 0). The random source is the
 * generator's Philox stream (seed, frames numbered from frame0 upward, point p uses frames [frame0 + p*frames_per_point, ...)), not
 * libc rand() as in the reference, so curves agree statistically and — frame for frame — with the CPU oracle fed the same frames. */
typedef struct mgpu_error_rate {          /* cl_error_rate (error_rate.h) + two extras */
    double esn0_db;
    long long Frames_total, Error_frames_total, Bits_total, Error_bits_total;
    double BER, FER;
    double avg_iterations;                /* mean of iterations_done */
    long long crc_ok_frames;              /* frames whose CRC self-check passed (meaningful for payloads that carry one) */
} mgpu_error_rate;
int mgpu_baseband_test_esn0(mgpu_ctx* ctx, const double* esn0_db, int npoints, long long frames_per_point, uint64_t seed, uint64_t frame0,
                            int channel, mgpu_error_rate* out);

/* ---- synchroniser building blocks in front of the path, batched over W capture windows,
 * host buffers, blocking. Sample rate 48 kHz and carrier amplitude sqrt(2) as in the reference
 * (telecom_system.cc:69,1569); FIR designs as load_configuration makes them (physical_config.cc:90-98).
 *
 * mgpu_passband_to_baseband = cl_ofdm::passband_to_baseband (ofdm.cc:2316-2339): mixer, 33-tap FIR
 *   (filter 0 = FIR_rx_time_sync, 1 = FIR_rx_data), decimation. Output k of window w is the filtered sample at
 *   input index start[w] + k*decimation (start == NULL means 0), k < count; (start = NULL, count = in_size/decimation)
 *   is the reference call, (start = delay, decimation = 4, count = frame samples) additionally fuses the
 *   rational_resampler call of telecom_system.cc:1105 and filters only the samples that are kept.
 * mgpu_time_sync_preamble = cl_ofdm::time_sync_preamble_with_metric (ofdm.cc:1846-1967), interpolation rate 4.
 * mgpu_freq_sync = cl_ofdm::carrier_sampling_frequency_sync (Moose, ofdm.cc:540-595) on the preamble of each
 *   window; `stride` = complex samples between windows; returns Hz (carrier_freq_width = bandwidth/Nc). */
int mgpu_passband_to_baseband(mgpu_ctx* ctx, const double* passband /*[W][in_size]*/, int W, int in_size,
                              const double* carrier_hz /*[W]*/, int filter, const int* start /*[W] or NULL*/, int count,
                              int decimation, double* out_c128 /*[W][count][2]*/);
int mgpu_time_sync_preamble(mgpu_ctx* ctx, const double* baseband_interp_c128 /*[W][size][2]*/, int W, int size, int step,
                            int location_to_return, int nTrials_max, int* delay /*[W]*/, double* correlation /*[W] or NULL*/);
int mgpu_freq_sync(mgpu_ctx* ctx, const double* baseband_c128 /*[W][stride][2]*/, int W, int stride, double* freq_offset_hz /*[W]*/);
/* MFSK modes: cl_ofdm::time_sync_mfsk (ofdm.cc:1969-2062) with the arguments receive_byte passes
 * (telecom_system.cc:686: the mode's preamble tones / streams, interpolation rate 4): symbol-slot search of the known
 * preamble tone sequence; delay[w] is in interpolated samples. cfg 100..102 only.
 * Every mode: cl_ofdm::detect_ack_pattern (ofdm.cc:2064-2187) on the universal 16-tone pattern of
 * detect_ack_pattern_from_passband (telecom_system.cc:1643-1651, pattern = 1) or the BREAK tones of
 * detect_break_pattern_from_passband (:1698-1706, pattern = 2); metric[w] is the best window metric (0..16),
 * matched[w] (may be NULL) its count of symbols whose expected tone was the band's peak.
 * Both take baseband at the interpolated rate as passband_to_baseband (decimation 1) leaves it. */
int mgpu_time_sync_mfsk(mgpu_ctx* ctx, const double* baseband_interp_c128 /*[W][size][2]*/, int W, int size, int search_start_symb,
                        int* delay /*[W]*/);
int mgpu_detect_ack_pattern(mgpu_ctx* ctx, const double* baseband_interp_c128 /*[W][size][2]*/, int W, int size, int pattern,
                            double* metric /*[W]*/, int* matched /*[W] or NULL*/);
/* cl_telecom_system::detect_ack_pattern_from_passband / detect_break_pattern_from_passband (telecom_system.cc:1628-1655,
 * :1690-1710): passband audio in, mixed down at carrier_hz and filtered with FIR_rx_data on the device, then the detector. */
int mgpu_detect_ack_pattern_from_passband(mgpu_ctx* ctx, const double* passband /*[W][size]*/, int W, int size, double carrier_hz,
                                          int pattern, double* metric /*[W]*/, int* matched /*[W] or NULL*/);
/* duration (ms, HIP events on the launch stream) of the kernel of the most recent synchroniser call */
int mgpu_last_sync_kernel_ms(mgpu_ctx* ctx, float* ms);

/* Kernel timing with HIP events recorded on the launch stream around each kernel.
 * mgpu_enable_timing(ctx,1) resets the counters; mgpu_kernel_ms_avg returns the average launch
 * duration in ms over the launches since then (at most the last 64): [0]=front-end kernel,
 * [1]=LDPC decoder kernel (incl. fused tail). Synchronises on the recorded events. */
int mgpu_enable_timing(mgpu_ctx* ctx, int on);
int mgpu_kernel_ms_avg(mgpu_ctx* ctx, float ms[2], int* n_launches);
/* Frames the fp64 decoder of this context has decided WITHOUT iterating, since the context was created: every |LLR| >= 200 (no NaN) and
 * a parity check that fails. tanh of such inputs is +-1 for ever, the iterations of cl_ldpc::decode (ldpc_decoder_SPA.cc:127-210) change no
 * bit of such a frame, and its iteration count is reported as max_iterations + 1 exactly as the loop would. The zero-forcing modes hand the
 * decoder nothing else behind RX_SHM. For throughput accounting: reported iterations - (max_iterations + 1) x this = iterations executed.
 * Synchronises the context's stream. */
int mgpu_decoder_hard_frames(mgpu_ctx* ctx, long long* frames);
/* Profile of the most recent mgpu_rx_batch call that went through the chunked host-buffer pipeline (F > 1): frames per chunk,
 * number of chunks, and from device events: fill = the first chunk's host-to-device copy (nothing can compute before it has
 * landed), drain = from the last input byte landing to the last result copied back, total = first copy start to that point. */
int mgpu_host_path_last(mgpu_ctx* ctx, int* chunk_frames, int* n_chunks, float* fill_ms, float* drain_ms, float* total_ms);
int mgpu_last_kernel_ms(mgpu_ctx* ctx, float ms[2]);

/* Test hook: the preamble-tone search of cl_ofdm::time_sync_mfsk (ofdm.cc:2026-2061) on given slot energies [W][nslots][Nc] (host): variant 0
 * the host statement (what a single-window call uses), 1 the device kernel (what receive_byte and batched calls launch). search_start: [W] or
 * NULL; delay: [W], interpolated samples. MFSK modes only. */
int mgpu_debug_mfsk_sync(mgpu_ctx* ctx, const double* energy, int W, int nslots, int size, const int* search_start, int variant, int* delay);
/* Test hook, process-wide: which passband_to_baseband kernel the library launches. -1 (default): the sliding-tap kernels where they apply
 * (33 taps, decimation 1 or 4), 0: always the generic one-output-per-lane kernel. Returns the previous setting. */
int mgpu_debug_p2b_variant(int variant);
/* Test hook: the span energies receive_byte's gates and recoveries ask for (telecom_system.cc:758-766, :826-834, :1044-1066: sum of
 * re^2 + im^2 over len samples from off[j] in window wv[j], clipped at the window end, added in sample order) on W host windows of `size`
 * complex samples. variant 0: one wavefront per span; 1: one lane per span (what receive_byte launches from 4096 spans up). */
int mgpu_debug_span_energy(mgpu_ctx* ctx, const double* bb, int W, int size, const int* wv, const int* off, int n, int len, int variant, double* sum,
                           int* cnt);
/* test hook: evaluates the decoder's device tanh/atanh (csrc/spa_math.h) on n host doubles so the tests
 * can compare them bit for bit with the libm the reference calls (ldpc_decoder_SPA.cc:145,156).
 * atanh_out[i] is 0 where |in[i]| >= 1. */
int mgpu_debug_spa_math(mgpu_ctx* ctx, const double* in, int n, double* tanh_out, double* atanh_out);
/* Test hook: the device's peak selection (ofdm.cc:1943-1964, the kernel receive_byte uses from 32 windows up) on n rows of candidate metrics
 * [n][ncand_max] (row w: ncand[w] candidates `step` samples apart in a buffer of size[w] samples, pass loc[w] of nTrials_max), for
 * comparison with mgpu_host_select_peak. */
int mgpu_debug_select_peak(mgpu_ctx* ctx, const double* cand_vals, int n, int ncand_max, const int* ncand, const int* size, const int* loc, int step,
                           int nTrials_max, int* delay, double* corr);

/* Test / tuning hook: which = 0: workgroups of the front-end kernel that fit one compute unit with this context's LDS carve, as the
 * runtime's occupancy calculator reports it (it does not know that LDS is handed out in 1280-byte blocks); 1 / 2: dynamic LDS bytes of a
 * front-end / decoder workgroup; 3: threads of a decoder workgroup; -1 on error. */
int mgpu_debug_occupancy(mgpu_ctx* ctx, int which);

/* Test hook: the Schmidl-Cox metric of every candidate (time_sync_preamble_with_metric, ofdm.cc:1893-1941, before the peak selection) for W
 * windows of `size` interpolated baseband samples; vals: [W][ceil((size - preamble_nSymb*Nofdm*4) / step)]. variant: -1 = the library's choice,
 * 0 = the staged kernel, 1 = the streaming kernel (falls back to the staged one when the geometry does not fit it). start / sub_size (both or
 * neither): search only [start[w], start[w] + sub_size[w]) of window w, as receive_byte's recoveries do; entries behind a window's last candidate
 * keep the fill pattern 0xff. */
int mgpu_debug_tsync_metric(mgpu_ctx* ctx, const double* baseband_interp, int W, int size, int step, int variant, const int* start, const int* sub_size,
                            double* vals);

/* test hook: the device atan / sincos of csrc/glibc_trig.h (restore_channel_amplitude: misc.cc:34-71; receive mixer: ofdm.cc:2331-2332)
 * on n host doubles, for bit-for-bit comparison with the reference platform's libm. */
int mgpu_debug_glibc_trig(mgpu_ctx* ctx, const double* in, int n, double* atan_out, double* sin_out, double* cos_out);

#ifdef __cplusplus
}
#endif
#endif /* MERCURY_GPU_H */
