/* mercury_rxloop.h — the whole of cl_telecom_system::receive_byte, batched.
 *
 * st_receive_stats cl_telecom_system::receive_byte(double* data, int* out)
 * (include/physical_layer/telecom_system.h:142, source/physical_layer/telecom_system.cc:646-1503) takes one capture
 * window of passband audio (buffer_Nsymb * Nofdm * 4 samples at 48 kHz), synchronises, runs up to
 * time_sync_trials_max + 1 decode trials and returns the receive statistics. mgpu_receive_byte_batch does that for W
 * independent windows at once; each window behaves as a fresh call whose cross-call memory (last good delay /
 * frequency offset) is passed in and out through mgpu_link_state.
 *
 * PARITY of the control flow (gates, recoveries, the retry loop — restated from the source) is pinned since round 4 against the
 * reference's own cl_telecom_system::receive_byte: telecom_system.cc compiles unmodified with the reference's own include directories
 * (oracle/ref_ts_harness.cc -> oracle/_ref/libmercury_ref_ts.so), and on randomised capture windows of all 20 modes every field of
 * st_receive_stats (the doubles bit for bit), the payload and the cross-call state equal the CPU restatement's
 * (oracle/mercury_oracle.c:morc_receive_byte, tests/test_receive_byte_vs_reference.py) and this library's
 * (tests/test_receive_byte.py). Every DSP block underneath is checked against the compiled reference on its own as well.
 */
#ifndef MERCURY_RXLOOP_H
#define MERCURY_RXLOOP_H

#include <stdint.h>

#include "mercury_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgpu_receive_config {
    double carrier_hz;              /* carrier_frequency (physical_config.cc:84: bandwidth/2 + 300 + offset) */
    int time_sync_trials_max;       /* physical_config.cc:85 (2); 1..63 */
    int use_last_good_time_sync;    /* physical_config.cc:86 (YES) */
    int use_last_good_freq_offset;  /* physical_config.cc:87 (YES) */
    int coarse_freq_sync_enabled;   /* g_gui_state.coarse_freq_sync_enabled (gui_state.h:143, default false): +-30 Hz search before trial 1 */
} mgpu_receive_config;

/* the members of st_receive_stats that survive from one receive_byte call to the next and steer it */
typedef struct mgpu_link_state {
    int delay_of_last_decoded_message;             /* -1 = none yet (telecom_system.cc:1972) */
    double freq_offset_of_last_decoded_message;
    int mfsk_search_start;                         /* mfsk_search_raw - nUnder_processing_events, >= 0 (telecom_system.cc:683-685) */
    int fixed_delay_plus_one;                      /* cl_telecom_system::mfsk_fixed_delay + 1, 0 = none (so a zeroed state means "search"):
                                                      the known delay of this window in passband samples, as the BER test (:293) and the
                                                      ARQ overflow recapture (arq_common.cc:2830) set it; the time sync and the signal
                                                      level are skipped (:663-672) and the field is cleared, as the reference uses it once.
                                                      MFSK modes only (in the OFDM modes the reference's gates would then read the previous
                                                      call's baseband); an error elsewhere */
} mgpu_link_state;

/* st_receive_stats (telecom_system.h:63-82), the fields this function sets */
typedef struct mgpu_receive_stats {
    int iterations_done, crc, all_zeros, message_decoded;
    double snr_db;                  /* SNR */
    int delay, sync_trials;
    double freq_offset, coarse_metric;
    int frame_overflow_symbols;
    double mean_H;                  /* mean |H| of the last trial that reached the channel estimate (the :1269 gate); -1 if none */
    double signal_strength_dbm;     /* signal_stregth_dbm: 10 log10(mean |x|^2 / 1 mW) of the time-sync baseband (:678, ofdm.cc:1523-1539) */
} mgpu_receive_stats;

/* buffer_Nsymb (data_container.cc:133-143); a capture window is buffer_Nsymb * Nofdm * 4 passband samples */
int mgpu_receive_buffer_nsymb(mgpu_ctx* ctx);

/* passband: [W][buffer_Nsymb*Nofdm*4] doubles, in host memory (pageable or from mgpu_alloc_host) or in device memory — the
 * pointer kind is detected. state: NULL or [W], read and updated. payload: [W][payload_stride]. stats: [W]; state, payload
 * and stats are host arrays. W <= max_batch. Blocking. */
int mgpu_receive_byte_batch(mgpu_ctx* ctx, const double* passband, int W, const mgpu_receive_config* config,
                            mgpu_link_state* state, uint8_t* payload, mgpu_receive_stats* stats);

/* The same on the samples as the audio device delivers them. The reference captures INT32 (radio_capture_thread, audioio.c:744) and its
 * capture thread widens every sample to the double receive_byte works on (audioio.c:893-936): INT32 / INT_MAX (:909), INT16 / 32768.0 (:907),
 * FLOAT32 widened (:905). Here that widening runs on the device - int -> double is exact and the division is the correctly rounded IEEE one
 * on both sides, so the doubles (hence every result) are bit for bit those of mgpu_receive_byte_batch given the widened window - and 4 (2)
 * bytes per sample cross PCIe instead of 8: a mode-8 window is 361 KB instead of 722 KB. capture: [W][buffer_Nsymb*Nofdm*4] samples of
 * sample_format, host or device memory (detected). MGPU_SAMPLES_F64 is mgpu_receive_byte_batch itself. */
#define MGPU_SAMPLES_F64 0
#define MGPU_SAMPLES_INT32 1
#define MGPU_SAMPLES_INT16 2
#define MGPU_SAMPLES_F32 3
int mgpu_receive_byte_batch_samples(mgpu_ctx* ctx, const void* capture, int sample_format, int W, const mgpu_receive_config* config,
                                    mgpu_link_state* state, uint8_t* payload, mgpu_receive_stats* stats);

/* double cl_telecom_system::measure_signal_only(double* data) (telecom_system.cc:1520-1541): the capture window through the
 * time-sync filter and its mean power in dBm (receive_stats.signal_stregth_dbm), without looking for a frame. passband as above
 * (host or device), signal_strength_dbm: [W] host. */
int mgpu_measure_signal_only(mgpu_ctx* ctx, const double* passband, int W, double carrier_hz, double* signal_strength_dbm);

/* cl_error_rate cl_telecom_system::passband_test_EsN0(float EsN0, int max_frame_no) (telecom_system.h:130, .cc:231-330), the audio-path
 * self-simulation (operation mode PLOT_PASSBAND), per Es/N0 point and batched: frames_per_point random payloads -> transmit_byte
 * (SINGLE_MESSAGE, the physical_config.cc defaults, output_power_watt: 0.1 by default, BER_PLOT_passband_process_main :2442 sets 1) -> cl_awgn::apply_with_delay on the audio (awgn.cc:65-77; the frame starts
 * ((preamble_nSymb + 2) * Nofdm + 50) * 4 samples into the capture window; sigma as :236-239, for the MFSK modes calibrated from the first
 * frame's power as :266-279 and with mfsk_fixed_delay set as :293-296) -> receive_byte -> cl_error_rate::check over the payload bits.
 * Differences from the reference's harness, none of which touches reference arithmetic: payloads and noise come from the library's Philox
 * streams (the reference: libc rand()); the part of the window behind the frame holds noise (the reference: whatever the previous
 * iteration left in the buffer); every frame is received with fresh link state (the reference carries receive_stats from frame to
 * frame, and leaves the previous frame's bytes in hd_decoded_data_byte when no trial of a window reaches the decoder: here such a window
 * counts as an all-zero payload; a window that reached the decoder counts with its hard decisions, CRC ok or not, as in the reference). windows_out ([npoints * frames_per_point][window] doubles) / sent_out ([...][payload_stride]): optional host copies of what was
 * received / sent, for tests. Blocking. */
int mgpu_passband_test_esn0(mgpu_ctx* ctx, const double* esn0_db, int npoints, long long frames_per_point, uint64_t seed, uint64_t frame0,
                            double carrier_hz, double output_power_watt, mgpu_error_rate* out, double* windows_out, uint8_t* sent_out);

#ifdef __cplusplus
}
#endif
#endif /* MERCURY_RXLOOP_H */
