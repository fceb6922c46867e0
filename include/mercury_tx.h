/* mercury_tx.h — cl_telecom_system::transmit_byte, batched (the TX mirror taken to the audio samples).
 *
 * void cl_telecom_system::transmit_byte(int* data, int nBytes, double* out, int message_location)
 * (include/physical_layer/telecom_system.h, source/physical_layer/telecom_system.cc:342-382) pads the message to the frame,
 * appends the CRC and calls transmit_bit (:384-556): bit energy dispersal, LDPC encode, bit interleave, PSK/QAM map (or MFSK
 * tone select), time/frequency interleave, framer (pilots), IFFT + guard interval, preamble in front, power scaling,
 * x4 linear interpolation + mixer to the carrier (cl_ofdm::baseband_to_passband, ofdm.cc:2294-2315), peak_clip of the
 * preamble and data parts (:1565-1592) and, for SINGLE_MESSAGE, the two transmit filters FIR_tx1 / FIR_tx2
 * (fir_filter.cc:189-210). mgpu_transmit_byte_batch does that for F messages at once, every step on the GPU.
 *
 * Parity: pinned. The same composition built from the reference's own objects (oracle/ref_harness.cc:mref_transmit_byte)
 * fixes the CPU restatement bit for bit, and the GPU output must equal it bit for bit (tests/test_transmit_byte.py).
 * The FIRST / MIDDLE / FLUSH_MESSAGE overlap-save variants (:559-590), which filter across consecutive calls through the
 * 3-frame passband_data_tx_buffer, are batched too: F consecutive calls in one, the buffer kept in the context between
 * calls (mgpu_transmit_buffer reads or replaces it).
 *
 * pre_equalization_channel (telecom_system.h:186): init() measures the transmit / receive filter chain once per loaded modulation
 * (get_pre_equalization_channel, telecom_system.cc:1954-1958, :3108-3145: 1000 random symbols through symbol_mod, the mixer, FIR_tx1,
 * FIR_tx2, the receive mixer + FIR_rx_data and symbol_demod; mean of sent / received per carrier) and transmit_bit multiplies the
 * preamble and data grids with it (:474-494). mgpu_host_pre_equalization_channel computes that table for a freshly loaded
 * configuration and a carrier (on the host: it is an init-time table like the filters), mgpu_set_pre_equalization_channel installs
 * it in a context; a new context has none installed (= all ones), the C++ mirror installs it in its constructor like init() does.
 * Pinned like the rest: the table and the audio with it equal oracle/ref_harness.cc's composition of the reference's objects bit
 * for bit (tests/test_transmit_byte.py, tests/golden/golden_tx.json).
 */
#ifndef MERCURY_TX_H
#define MERCURY_TX_H

#include <stdint.h>

#include "mercury_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MGPU_FIRST_MESSAGE 0       /* include/common/common_defines.h:197: first call of a stream (telecom_system.cc:559-566); message 0 of the
                                      batch is the FIRST_MESSAGE call, the following ones MIDDLE_MESSAGE calls (TX_RAND_process_main, :2023-2041) */
#define MGPU_MIDDLE_MESSAGE 1      /* :198 (telecom_system.cc:568-574) */
#define MGPU_FLUSH_MESSAGE 2       /* :199 (same path as MIDDLE_MESSAGE) */
#define MGPU_SINGLE_MESSAGE 3      /* include/common/common_defines.h:200 */
#define MGPU_NO_FILTER_MESSAGE 4   /* :201 — stop after peak_clip (what the ARQ batch sender asks for, arq_common.cc:2224) */
#define MGPU_BATCH_MESSAGE 16      /* not a reference constant: the whole signal path of cl_arq_controller::send_batch
                                      (datalink_layer/arq_common.cc:2224-2248) — the F messages as NO_FILTER_MESSAGE frames with the
                                      carrier phase running on from one to the next, the first frame repeated in front and the last
                                      behind, FIR_tx1 / FIR_tx2 over the concatenation; the F filtered frames are returned */

typedef struct mgpu_transmit_config {
    double carrier_hz;          /* carrier_frequency (+ test_tx_carrier_offset); physical_config.cc:84 */
    double carrier_amplitude;   /* telecom_system.cc:69: sqrt(2) */
    double output_power_watt;   /* physical_config.cc:88: 0.1 */
    double preamble_papr_cut;   /* physical_config.cc:115: 7 (dB) */
    double data_papr_cut;       /* :116: 10 (dB) */
    uint64_t start_sample;      /* cl_ofdm::passband_start_sample when the call starts: the carrier phase origin (ofdm.cc:2311-2313) */
    int message_location;       /* MGPU_FIRST / MIDDLE / FLUSH / SINGLE / NO_FILTER / BATCH_MESSAGE; with the first three a call returns,
                                   per message, what the reference's call returns: the PREVIOUS frame filtered with its neighbours */
    int phase_continuous;       /* 0: every message starts at start_sample (F independent transmitters);
                                   1: message f starts at start_sample + f * (active samples), as F consecutive calls would
                                   (always so for MGPU_BATCH_MESSAGE) */
} mgpu_transmit_config;

/* cl_telecom_system::get_pre_equalization_channel for a process that has loaded configuration `cfg` (0..16 or an MGPU_CFG_EXPLICIT id)
 * with carrier_frequency = carrier_hz: channel_c128 [Nc = 50] complex128. Host only (no GPU needed), about 0.2 s. */
int mgpu_host_pre_equalization_channel(int cfg, double carrier_hz, double* channel_c128);
/* the same for the configuration a context holds, explicit parameters included (mgpu_create_explicit: the measurement's random symbols
 * continue the PRNG stream the pilot seed started) */
int mgpu_context_pre_equalization_channel(mgpu_ctx* ctx, double carrier_hz, double* channel_c128);
/* installs the table transmit_bit multiplies the carrier grids with (telecom_system.cc:474-494) in this context, [Nc] complex128;
 * NULL removes it (all ones). Waits for the whole device (hipDeviceSynchronize: the table is read by kernels on every stream of the context). The MFSK modes have none (telecom_system.cc:474). */
int mgpu_set_pre_equalization_channel(mgpu_ctx* ctx, const double* channel_c128);

/* total_frame_size = Nofdm * (Nsymb + preamble_nSymb) * 4 (data_container.cc:159): samples written per message */
int mgpu_transmit_frame_samples(mgpu_ctx* ctx);

/* payload: [F] rows of payload_stride bytes, the message in the first nbytes[f] (NULL: payload_bytes) of each; longer than
 * payload_bytes is an error ("message too long.. not sent."). passband: [F][total_frame_size] doubles; behind a short MFSK
 * control frame the row is zero. Host buffers, blocking. */
int mgpu_transmit_byte_batch(mgpu_ctx* ctx, const uint8_t* payload, int payload_stride, const int* nbytes, int F,
                             const mgpu_transmit_config* config, double* passband);
/* void cl_telecom_system::transmit_bit(int* data, double* out, int message_location) (telecom_system.h:138, .cc:384-556) for F frames:
 * bits [F][nReal] (mgpu_info.nReal), one byte per data bit, sent as they are - no CRC, no padding (transmit_byte makes those and calls
 * this). Everything else as mgpu_transmit_byte_batch. */
int mgpu_transmit_bit_batch(mgpu_ctx* ctx, const uint8_t* bits, int F, const mgpu_transmit_config* config, double* passband);
/* the same on device buffers: asynchronous on `stream` when one is given (the context's work buffers are reused by its next
 * transmit call, so keep one context's calls on one stream); with NULL the context's own stream is used and synchronised */
int mgpu_transmit_byte_batch_dev(mgpu_ctx* ctx, const void* d_payload, int payload_stride, const void* d_nbytes, int F,
                                 const mgpu_transmit_config* config, void* d_passband, void* stream);

/* data_container.passband_data_tx_buffer (data_container.cc:163): the 3 frames of unfiltered audio the FIRST / MIDDLE / FLUSH_MESSAGE
 * calls carry from one to the next, [3 * total_frame_size] host doubles. set = 0 copies it out, set = 1 replaces it. A new context
 * starts with zeros (the reference with uninitialised memory). */
int mgpu_transmit_buffer(mgpu_ctx* ctx, double* buffer, int set);

/* cl_telecom_system::generate_ack_pattern_passband (pattern 1; telecom_system.cc:1589-1631) and
 * generate_break_pattern_passband (pattern 2; :1659-1689): the 16 known tone symbols the detector of mercury_gpu.h
 * (mgpu_detect_ack_pattern_from_passband) looks for, as 16 * Nofdm * 4 passband samples; the same in every mode. Uses
 * carrier_hz, carrier_amplitude, output_power_watt, data_papr_cut and start_sample of `config`. */
int mgpu_generate_ack_pattern_passband(mgpu_ctx* ctx, int pattern, const mgpu_transmit_config* config, double* passband);

/* cl_ofdm::symbol_mod (ofdm.cc:855-860: zero_padder, unnormalised IFFT, gi_adder): [n][Nc] carriers -> [n][Nofdm] samples */
int mgpu_symbol_mod(mgpu_ctx* ctx, const double* carriers_c128, int n_symbols, double* out_c128);

#ifdef __cplusplus
}
#endif
#endif
