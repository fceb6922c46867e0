/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Plain-C, CPU restatement of the Mercury physical-layer RX hot path (and of the TX chain
 * that is only needed to make synthetic inputs).  It exists to CHECK the HIP implementation:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED for every DSP block (the RX path of the OFDM and MFSK modes, the TX chain, the synchroniser
 * blocks, the ACK / BREAK detector).  The reference ships no golden vectors for this path, so the
 * restatement is pinned against the reference itself: oracle/_ref/libmercury_ref.so is the reference's own DSP
 * objects compiled from /root/reference (oracle/Makefile), and tests/test_oracle_vs_ref.py, tests/test_sync_blocks.py
 * + the committed fixtures in tests/golden/ (golden_rx, golden_mfsk, golden_sync; generated from that build by
 * tests/golden/make_golden.py) require bit-identical outputs for every stage.
 * morc_receive_byte, the restatement of receive_byte's control flow, and morc_get_info's mode table are pinned since round 4 against
 * the reference's own cl_telecom_system (telecom_system.cc compiled unmodified: oracle/_ref/libmercury_ref_ts.so, oracle/ref_ts_harness.cc,
 * tests/test_receive_byte_vs_reference.py) — see the comment at its declaration below.
 *
 * The struct layouts deliberately match oracle/ref_harness.cc (mref_*) so one test body can
 * drive either library.
 */
#ifndef MERCURY_ORACLE_H
#define MERCURY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct morc morc;

typedef struct morc_info {
    int cfg, M, bits_per_symbol, K, P, N;
    int Nsymb, Nc, Nfft, Ngi, Nofdm;
    int nData, nBits, nPilots, nVirtual, nReal;
    int bit_blk, tf_blk, preamble_nsymb;
    int estimator, amp_restore, ls_window;
    int Cwidth, Vwidth, dwidth;
    int payload_bytes;
    int mfsk_M, mfsk_nStreams;      /* 0 for the OFDM modes; cfg 100..102 = ROBUST_0..2 (MFSK) */
    int active_nsymb, active_nbits; /* symbols / interleaved bits on the air (short MFSK control frames) */
} morc_info;

typedef struct morc_rx_out {
    double* grid;     /* [Nsymb*Nc*2] (MFSK: first active_nsymb rows) */
    double* H;        /* [Nsymb*Nc*2] */
    double* H_noamp;  /* [Nsymb*Nc*2] */
    double* eq;       /* [Nsymb*Nc*2] */
    double* syms;     /* [nData*2]    */
    float* llr_demod; /* [nBits]      */
    float* llr_ldpc;  /* [1600]       */
    int* bits;        /* [K]          */
    int* bytes;       /* [ceil(nReal/8)] */
    double variance;
    float variance_f;
    double agc_gain;
    double mean_H;
    int iterations;
    int crc;
    int all_zeros;
    double snr_db;    /* receive_stats.SNR per telecom_system.cc:1343-1396; -99.9 when not decoded */
} morc_rx_out;

#define MORC_FLAG_AGC 1        /* receive_byte variant: automatic_gain_control first */
#define MORC_FLAG_VAR_EQ 2     /* variance from the equalised grid (receive_byte) */
#define MORC_FLAG_NO_LDPC 4    /* stop after the LDPC input is formed */

#define MORC_DEC_GBF 0
#define MORC_DEC_SPA 1

/* tables_path: mercury_ldpc_tables.bin (derived data, see oracle/gen_ldpc_tables.py) */
morc* morc_create(int cfg, int max_iters, const char* tables_path);
/* with the parameters physical_config.cc:35-65 gives every mode spelled out (morc_create: 1.33f, 20, 0, 0, 1) */
morc* morc_create_explicit(int cfg, int max_iters, const char* tables_path, float pilot_boost, int ls_window, unsigned pilot_seed,
                           unsigned scrambler_seed, unsigned preamble_seed);
/* ... and ofdm_Nsymb / ofdm_pilot_configurator_Dy (telecom_system.cc:2772-2778; 0 = the HIGH_DENSITY defaults of init(), :1810-1869) */
morc* morc_create_geometry(int cfg, int max_iters, const char* tables_path, float pilot_boost, int ls_window, unsigned pilot_seed,
                           unsigned scrambler_seed, unsigned preamble_seed, int Nsymb, int Dy);
void morc_destroy(morc*);
void morc_get_info(morc*, morc_info*);
void morc_set_ctrl_mode(morc*, int enable);         /* cl_telecom_system::set_mfsk_ctrl_mode, telecom_system.cc:1572 */
/* cl_telecom_system::test_puncture_nBits (telecom_system.h:111; telecom_system.cc:1186-1192): MFSK LLRs from this position on are erasures */
void morc_set_test_puncture(morc*, int nBits);

void morc_get_frame_types(morc*, int* types);       /* [Nsymb*Nc] 0=DATA 1=PILOT */
void morc_get_pilot_seq(morc*, double* seq);        /* [nPilots*2] */
void morc_get_scrambler(morc*, int* seq);           /* [1600] */
void morc_get_constellation(morc*, double* c);      /* [M*2] */
void morc_prng(unsigned seed, int n, int* out);
unsigned morc_crc16(const int* bytes, int n);

void morc_tx(morc*, const int* bits, int scramble, double* out_c128);
void morc_payload_to_bits(morc*, const int* payload, int nBytes, int* bits);
void morc_rx(morc*, const double* baseband_c128, int flags, morc_rx_out* out);
int morc_ldpc_decode(morc*, const float* llr, int* bits_K, int alg);
void morc_ldpc_encode(morc*, const int* data_K, int* enc_N);     /* cl_ldpc::encode, ldpc.cc:111-132 */

/* ---- synthetic workload generator (the repo's own) ---------------------
 * Philox4x32-10 keyed by seed, counter = (index, stream, frame_lo, frame_hi).
 * channel: 0 = AWGN, 1 = static 2-path h=[1, 0.5 e^{j phi_f}] at delay 6 + AWGN.
 * noise_amp = per-component noise amplitude applied at the reference's 1/sqrt(Nfft) scale
 * (telecom_system.cc:141-153), i.e. 10^(-EsN0/20)/sqrt(2). */
void morc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
void morc_gen_payload(morc*, uint64_t seed, uint64_t frame, int* payload_bytes);
void morc_gen_frame(morc*, uint64_t seed, uint64_t frame, double noise_amp, int channel,
                    double* baseband_c128, int* payload_bytes_out);
/* apply the channel of morc_gen_frame to an already modulated frame (in place) */
void morc_channel(morc*, uint64_t seed, uint64_t frame, double noise_amp, int channel, double* frame_c128);

/* ---- synchroniser building blocks; same argument meaning as the cl_ofdm methods ---- */
void morc_get_preamble(morc*, double* out_c128);   /* [preamble_nsymb*Nc] */
int morc_fir_taps(morc*, int filter, double* taps);   /* 0 = FIR_rx_time_sync, 1 = FIR_rx_data */
void morc_passband_to_baseband(morc*, const double* in, int in_size, double fs, double carrier_hz, double amplitude,
                               int decimation, int filter, double* out_c128);
int morc_time_sync_preamble(morc*, const double* in_c128, int size, int interpolation_rate, int location_to_return,
                            int step, int nTrials_max, double* correlation);
double morc_freq_sync(morc*, const double* in_c128, double carrier_freq_width, int preamble_nSymb, double fs);
int morc_tx_passband(morc*, const int* bits, double fs, double carrier_hz, double amplitude, double* out_passband);

/* ---- cl_telecom_system::transmit_byte (telecom_system.cc:342-556): payload bytes -> CRC -> scramble -> encode -> interleave ->
 * map -> frame -> IFFT+GI -> preamble -> scale -> x4 interpolation + mixer -> peak_clip -> (message_location 3 = SINGLE_MESSAGE)
 * FIR_tx1, FIR_tx2. message_location 4 = NO_FILTER_MESSAGE stops before the filters. out: Nofdm*(Nsymb+preamble)*4 samples
 * (zeros behind a short MFSK control frame). Returns the sample count, -1 = message too long. ---- */
typedef struct morc_tx_config {
    double carrier_hz, carrier_amplitude, output_power_watt, preamble_papr_cut, data_papr_cut;
    unsigned long long start_sample;      /* cl_ofdm::passband_start_sample when the call starts (ofdm.cc:2311-2313) */
    int message_location, reserved;
} morc_tx_config;
int morc_transmit_byte(morc*, const int* payload, int nBytes, const morc_tx_config* cfg, double* out_passband);
int morc_tx_fir_taps(double carrier_hz, int which, double* taps);
/* the signal path of cl_arq_controller::send_batch (arq_common.cc:2224-2248): F messages unfiltered with the carrier running on,
 * the first and last frame repeated as padding, both transmit filters over the concatenation; out: [F][total_frame_size] */
int morc_transmit_batch(morc*, const int* payloads, int stride, const int* nbytes, int F, const morc_tx_config* cfg, double* out);
/* FIRST / MIDDLE / FLUSH_MESSAGE (cfg->message_location 0 / 1 / 2; telecom_system.cc:559-590): F consecutive calls on one 3-frame
 * passband_data_tx_buffer (buffer: [3*total_frame_size], read and updated); out: [F][total_frame_size] */
int morc_transmit_stream(morc*, const int* payloads, int stride, const int* nbytes, int F, const morc_tx_config* cfg, double* buffer, double* out);
/* generate_ack_pattern_passband / generate_break_pattern_passband (telecom_system.cc:1589-1631, :1659-1689): which 1 = ACK,
 * 2 = BREAK; out: 16*Nofdm*4 samples; uses carrier_hz, carrier_amplitude, output_power_watt, data_papr_cut, start_sample */
int morc_generate_ack_pattern_passband(morc*, int which, const morc_tx_config* cfg, double* out_passband);   /* 0 = FIR_tx1 (HPF, Hamming), 1 = FIR_tx2 (LPF, Blackman) */

/* ---- MFSK synchroniser / signalling blocks: time_sync_mfsk (ofdm.cc:1969-2062, arguments of telecom_system.cc:686),
 * detect_ack_pattern (ofdm.cc:2064-2187; which 1 = ACK tones as telecom_system.cc:1643, 2 = BREAK tones as :1698), and the
 * known tone patterns as unscaled time-domain symbols (which 0 = the mode's MFSK preamble, 1 = ACK, 2 = BREAK) ---- */
int morc_mfsk_pattern(morc*, int which, double* out_c128);   /* returns the symbol count; out [n*Nofdm] */
int morc_time_sync_mfsk(morc*, const double* in_c128, int size, int interpolation_rate, int search_start_symb);
double morc_detect_ack_pattern(morc*, const double* in_c128, int size, int interpolation_rate, int which, int* matched);

/* ---- the whole of cl_telecom_system::receive_byte (telecom_system.cc:646-1503): one passband capture window of
 * morc_buffer_nsymb()*Nofdm*4 samples in, payload + receive_stats out. PARITY PINNED (round 4) against the reference's own
 * cl_telecom_system::receive_byte, window for window: tests/test_receive_byte_vs_reference.py. See the comment at the definition. ---- */
typedef struct morc_link_state {          /* the cross-call members of st_receive_stats the loop consults */
    int delay_of_last_decoded_message;    /* -1 = none yet (telecom_system.cc:1972) */
    double freq_offset_of_last_decoded_message;
    int mfsk_search_start;                /* receive_stats.mfsk_search_raw - nUnder_processing_events, clamped at 0 (:683-685) */
    int fixed_delay_plus_one;             /* cl_telecom_system::mfsk_fixed_delay + 1 (0 = none): bypass the time sync once (:663-672); MFSK modes */
} morc_link_state;
typedef struct morc_receive_stats {
    int iterations_done, crc, all_zeros, message_decoded;
    double snr_db;
    int delay, sync_trials;
    double freq_offset, coarse_metric;
    int frame_overflow_symbols;
    double mean_H;                        /* of the last trial that got as far as the channel estimate */
    double signal_strength_dbm;           /* receive_stats.signal_stregth_dbm (telecom_system.cc:678) */
} morc_receive_stats;
int morc_buffer_nsymb(morc*);             /* data_container.cc:133-143 */
void morc_receive_byte(morc*, const double* passband, double carrier_hz, int time_sync_trials_max, int use_last_good_time_sync,
                       int use_last_good_freq_offset, int coarse_freq_sync_enabled, morc_link_state* state_or_null, int* out_bytes,
                       morc_receive_stats* stats);

/* host libm tanh / atanh as the reference's decoder calls them; atanh_out is 0 where |x| >= 1 */
void morc_libm_tanh_atanh(const double* in, int n, double* tanh_out, double* atanh_out);
/* host libm atan and sincos as get_angle / set_complex call them (misc.cc:34-71) */
void morc_libm_atan_sincos(const double* in, int n, double* atan_out, double* sin_out, double* cos_out);

/* cpu_baseline helper: run morc_rx on n frames laid out back to back; returns sum of iterations */
long morc_rx_many(morc*, const double* baseband_c128, int n, int flags, int* iters_out, int* crc_out,
                  unsigned char* payload_out);

#ifdef __cplusplus
}
#endif
#endif
