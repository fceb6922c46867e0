// TEST INFRASTRUCTURE. C entry points onto the reference's OWN cl_telecom_system, compiled unmodified from /root/reference by
// oracle/Makefile (target ref_ts -> oracle/_ref/libmercury_ref_ts.so): telecom_system.cc, main.cc, gui/gui_main.cc, audioio/audioio.c
// and the DSP translation units, with the reference's own include directories and -DMERCURY_GUI_ENABLED (without it telecom_system.cc
// does not compile: :949 reads g_gui_state outside its #ifdef). Nothing of the reference is restated, stubbed or defined here:
//   * g_verbose / test_tx_carrier_offset come from main.cc (compiled with -Dmain=mercury_reference_main so that the library has no main),
//     get_gui_state() from gui_main.cc, capture_buffer / capture_prep_mutex / tx_transfer from audioio.c;
//   * what those units call in the GUI toolkit, the ARQ layer and the audio drivers stays UNDEFINED in the shared object (function symbols
//     bind lazily and nothing on the paths driven here calls them); the two driver tables audioio.c points at (ffalsa, ffpulse: ALSA /
//     PulseAudio, which this image lacks) are data symbols and are left undefined-weak by objcopy. No header, library or function of the
//     reference or of its dependencies is written by this repository.
// What it is for: cl_telecom_system::load_configuration (telecom_system.cc:2487-3025) and cl_telecom_system::receive_byte
// (telecom_system.cc:646-1503) as the reference runs them, to pin oracle/mercury_oracle.c:morc_get_info and morc_receive_byte — the
// restatement of receive_byte's control flow that rounds 1-3 could only check by reading (tests/test_receive_byte_vs_reference.py).
// The object is driven the way main.cc drives it for RX_SHM (:505, :821-835): operation_mode, load_configuration(cfg), then
// receive_byte per capture window as RX_SHM_process_main does (:2266-2390).
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>

#include "physical_layer/telecom_system.h"
#include "gui/gui_state.h"

namespace {
// the reference prints its configuration and per-call diagnostics on stdout; keep the test log readable
struct Silence {
    int saved;
    Silence() {
        fflush(stdout);
        saved = dup(1);
        const int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1);
        close(nul);
    }
    ~Silence() {
        fflush(stdout);
        dup2(saved, 1);
        close(saved);
    }
};
}  // namespace

extern "C" {

struct mrefts_link_state {         // = morc_link_state / mgpu_link_state
    int delay_of_last_decoded_message;
    double freq_offset_of_last_decoded_message;
    int mfsk_search_start;
    int fixed_delay_plus_one;
};
struct mrefts_receive_stats {      // = morc_receive_stats / mgpu_receive_stats
    int iterations_done, crc, all_zeros, message_decoded;
    double snr_db;
    int delay, sync_trials;
    double freq_offset, coarse_metric;
    int frame_overflow_symbols;
    double mean_H;                 // not a member of st_receive_stats: reported as NaN here, the test does not compare it
    double signal_strength_dbm;
};

void* mrefts_create(int cfg) {
    Silence s;
    cl_telecom_system* t = new cl_telecom_system();
    t->operation_mode = RX_SHM;                      // main.cc:505
    t->load_configuration(cfg);                      // main.cc:824
    return t;
}
// The same with the frame geometry set where a caller of the reference sets it: the public members ofdm_Nsymb / ofdm_pilot_configurator_Dy of
// default_configurations_telecom_system (physical_config.cc:38-40; AUTO_SELLECT unless given), which load_configuration copies into the DSP
// objects (telecom_system.cc:2775-2778) in front of init() (telecom_system.cc:1806-1869). <= 0: left at AUTO_SELLECT.
void* mrefts_create_geometry(int cfg, int Nsymb, int Dy) {
    Silence s;
    cl_telecom_system* t = new cl_telecom_system();
    t->operation_mode = RX_SHM;
    if (Nsymb > 0) t->default_configurations_telecom_system.ofdm_Nsymb = Nsymb;
    if (Dy > 0) t->default_configurations_telecom_system.ofdm_pilot_configurator_Dy = Dy;
    t->load_configuration(cfg);
    return t;
}
void mrefts_destroy(void* h) {
    Silence s;
    delete static_cast<cl_telecom_system*>(h);
}

// the members SURVEY.md section 0 printed (tests/golden/survey_mode_table.json), in this order:
// M K P N Nsymb Nc Nfft Ngi Nofdm nData nBits nPilots nVirtual nReal bit_blk tf_blk preamble_nsymb estimator amp_restore ls_window buffer_Nsymb payload_bytes
int mrefts_info(void* h, int* o) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    int i = 0;
    o[i++] = int(t->M); o[i++] = t->ldpc.K; o[i++] = t->ldpc.P; o[i++] = t->ldpc.N;
    o[i++] = t->data_container.Nsymb; o[i++] = t->data_container.Nc; o[i++] = t->data_container.Nfft; o[i++] = t->data_container.Ngi;
    o[i++] = t->data_container.Nofdm; o[i++] = t->data_container.nData; o[i++] = t->data_container.nBits;
    o[i++] = t->ofdm.pilot_configurator.nPilots;
    o[i++] = t->ldpc.N - t->data_container.nBits; o[i++] = t->data_container.nBits - t->ldpc.P;
    o[i++] = t->bit_interleaver_block_size; o[i++] = t->time_freq_interleaver_block_size;
    o[i++] = t->data_container.preamble_nSymb; o[i++] = t->ofdm.channel_estimator; o[i++] = t->ofdm.channel_estimator_amplitude_restoration;
    o[i++] = t->ofdm.LS_window_width; o[i++] = t->data_container.buffer_Nsymb; o[i++] = t->get_frame_size_bytes();
    return i;
}

int mrefts_buffer_samples(void* h) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    return t->data_container.Nofdm * t->data_container.buffer_Nsymb * t->data_container.interpolation_rate;
}

// One capture window through cl_telecom_system::receive_byte with the cross-call members set from `state` (a zeroed / -1 state = a link that
// has decoded nothing yet, telecom_system.cc:1971-1973) and read back afterwards, as mgpu_receive_byte_batch / morc_receive_byte define a call.
void mrefts_receive_byte(void* h, const double* passband, double carrier_hz, int time_sync_trials_max, int use_last_good_time_sync,
                         int use_last_good_freq_offset, int coarse_freq_sync_enabled, mrefts_link_state* state, int* out_bytes,
                         mrefts_receive_stats* rs) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    t->carrier_frequency = carrier_hz;
    t->time_sync_trials_max = time_sync_trials_max;
    t->use_last_good_time_sync = use_last_good_time_sync;
    t->use_last_good_freq_offset = use_last_good_freq_offset;
    g_gui_state.coarse_freq_sync_enabled.store(coarse_freq_sync_enabled != 0);
    // a call's own outputs start from the constructor's values (telecom_system.cc:38-51; crc / all_zeros / coarse_metric, which it leaves
    // unset, from 0): receive_byte assigns some of them only on the paths that reach them, and a window must not inherit the previous one's
    st_receive_stats& q = t->receive_stats;
    q.iterations_done = -1; q.delay = 0; q.sync_trials = 0; q.freq_offset = 0; q.message_decoded = NO; q.SNR = -99.9; q.signal_stregth_dbm = -999;
    q.crc = 0; q.all_zeros = 0; q.coarse_metric = 0; q.frame_overflow_symbols = 0;
    t->receive_stats.delay_of_last_decoded_message = state ? state->delay_of_last_decoded_message : -1;
    t->receive_stats.freq_offset_of_last_decoded_message = state ? state->freq_offset_of_last_decoded_message : 0.0;
    t->receive_stats.mfsk_search_raw = state ? state->mfsk_search_start : 0;
    t->data_container.nUnder_processing_events = 0;
    t->mfsk_fixed_delay = state && state->fixed_delay_plus_one > 0 ? state->fixed_delay_plus_one - 1 : -1;
    const int n = mrefts_buffer_samples(h);
    // RX_SHM_process_main hands receive_byte its own copy of the capture buffer (:2304-2307)
    memcpy(t->data_container.ready_to_process_passband_delayed_data, passband, size_t(n) * sizeof(double));
    const st_receive_stats r = t->receive_byte(t->data_container.ready_to_process_passband_delayed_data, out_bytes);
    rs->iterations_done = r.iterations_done; rs->crc = r.crc; rs->all_zeros = r.all_zeros; rs->message_decoded = r.message_decoded;
    rs->snr_db = r.SNR; rs->delay = r.delay; rs->sync_trials = r.sync_trials; rs->freq_offset = r.freq_offset;
    rs->coarse_metric = r.coarse_metric; rs->frame_overflow_symbols = r.frame_overflow_symbols;
    rs->mean_H = __builtin_nan(""); rs->signal_strength_dbm = r.signal_stregth_dbm;
    if (state) {
        state->delay_of_last_decoded_message = t->receive_stats.delay_of_last_decoded_message;
        state->freq_offset_of_last_decoded_message = t->receive_stats.freq_offset_of_last_decoded_message;
        const int ss = t->receive_stats.mfsk_search_raw - t->data_container.nUnder_processing_events;
        state->mfsk_search_start = ss < 0 ? 0 : ss;
        state->fixed_delay_plus_one = t->mfsk_fixed_delay >= 0 ? t->mfsk_fixed_delay + 1 : 0;
    }
}

// The two random sources of the reference's self-simulations: libc rand() (cl_awgn, awgn.cc:41-95) and the reference's own generator
// (__random, os_interop.cc:192-283: the data bits of baseband_test_EsN0 / passband_test_EsN0), so that two objects can be given the same frames.
void mrefts_seed(unsigned libc_seed, unsigned reference_seed) {
    srand(libc_seed);
    __srandom(reference_seed);
}

// receive_byte WITHOUT any reset of the object's members between calls: consecutive capture windows through one cl_telecom_system as
// RX_SHM_process_main runs them (telecom_system.cc:2266-2390), so that what a call leaves UNWRITTEN in receive_stats on the paths that do not
// reach the decoder (:646-1131) is visible. Only the receiver's settings are (re)stated; nUnder_processing_events / mfsk_fixed_delay are the
// caller's to set (mrefts_set_loop_members). returned / held: {iterations_done delay delay_of_last_decoded_message sync_trials
// message_decoded crc all_zeros mfsk_search_raw frame_overflow_symbols} + {freq_offset freq_offset_of_last_decoded_message SNR
// signal_stregth_dbm coarse_metric} of the struct receive_byte returned and of the member afterwards.
void mrefts_receive_byte_raw(void* h, const double* passband, double carrier_hz, int time_sync_trials_max, int use_last_good_time_sync,
                             int use_last_good_freq_offset, int coarse_freq_sync_enabled, int* out_bytes, int* returned_ints,
                             double* returned_doubles, int* held_ints, double* held_doubles) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    t->carrier_frequency = carrier_hz;
    t->time_sync_trials_max = time_sync_trials_max;
    t->use_last_good_time_sync = use_last_good_time_sync;
    t->use_last_good_freq_offset = use_last_good_freq_offset;
    g_gui_state.coarse_freq_sync_enabled.store(coarse_freq_sync_enabled != 0);
    const int n = mrefts_buffer_samples(h);
    memcpy(t->data_container.ready_to_process_passband_delayed_data, passband, size_t(n) * sizeof(double));
    const st_receive_stats r = t->receive_byte(t->data_container.ready_to_process_passband_delayed_data, out_bytes);
    const st_receive_stats* both[2] = {&r, &t->receive_stats};
    int* ints[2] = {returned_ints, held_ints};
    double* dbl[2] = {returned_doubles, held_doubles};
    for (int k = 0; k < 2; k++) {
        const st_receive_stats& q = *both[k];
        int* o = ints[k]; double* d = dbl[k];
        o[0] = q.iterations_done; o[1] = q.delay; o[2] = q.delay_of_last_decoded_message; o[3] = q.sync_trials; o[4] = q.message_decoded;
        o[5] = q.crc; o[6] = q.all_zeros; o[7] = q.mfsk_search_raw; o[8] = q.frame_overflow_symbols;
        d[0] = q.freq_offset; d[1] = q.freq_offset_of_last_decoded_message; d[2] = q.SNR; d[3] = q.signal_stregth_dbm; d[4] = q.coarse_metric;
    }
}
// the loop's own members around receive_byte: data_container.nUnder_processing_events (:683, reset by the process loops :2158), the one-shot
// mfsk_fixed_delay (:663-672), receive_stats.mfsk_search_raw (ARQ sets it). A value of INT_MIN leaves the member alone.
void mrefts_set_loop_members(void* h, int nUnder_processing_events, int mfsk_fixed_delay, int mfsk_search_raw) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    const int keep = -2147483647 - 1;
    if (nUnder_processing_events != keep) t->data_container.nUnder_processing_events = nUnder_processing_events;
    if (mfsk_fixed_delay != keep) t->mfsk_fixed_delay = mfsk_fixed_delay;
    if (mfsk_search_raw != keep) t->receive_stats.mfsk_search_raw = mfsk_search_raw;
}

// ---- the transmit side and the small members the C-ABI mirrors -------------------------------------------------------------------
double mrefts_carrier(void* h) { return static_cast<cl_telecom_system*>(h)->carrier_frequency; }

// void cl_telecom_system::transmit_byte(int* data, int nBytes, double* out, int message_location) (telecom_system.cc:343-383 -> transmit_bit
// :384-634) with the members load_configuration left (output power, PAPR cuts, the measured pre-equalisation table) and the mixer phase at
// start_sample (ofdm.passband_start_sample, ofdm.cc:2311-2313). Returns total_frame_size.
int mrefts_transmit_byte(void* h, const int* data, int nbytes, int message_location, unsigned long start_sample, double* out) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    t->ofdm.passband_start_sample = start_sample;
    t->transmit_byte(const_cast<int*>(data), nbytes, out, message_location);
    return t->data_container.total_frame_size;
}

// pre_equalization_channel as get_pre_equalization_channel (telecom_system.cc:3108-3145) measured it during load_configuration: [Nc] complex
int mrefts_pre_equalization_channel(void* h, double* out) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    for (int j = 0; j < t->data_container.Nc; j++) { out[2 * j] = t->pre_equalization_channel[j].value.real(); out[2 * j + 1] = t->pre_equalization_channel[j].value.imag(); }
    return t->data_container.Nc;
}

// generate_ack_pattern_passband (:1589-1630) / generate_break_pattern_passband; returns the samples written
int mrefts_generate_pattern(void* h, int which, unsigned long start_sample, double* out) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    t->ofdm.passband_start_sample = start_sample;
    return which == 2 ? t->generate_break_pattern_passband(out) : t->generate_ack_pattern_passband(out);
}
// detect_ack_pattern_from_passband (:1633-1680) / detect_break_pattern_from_passband on `size` passband samples
double mrefts_detect_pattern(void* h, int which, const double* data, int size, int* matched) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    return which == 2 ? t->detect_break_pattern_from_passband(const_cast<double*>(data), size, matched)
                      : t->detect_ack_pattern_from_passband(const_cast<double*>(data), size, matched);
}
// measure_signal_only (:1520-1541) on one capture window
double mrefts_measure_signal_only(void* h, const double* passband) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    return t->measure_signal_only(const_cast<double*>(passband));
}
// load_configuration(cfg) / return_to_last_configuration() on the live object; out = {current_configuration, last_configuration, Nsymb, nBits - P}
void mrefts_load_configuration(void* h, int cfg, int* out) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    if (cfg == -1) t->return_to_last_configuration(); else t->load_configuration(cfg);
    out[0] = t->current_configuration; out[1] = t->last_configuration; out[2] = t->data_container.Nsymb; out[3] = t->data_container.nBits - t->ldpc.P;
}

// data_container.passband_data_tx_buffer (3 * total_frame_size doubles), the memory of the FIRST / MIDDLE / FLUSH_MESSAGE calls (:559-596):
// set != 0 writes buf into it, else reads it out. Returns its length.
int mrefts_transmit_buffer(void* h, double* buf, int set) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    const int n = 3 * t->data_container.total_frame_size;
    if (set) memcpy(t->data_container.passband_data_tx_buffer, buf, size_t(n) * sizeof(double));
    else memcpy(buf, t->data_container.passband_data_tx_buffer, size_t(n) * sizeof(double));
    return n;
}

// One frame of the reference's own BER loop, cl_telecom_system::baseband_test_EsN0(EsN0, 1) (telecom_system.cc:95-229: random data -> encode ->
// map -> frame -> IFFT -> cl_awgn -> the RX chain without AGC, variance from the un-equalised pilots -> cl_ldpc::decode), and what it left in
// data_container: the noisy baseband frame it received and every stage's output, so that the same samples can go through the restatement.
// err = {Bits_total, Error_bits_total, Frames_total, Error_frames_total}. (One frame per call: the loop plots every tenth.)
void mrefts_baseband_test_one_frame(void* h, float esn0, double* baseband, double* grid, double* eq, double* syms, float* llr_demod,
                                    float* llr_ldpc, int* data_bits, int* decoded_bits, double* err) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    Silence s;
    const cl_error_rate e = t->baseband_test_EsN0(esn0, 1);
    const cl_data_container& d = t->data_container;
    const int nReal = d.nBits - t->ldpc.P, G = d.Nsymb * d.Nc;
    memcpy(baseband, d.baseband_data, size_t(d.Nofdm) * d.Nsymb * 16);
    memcpy(grid, d.ofdm_symbol_demodulated_data, size_t(G) * 16);
    memcpy(eq, d.equalized_data, size_t(G) * 16);
    memcpy(syms, d.ofdm_time_freq_deinterleaved_data, size_t(d.nData) * 16);
    memcpy(llr_demod, d.demodulated_data, size_t(d.nBits) * 4);
    memcpy(llr_ldpc, d.deinterleaved_data, size_t(t->ldpc.N) * 4);
    memcpy(data_bits, d.data_bit, size_t(nReal) * 4);
    memcpy(decoded_bits, d.hd_decoded_data_bit, size_t(nReal) * 4);
    err[0] = e.Bits_total; err[1] = e.Error_bits_total; err[2] = e.Frames_total; err[3] = e.Error_frames_total;
}

// char cl_telecom_system::get_configuration(double SNR) (telecom_system.cc:3036-3106)
int mrefts_get_configuration(void* h, double snr) { return static_cast<cl_telecom_system*>(h)->get_configuration(snr); }

// set_mfsk_ctrl_mode (telecom_system.cc:1572-1585): the MFSK modes' short control frames; returns get_active_nsymb()
int mrefts_set_mfsk_ctrl_mode(void* h, int enable) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    t->set_mfsk_ctrl_mode(enable != 0);
    return t->get_active_nsymb();
}

}  // extern "C"
