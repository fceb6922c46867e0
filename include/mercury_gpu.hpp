// mercury_gpu.hpp — C++ host-side mirror of the reference's physical-layer RX surface, header-only,
// over the C-ABI of mercury_gpu.h.  Same names, argument meaning and failure signalling as the
// reference classes so that source/physical_layer/telecom_system.cc can call it unchanged:
//
//   mgpu::cl_ldpc          <->  class cl_ldpc            (include/physical_layer/ldpc.h:32-93)
//        N, P, K, rate, framesize, standard, decoding_algorithm, GBF_eta, nIteration_max,
//        init(), deinit(), int decode(const float* data, int* decoded_data), void encode(const int* data, int* encoded_data)
//   mgpu::cl_rx_phy        <->  the RX members of class cl_telecom_system
//                                (include/physical_layer/telecom_system.h:85-198)
//        load_configuration(int)              telecom_system.cc:2487
//        get_frame_size_bytes()/bits()        telecom_system.cc:332-340
//        receive_frame(baseband, out)         the per-frame span of receive_byte, telecom_system.cc:1132-1345
//        receive_batch(...)                   the same over F frames (what RX_SHM would batch)
//        receive_byte(passband, out)          the whole of receive_byte, telecom_system.cc:646-1503
//        transmit_byte(data, nBytes, out, message_location)   telecom_system.cc:342-556
//        set_mfsk_ctrl_mode / get_active_nsymb / measure_signal_only / get_configuration(SNR)
//        generate_ack_pattern_passband, generate_break_pattern_passband, detect_ack_pattern_from_passband,
//        detect_break_pattern_from_passband      telecom_system.cc:1520-1716 — i.e. every cl_telecom_system call the ARQ layer makes
//   mgpu::st_receive_stats <->  struct st_receive_stats  (telecom_system.h:63-82; fields this path produces)
//
// Like the reference, failure to decode is reported through the stats (iterations_done > max-1,
// crc != 0, all_zeros, message_decoded == NO, SNR == -99.9); unlike the reference nothing exit()s:
// set-up errors throw std::runtime_error carrying mgpu_last_error().
#pragma once
#include <complex>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "mercury_gpu.h"
#include "mercury_rxloop.h"
#include "mercury_stages.h"
#include "mercury_tx.h"

namespace mgpu {

enum { GBF = 0, SPA = 1, MINSUM = 2 };   // physical_defines.h:44-45 (+ the fp32 variant)
enum { NO = 0, YES = 1 };

struct st_receive_stats {
    int iterations_done = -1;
    int message_decoded = NO;
    double SNR = -99.9;
    int crc = 0;
    int all_zeros = NO;
    float variance = 0;
    // synchroniser side (set by cl_rx_phy::receive_byte only), telecom_system.h:63-82
    int delay = 0;
    int delay_of_last_decoded_message = -1;
    int sync_trials = 0;
    double freq_offset = 0;
    double freq_offset_of_last_decoded_message = 0;
    double coarse_metric = 0;
    double signal_stregth_dbm = -999;   // (the reference's spelling) 10 log10 of the time-sync baseband's mean power in mW, telecom_system.cc:678
    int frame_overflow_symbols = 0;
    int mfsk_search_raw = 0;      // telecom_system.h:79: MFSK anti-re-decode base search position (symbols); the ARQ layer sets it
};

namespace detail {
inline void check(int rc, mgpu_ctx* ctx, const char* what) {
    if (rc != MGPU_OK) throw std::runtime_error(std::string(what) + ": " + mgpu_last_error(ctx));
}
inline st_receive_stats convert(const mgpu_frame_stats& s) {
    st_receive_stats r;
    r.iterations_done = s.iterations_done;
    r.message_decoded = s.message_decoded ? YES : NO;
    r.SNR = s.message_decoded ? double(s.snr_db) : -99.9;
    r.crc = s.crc;
    r.all_zeros = s.all_zeros ? YES : NO;
    r.variance = s.variance;
    return r;
}
// What ONE receive_byte call writes into the receive_stats member it returns, path by path (telecom_system.cc:646-1503) - everything else
// keeps the previous call's value, which is what the caller of the reference reads too (pinned against the real object run window after
// window without resets: tests/test_receive_byte_stale_fields.py):
//   message_decoded, frame_overflow_symbols, sync_trials          every call (:653-655, :710-712)
//   delay, signal_stregth_dbm                                      every call (:668-692)
//   coarse_metric                                                  every call in the OFDM modes (:693), never in the MFSK modes
//   iterations_done, all_zeros, crc, SNR                           only when a trial reached the decoder (:1310-1347, :1362-1398)
//   freq_offset                                                    only on a decoded OFDM frame (:1421-1425)
//   delay_of_last_decoded_message, freq_offset_of_last_...         through mgpu_link_state (:1423-1427)
// r.iterations_done == -1 <=> no trial reached the decoder (the decoder itself returns 0 .. nIteration_max + 1).
inline void apply_receive_byte(st_receive_stats& q, const mgpu_receive_stats& r, const mgpu_link_state& ls, bool mfsk) {
    q.message_decoded = r.message_decoded ? YES : NO; q.frame_overflow_symbols = r.frame_overflow_symbols; q.sync_trials = r.sync_trials;
    q.delay = r.delay; q.signal_stregth_dbm = r.signal_strength_dbm;
    if (!mfsk) q.coarse_metric = r.coarse_metric;
    if (r.iterations_done != -1 || r.message_decoded) {
        q.iterations_done = r.iterations_done; q.crc = r.crc; q.all_zeros = r.all_zeros; q.SNR = r.snr_db;
    }
    if (r.message_decoded && !mfsk) q.freq_offset = r.freq_offset;
    q.delay_of_last_decoded_message = ls.delay_of_last_decoded_message;
    q.freq_offset_of_last_decoded_message = ls.freq_offset_of_last_decoded_message;
}
}  // namespace detail

// ---- cl_ldpc --------------------------------------------------------------------------------
class cl_ldpc {
public:
    int N = 0, P = 0, K = 0;
    int standard = 0;                 // MERCURY
    int framesize = 1600;             // MERCURY_NORMAL
    float rate = 0;
    int decoding_algorithm = SPA;
    float GBF_eta = 0.5f;
    int nIteration_max = 50;
    int print_nIteration = NO;
    int device = 0;

    ~cl_ldpc() { deinit(); }

    // ldpc.cc:62-74: K = (int)((float)N*rate); the graph of that rate is selected (ldpc.cc:140-251)
    void init() {
        deinit();
        N = framesize;
        K = int(float(N) * rate);
        P = N - K;
        static const int rate_to_cfg[8][2] = {{100, 0}, {200, 1}, {300, 2}, {400, 3}, {500, 4}, {600, 5}, {800, 6}, {1400, 12}};
        int cfg = -1;
        for (auto& rc : rate_to_cfg) if (rc[0] == K) cfg = rc[1];
        if (cfg < 0) throw std::runtime_error("cl_ldpc::init: wrong code rate");   // the reference exits here (ldpc.cc:246-249)
        mgpu_config c{};
        c.cfg = cfg; c.max_iters = nIteration_max; c.decoder = decoding_algorithm; c.agc = 1; c.variance_source = 1;
        c.device = device; c.max_batch = 1;
        detail::check(mgpu_create(&c, &ctx_), nullptr, "cl_ldpc::init");
    }
    void deinit() {
        if (ctx_) mgpu_destroy(ctx_);
        ctx_ = nullptr;
    }
    // ldpc.h:90 — returns the number of iterations used; > nIteration_max means the word may be corrupt
    int decode(const float* data, int* decoded_data) {
        std::vector<uint8_t> bits(K);
        int iters = 0;
        detail::check(mgpu_ldpc_batch(ctx_, data, 1, bits.data(), &iters), ctx_, "cl_ldpc::decode");
        for (int i = 0; i < K; ++i) decoded_data[i] = bits[i];
        return iters;
    }
    // ldpc.h:82 — encoded_data = the K data bits followed by the P parity bits
    void encode(const int* data, int* encoded_data) {
        std::vector<uint8_t> in(K), out(N);
        for (int i = 0; i < K; ++i) in[i] = uint8_t(data[i] & 1);
        detail::check(mgpu_ldpc_encode_batch(ctx_, in.data(), 1, out.data()), ctx_, "cl_ldpc::encode");
        for (int i = 0; i < N; ++i) encoded_data[i] = out[i];
    }

private:
    mgpu_ctx* ctx_ = nullptr;
};

// ---- RX half of cl_telecom_system ---------------------------------------------------------------
class cl_rx_phy {
public:
    int current_configuration = -1;   // CONFIG_NONE
    int ldpc_nIteration_max = 50;     // cl_configuration_telecom_system default (physical_config.cc:74)
    int ldpc_decoding_algorithm = SPA;
    int max_batch = 1;
    int device = 0;
    st_receive_stats receive_stats;
    mgpu_info info{};
    // cl_telecom_system::default_configurations_telecom_system (cl_configuration_telecom_system, physical_config.cc:30-65): the values
    // load_configuration copies into the DSP objects for every mode (telecom_system.cc:2772-2811). Change them BEFORE load_configuration,
    // as the reference's callers do. ofdm_Nsymb / ofdm_pilot_configurator_Dy: -1 = AUTO_SELLECT (physical_config.cc:38-40; resolved as
    // cl_telecom_system::init does for HIGH_DENSITY pilots, telecom_system.cc:1810-1869); the reference's LOW_DENSITY option is Dy 5 with
    // Nsymb 40 / 20 / 10 for BPSK / QPSK / 16QAM. (Nc, Nfft, Dx are fixed: the kernels are specialised for the reference's 50 / 256 / 1.)
    struct {
        float ofdm_pilot_configurator_pilot_boost = 1.33f;
        int ofdm_LS_window_width = 20;            // = ofdm_LS_window_hight
        unsigned ofdm_pilot_configurator_seed = 0, bit_energy_dispersal_seed = 0, ofdm_preamble_configurator_seed = 1;
        int ofdm_Nsymb = -1, ofdm_pilot_configurator_Dy = -1;
    } default_configurations_telecom_system;

    ~cl_rx_phy() { release(); }

    // telecom_system.cc:2487 — re-initialises everything the mode owns; a no-op for the current mode
    int last_configuration = -1;      // telecom_system.h: last_configuration (CONFIG_NONE)
    void load_configuration(int configuration) {
        if (configuration == current_configuration && ctx_) return;
        if (!((configuration >= 0 && configuration <= 16) || (configuration >= 100 && configuration <= 102))) return;   // the reference returns silently too (:2494-2497)
        release();
        last_configuration = current_configuration < 0 ? configuration : current_configuration;     // telecom_system.cc:2671-2680
        current_configuration = configuration;
        mfsk_ctrl_mode = false;               // load_configuration leaves control-frame mode (telecom_system.cc:2990)
        select(0);
        // init() measures the filter chain only when the modulation or the preamble length changed (telecom_system.cc:1954-1958, :2682-2691).
        // The measurement is ~0.2 s of host work (1000 symbols through three filters), so the mirror does the same: a gear shift between
        // modes of one modulation and preamble length re-installs the table it has (every load makes a new context) instead of re-measuring.
        if (configuration < 100) {
            const auto& d = default_configurations_telecom_system;
            const PreEqKey key{info.M, info.preamble_nsymb, carrier_frequency, d.ofdm_pilot_configurator_pilot_boost, d.ofdm_pilot_configurator_seed,
                               d.ofdm_preamble_configurator_seed};
            if (key == pre_eq_key_ && int(pre_equalization_channel.size()) == info.Nc)
                detail::check(mgpu_set_pre_equalization_channel(ctx_, reinterpret_cast<const double*>(pre_equalization_channel.data())), ctx_,
                              "load_configuration");
            else { get_pre_equalization_channel(); pre_eq_key_ = key; }
        } else {
            // An MFSK mode changes the modulation, which sets the reference's sticky reinit flag (telecom_system.cc:2682-2691); init() only
            // measures and clears it in an OFDM mode (:1954-1958), so the next OFDM load re-measures whatever its modulation: drop the key.
            pre_eq_key_ = PreEqKey{};
        }
    }
    // void cl_telecom_system::return_to_last_configuration() — telecom_system.cc:3027-3034, statement for statement. Note what the reference's
    // bookkeeping amounts to: load_configuration(last) already moves current -> last and last -> current, and the swap that follows moves
    // them back, so afterwards current_configuration / last_configuration read as BEFORE the call while the mode actually loaded (info.cfg,
    // every buffer size) is the former last_configuration. The mirror reproduces that state exactly (the reference's ARQ layer keeps its own
    // bookkeeping, arq_common.cc:783-794, and never calls this member); load_configuration's "already current" test reads
    // current_configuration, as the reference's does (:2489-2492).
    void return_to_last_configuration() {
        load_configuration(last_configuration);
        const int tmp = last_configuration;
        last_configuration = current_configuration;
        current_configuration = tmp;
    }
    // void cl_telecom_system::get_pre_equalization_channel() — telecom_system.cc:3108-3145: measured for carrier_frequency and installed;
    // transmit_bit / transmit_byte multiply the preamble and data grids with it (:474-494). The OFDM modes only (:1954).
    std::vector<std::complex<double>> pre_equalization_channel;
    struct PreEqKey {
        int M = -1, preamble = -1; double carrier = 0; float boost = 0; unsigned pilot_seed = 0, preamble_seed = 0;
        bool operator==(const PreEqKey& o) const {
            return M == o.M && preamble == o.preamble && carrier == o.carrier && boost == o.boost && pilot_seed == o.pilot_seed && preamble_seed == o.preamble_seed;
        }
    } pre_eq_key_;
    void get_pre_equalization_channel() {
        pre_equalization_channel.assign(info.Nc, std::complex<double>(0, 0));
        detail::check(mgpu_context_pre_equalization_channel(ctx_, carrier_frequency, reinterpret_cast<double*>(pre_equalization_channel.data())), ctx_,
                      "get_pre_equalization_channel");
        detail::check(mgpu_set_pre_equalization_channel(ctx_, reinterpret_cast<const double*>(pre_equalization_channel.data())), ctx_,
                      "get_pre_equalization_channel");
    }
    // void cl_telecom_system::set_mfsk_ctrl_mode(bool) (telecom_system.cc:1572-1584): short control frames in the MFSK modes.
    // The frame geometry is part of a context's tables, so the mirror keeps one context per setting and switches between them.
    bool mfsk_ctrl_mode = false;
    void set_mfsk_ctrl_mode(bool enable) {
        if (current_configuration < 100) return;          // only the ROBUST modes have control frames (:1574)
        mfsk_ctrl_mode = enable;
        select(enable ? 1 : 0);
    }
    int get_active_nsymb() const { return info.active_nsymb; }                // telecom_system.h: get_active_nsymb()
    // char cl_telecom_system::get_configuration(double SNR) — telecom_system.cc:3036-3108: the fastest mode whose threshold the
    // measured SNR clears (the thresholds of common_defines.h:130-147 as that function states them)
    static char get_configuration(double SNR) {
        static const double above[15] = {12.5, 9, 7.5, 6.5, 4, 3, 1.5, 0.5, -0.5, -1.5, -2.5, -3.5, -4.5, -6, -7.5};   // CONFIG_15 .. CONFIG_1
        for (int i = 0; i < 15; ++i) if (SNR > above[i]) return char(15 - i);
        return 0;
    }
    int get_frame_size_bytes() const { return info.payload_bytes; }          // telecom_system.cc:332-335
    int get_frame_size_bits() const { return info.payload_bytes * 8; }

    // synchroniser parameters, cl_configuration_telecom_system defaults (physical_config.cc:84-87)
    double carrier_frequency = 48000.0 * 50.0 / 256 / 4 / 2 + 300;
    int time_sync_trials_max = 2;
    int use_last_good_time_sync = YES;
    int use_last_good_freq_offset = YES;
    int coarse_freq_sync_enabled = NO;      // g_gui_state.coarse_freq_sync_enabled (gui_state.h:143)
    int mfsk_fixed_delay = -1;              // telecom_system.h:110: >= 0 bypasses the time sync once (MFSK modes)
    int nUnder_processing_events = 0;       // data_container.nUnder_processing_events: symbols consumed since the last decode (:683)

    // Samples of one capture window: Nofdm * buffer_Nsymb * frequency_interpolation_rate (data_container.cc:133-143)
    int capture_window_samples() const { return mgpu_receive_buffer_nsymb(ctx_) * info.Nofdm * 4; }

    // st_receive_stats cl_telecom_system::receive_byte(double* data, int* out) — telecom_system.h:142, .cc:646-1503:
    // one passband capture window in, get_frame_size_bytes() ints out, statistics returned and kept in
    // receive_stats; the last good delay / frequency offset carry over to the next call as in the reference.
    st_receive_stats receive_byte(const double* data, int* out) { return receive_byte_samples(data, MGPU_SAMPLES_F64, out); }
    // The same on the capture as the audio device delivers it (the reference asks for INT32, audioio.c:744; its capture thread widens to double,
    // :893-936): widened on the device instead, same results (mgpu_receive_byte_batch_samples)
    st_receive_stats receive_byte(const int32_t* data, int* out) { return receive_byte_samples(data, MGPU_SAMPLES_INT32, out); }
    st_receive_stats receive_byte(const int16_t* data, int* out) { return receive_byte_samples(data, MGPU_SAMPLES_INT16, out); }
    st_receive_stats receive_byte(const float* data, int* out) { return receive_byte_samples(data, MGPU_SAMPLES_F32, out); }
    st_receive_stats receive_byte_samples(const void* data, int sample_format, int* out) {
        mgpu_receive_config rc{carrier_frequency, time_sync_trials_max, use_last_good_time_sync, use_last_good_freq_offset, coarse_freq_sync_enabled};
        // the MFSK anti-re-decode offset (telecom_system.cc:683-685) and the one-shot known delay (:663-672)
        int search_start = receive_stats.mfsk_search_raw - nUnder_processing_events;
        if (search_start < 0) search_start = 0;
        mgpu_link_state ls{receive_stats.delay_of_last_decoded_message, receive_stats.freq_offset_of_last_decoded_message, search_start,
                           (info.mfsk_M > 0 && mfsk_fixed_delay >= 0) ? mfsk_fixed_delay + 1 : 0};
        mfsk_fixed_delay = -1;
        mgpu_receive_stats r{};
        std::vector<uint8_t> bytes(info.payload_stride);
        detail::check(mgpu_receive_byte_batch_samples(ctx_, data, sample_format, 1, &rc, &ls, bytes.data(), &r), ctx_, "receive_byte");
        if (r.iterations_done != -1)                               // no decode attempted: the reference leaves out[] as it was
            for (int i = 0; i < info.payload_bytes; ++i) out[i] = bytes[i];
        detail::apply_receive_byte(receive_stats, r, ls, info.mfsk_M > 0);
        return receive_stats;
    }

    // transmit side, cl_configuration_telecom_system defaults (telecom_system.cc:69, physical_config.cc:88,115-116)
    double carrier_amplitude = 1.4142135623730951;
    double output_power_Watt = 0.1;
    double preamble_papr_cut = 7, data_papr_cut = 10;
    unsigned long passband_start_sample = 0;     // cl_ofdm::passband_start_sample: runs on from call to call (ofdm.cc:2313)
    int total_frame_size() const { return mgpu_transmit_frame_samples(ctx_); }     // data_container.cc:159

    // void cl_telecom_system::transmit_byte(int* data, int nBytes, double* out, int message_location) — telecom_system.cc:342-556.
    // `out` receives total_frame_size() samples. message_location: SINGLE_MESSAGE (3) or NO_FILTER_MESSAGE (4). A message
    // longer than the frame is not sent (the reference prints "message too long.. not sent." and returns); returns whether
    // samples were written.
    bool transmit_byte(const int* data, int nBytes, double* out, int message_location) {
        if (nBytes > info.payload_bytes || nBytes < 0) return false;
        std::vector<uint8_t> bytes(info.payload_bytes, 0);
        for (int i = 0; i < nBytes; ++i) bytes[i] = uint8_t(data[i]);
        const mgpu_transmit_config tc{carrier_frequency, carrier_amplitude, output_power_Watt, preamble_papr_cut, data_papr_cut,
                                      passband_start_sample, message_location, 0};
        detail::check(mgpu_transmit_byte_batch(ctx_, bytes.data(), info.payload_bytes, &nBytes, 1, &tc, out), ctx_, "transmit_byte");
        passband_start_sample += static_cast<unsigned long>(info.preamble_nsymb + info.active_nsymb) * info.Nofdm * 4;
        return true;
    }

    // void cl_telecom_system::transmit_bit(int* data, double* out, int message_location) — telecom_system.h:138, .cc:384-556: the
    // get_frame_size_bits() + 16 = nReal data bits as they are (transmit_byte has appended the CRC and the padding by then).
    void transmit_bit(const int* data, double* out, int message_location) {
        std::vector<uint8_t> bits(info.nReal);
        for (int i = 0; i < info.nReal; ++i) bits[i] = uint8_t(data[i] & 1);
        const mgpu_transmit_config tc{carrier_frequency, carrier_amplitude, output_power_Watt, preamble_papr_cut, data_papr_cut,
                                      passband_start_sample, message_location, 0};
        detail::check(mgpu_transmit_bit_batch(ctx_, bits.data(), 1, &tc, out), ctx_, "transmit_bit");
        passband_start_sample += static_cast<unsigned long>(info.preamble_nsymb + info.active_nsymb) * info.Nofdm * 4;
    }
    // st_receive_stats cl_telecom_system::receive_bit(double* data, int* out) — telecom_system.h:139, .cc:636-644: receive_byte into the
    // decoder's byte buffer, then byte_to_bit over nReal / 8 bytes (misc.cc:93-105, LSB first): out receives (nReal / 8) * 8 ints, the
    // de-scrambled decoded bits CRC included. (Like receive_byte, nothing is written when no decode was attempted.)
    st_receive_stats receive_bit(const double* data, int* out) {
        mgpu_receive_config rc{carrier_frequency, time_sync_trials_max, use_last_good_time_sync, use_last_good_freq_offset, coarse_freq_sync_enabled};
        int search_start = receive_stats.mfsk_search_raw - nUnder_processing_events;
        if (search_start < 0) search_start = 0;
        mgpu_link_state ls{receive_stats.delay_of_last_decoded_message, receive_stats.freq_offset_of_last_decoded_message, search_start,
                           (info.mfsk_M > 0 && mfsk_fixed_delay >= 0) ? mfsk_fixed_delay + 1 : 0};
        mfsk_fixed_delay = -1;
        mgpu_receive_stats r{};
        std::vector<uint8_t> bytes(info.payload_stride);
        detail::check(mgpu_receive_byte_batch(ctx_, data, 1, &rc, &ls, bytes.data(), &r), ctx_, "receive_bit");
        if (r.iterations_done != -1)
            for (int i = 0; i < info.nReal / 8; ++i)
                for (int j = 0; j < 8; ++j) out[i * 8 + j] = (bytes[i] >> j) & 1;
        detail::apply_receive_byte(receive_stats, r, ls, info.mfsk_M > 0);
        return receive_stats;
    }

    // One synchronised frame: `baseband` points at the first data symbol, i.e. what receive_byte passes to
    // symbol_demod (&baseband_data[Nofdm*preamble_nSymb], telecom_system.cc:1137). `out` receives
    // get_frame_size_bytes() ints, one byte each, like receive_byte's `int* out` (:1329-1332).
    st_receive_stats receive_frame(const std::complex<double>* baseband, int* out) {
        std::vector<uint8_t> bytes(info.payload_stride);
        mgpu_frame_stats s{};
        detail::check(mgpu_rx_batch(ctx_, reinterpret_cast<const double*>(baseband), 1, bytes.data(), &s, nullptr), ctx_, "receive_frame");
        for (int i = 0; i < info.payload_bytes; ++i) out[i] = bytes[i];
        receive_stats = detail::convert(s);
        return receive_stats;
    }
    // F frames laid out back to back; payload: F x payload_stride bytes.
    void receive_batch(const std::complex<double>* baseband, int F, uint8_t* payload, std::vector<st_receive_stats>& stats) {
        std::vector<mgpu_frame_stats> s(F);
        detail::check(mgpu_rx_batch(ctx_, reinterpret_cast<const double*>(baseband), F, payload, s.data(), nullptr), ctx_, "receive_batch");
        stats.resize(F);
        for (int f = 0; f < F; ++f) stats[f] = detail::convert(s[f]);
    }

    // double cl_telecom_system::measure_signal_only(double* data) — telecom_system.cc:1520-1541
    double measure_signal_only(const double* data) {
        double dbm = 0;
        detail::check(mgpu_measure_signal_only(ctx_, data, 1, carrier_frequency, &dbm), ctx_, "measure_signal_only");
        return dbm;
    }
    // cl_error_rate cl_telecom_system::baseband_test_EsN0(float EsN0, int max_frame_no) — telecom_system.h:131, .cc:96-229: the
    // self-simulation BER_PLOT_baseband_process_main calls once per Es/N0 point (:2408-2414). Frames come from the generator's
    // Philox stream (seed / first frame below), not libc rand(); the counters are cl_error_rate's (error_rate.h).
    std::uint64_t ber_seed = 1, ber_next_frame = 0;
    mgpu_error_rate baseband_test_EsN0(float EsN0, int max_frame_no) {
        mgpu_error_rate r{};
        const double e = EsN0;
        detail::check(mgpu_baseband_test_esn0(ctx_, &e, 1, max_frame_no, ber_seed, ber_next_frame, 0, &r), ctx_, "baseband_test_EsN0");
        ber_next_frame += std::uint64_t(max_frame_no);
        return r;
    }
    // cl_error_rate cl_telecom_system::passband_test_EsN0(float EsN0, int max_frame_no) — telecom_system.h:130, .cc:231-330: the audio-path
    // self-simulation BER_PLOT_passband_process_main calls once per point (:2451; it sets output_power_Watt = 1 first, :2442) —
    // transmit_byte, AWGN with delay, receive_byte — with this object's carrier_frequency and output_power_Watt; frames from the same
    // Philox stream as above.
    mgpu_error_rate passband_test_EsN0(float EsN0, int max_frame_no) {
        mgpu_error_rate r{};
        const double e = EsN0;
        detail::check(mgpu_passband_test_esn0(ctx_, &e, 1, max_frame_no, ber_seed, ber_next_frame, carrier_frequency, output_power_Watt, &r, nullptr, nullptr), ctx_,
                      "passband_test_EsN0");
        ber_next_frame += std::uint64_t(max_frame_no);
        return r;
    }
    // int generate_ack_pattern_passband(double* out) / generate_break_pattern_passband — telecom_system.cc:1589-1631, :1659-1689;
    // returns the number of samples written (ack_pattern_passband_samples)
    int ack_pattern_passband_samples() const { return 16 * info.Nofdm * 4; }
    int generate_ack_pattern_passband(double* out) { return pattern_passband(1, out); }
    int generate_break_pattern_passband(double* out) { return pattern_passband(2, out); }
    // double detect_ack_pattern_from_passband(double* data, int size, int* out_matched) / detect_break_... — :1633-1655, :1692-1716
    double detect_ack_pattern_from_passband(const double* data, int size, int* out_matched) { return detect_pattern(1, data, size, out_matched); }
    double detect_break_pattern_from_passband(const double* data, int size, int* out_matched) { return detect_pattern(2, data, size, out_matched); }

    mgpu_ctx* context() const { return ctx_; }       // for the per-method mirrors below

private:
    int pattern_passband(int which, double* out) {
        const mgpu_transmit_config tc{carrier_frequency, carrier_amplitude, output_power_Watt, preamble_papr_cut, data_papr_cut,
                                      passband_start_sample, MGPU_SINGLE_MESSAGE, 0};
        detail::check(mgpu_generate_ack_pattern_passband(ctx_, which, &tc, out), ctx_, "generate_ack_pattern_passband");
        passband_start_sample += static_cast<unsigned long>(ack_pattern_passband_samples());
        return ack_pattern_passband_samples();
    }
    double detect_pattern(int which, const double* data, int size, int* out_matched) {
        double metric = 0;
        int matched = 0;
        detail::check(mgpu_detect_ack_pattern_from_passband(ctx_, data, 1, size, carrier_frequency, which, &metric, &matched), ctx_, "detect_ack_pattern");
        if (out_matched) *out_matched = matched;
        return metric;
    }
    void select(int which) {
        if (!ctxs_[which]) {
            mgpu_config c{};
            c.cfg = current_configuration; c.max_iters = ldpc_nIteration_max; c.decoder = ldpc_decoding_algorithm;
            c.agc = 1; c.variance_source = 1; c.device = device; c.max_batch = max_batch; c.mfsk_ctrl_mode = which;
            mgpu_explicit_params x{};
            x.pilot_boost = default_configurations_telecom_system.ofdm_pilot_configurator_pilot_boost;
            x.ls_window = default_configurations_telecom_system.ofdm_LS_window_width;
            x.seeds_set = 1;
            x.pilot_seed = default_configurations_telecom_system.ofdm_pilot_configurator_seed;
            x.scrambler_seed = default_configurations_telecom_system.bit_energy_dispersal_seed;
            x.preamble_seed = default_configurations_telecom_system.ofdm_preamble_configurator_seed;
            if (current_configuration < 100) {     // the MFSK modes derive their frame length from the codeword and carry no pilots (telecom_system.cc:1812-1816, :1873-1877)
                x.Nsymb = default_configurations_telecom_system.ofdm_Nsymb > 0 ? default_configurations_telecom_system.ofdm_Nsymb : 0;
                x.Dy = default_configurations_telecom_system.ofdm_pilot_configurator_Dy > 0 ? default_configurations_telecom_system.ofdm_pilot_configurator_Dy : 0;
            }
            detail::check(mgpu_create_explicit(&c, &x, &ctxs_[which]), nullptr, "load_configuration");
        }
        ctx_ = ctxs_[which];
        detail::check(mgpu_get_info(ctx_, &info), ctx_, "load_configuration");
    }
    void release() {
        for (auto& c : ctxs_) { if (c) mgpu_destroy(c); c = nullptr; }
        ctx_ = nullptr;
    }
    mgpu_ctx* ctxs_[2] = {nullptr, nullptr};     // [0] data frames, [1] MFSK control frames
    mgpu_ctx* ctx_ = nullptr;
};

// ---- the RX methods of cl_ofdm / cl_psk and the free deinterleaver, one call each ------------------------------
// For host code that keeps receive_byte's method-by-method sequence (telecom_system.cc:1132-1298). Same names, same
// argument meaning; each call is one small GPU launch on one frame (use cl_rx_phy for throughput). The objects share the
// context of a cl_rx_phy: `mgpu::cl_ofdm ofdm(phy); ofdm.symbol_demod(in, out); ...`.
class cl_ofdm {
public:
    int Nc = 0, Nfft = 0, Nsymb = 0, Ngi = 0;
    int channel_estimator = 0, channel_estimator_amplitude_restoration = NO;
    std::vector<std::complex<double>> estimated_channel;          // st_channel_complex::value of every cell (ofdm.h:181)

    cl_ofdm(mgpu_ctx* ctx, const mgpu_info& info) : ctx_(ctx), G_(info.Nsymb * info.Nc), nData_(info.nData) {
        Nc = info.Nc; Nfft = info.Nfft; Nsymb = info.Nsymb; Ngi = info.Ngi;
        channel_estimator = info.estimator; channel_estimator_amplitude_restoration = info.amp_restore ? YES : NO;
        estimated_channel.assign(G_, {0.0, 0.0});
    }
    void symbol_demod(const std::complex<double>* in, std::complex<double>* out) {                     // ofdm.cc:862-867
        detail::check(mgpu_symbol_demod(ctx_, reinterpret_cast<const double*>(in), 1, reinterpret_cast<double*>(out)), ctx_, "symbol_demod");
    }
    void automatic_gain_control(std::complex<double>* in) {                                            // ofdm.cc:1467-1498
        detail::check(mgpu_automatic_gain_control(ctx_, reinterpret_cast<double*>(in), 1), ctx_, "automatic_gain_control");
    }
    void LS_channel_estimator(const std::complex<double>* in) { estimate(in); }                        // ofdm.cc:1315-1451
    void ZF_channel_estimator(const std::complex<double>* in) { estimate(in); }                        // ofdm.cc:1266-1313
    void restore_channel_amplitude() {                                                                 // ofdm.cc:1453-1466
        detail::check(mgpu_restore_channel_amplitude(ctx_, reinterpret_cast<double*>(estimated_channel.data()), 1), ctx_, "restore_channel_amplitude");
    }
    void channel_equalizer(const std::complex<double>* in, std::complex<double>* out) {                // ofdm.cc:1637-1647
        detail::check(mgpu_channel_equalizer(ctx_, reinterpret_cast<const double*>(in), reinterpret_cast<const double*>(estimated_channel.data()), 1,
                                             reinterpret_cast<double*>(out)), ctx_, "channel_equalizer");
    }
    double measure_variance(const std::complex<double>* in) {                                          // ofdm.cc:1500-1521
        double v = 0;
        detail::check(mgpu_measure_variance(ctx_, reinterpret_cast<const double*>(in), 1, &v), ctx_, "measure_variance");
        return v;
    }
    void deframer(const std::complex<double>* in, std::complex<double>* out) {                         // ofdm.cc:837-852
        detail::check(mgpu_deframer(ctx_, reinterpret_cast<const double*>(in), 1, reinterpret_cast<double*>(out)), ctx_, "deframer");
    }

private:
    void estimate(const std::complex<double>* in) {
        detail::check(mgpu_channel_estimator(ctx_, reinterpret_cast<const double*>(in), 1, reinterpret_cast<double*>(estimated_channel.data())), ctx_,
                      "channel_estimator");
    }
    mgpu_ctx* ctx_;
    int G_, nData_;
};

class cl_psk {
public:
    cl_psk(mgpu_ctx* ctx, const mgpu_info& info) : ctx_(ctx), nBits_(info.nBits) {}
    // psk.h:55 — nItems is the number of BITS, as in the reference's call (telecom_system.cc:1296)
    void demod(const std::complex<double>* in, int nItems, float* out, float variance) {
        if (nItems != nBits_) throw std::runtime_error("cl_psk::demod: nItems must be the mode's nBits");
        detail::check(mgpu_psk_demod(ctx_, reinterpret_cast<const double*>(in), 1, &variance, out), ctx_, "cl_psk::demod");
    }

private:
    mgpu_ctx* ctx_;
    int nBits_;
};

// interleaver.h:28-34
inline void deinterleaver(mgpu_ctx* ctx, const std::complex<double>* in, std::complex<double>* out, int nItems, int block_size) {
    detail::check(mgpu_deinterleaver_c128(ctx, reinterpret_cast<const double*>(in), 1, nItems, block_size, reinterpret_cast<double*>(out)), ctx, "deinterleaver");
}
inline void deinterleaver(mgpu_ctx* ctx, const float* in, float* out, int nItems, int block_size) {
    detail::check(mgpu_deinterleaver_f32(ctx, in, 1, nItems, block_size, out), ctx, "deinterleaver");
}

// the integer tail, reference signatures (one int per bit / byte): interleaver.cc:111-117, misc.cc:107-130, crc16_modbus_rtu.cc:25-45.
// `seq` is accepted for signature compatibility; the mode's own dispersal sequence (telecom_system.cc:1961-1966) is applied.
inline void bit_energy_dispersal(mgpu_ctx* ctx, const int* in, const int* /*seq*/, int* out, int nItems) {
    std::vector<uint8_t> a(nItems), b(nItems);
    for (int i = 0; i < nItems; ++i) a[i] = uint8_t(in[i] & 1);
    detail::check(mgpu_bit_energy_dispersal(ctx, a.data(), 1, nItems, b.data()), ctx, "bit_energy_dispersal");
    for (int i = 0; i < nItems; ++i) out[i] = b[i];
}
inline void bit_to_byte(mgpu_ctx* ctx, const int* in, int* out, int nItems) {
    std::vector<uint8_t> a(nItems), b((nItems + 7) / 8);
    for (int i = 0; i < nItems; ++i) a[i] = uint8_t(in[i] & 1);
    detail::check(mgpu_bit_to_byte(ctx, a.data(), 1, nItems, b.data()), ctx, "bit_to_byte");
    for (size_t i = 0; i < b.size(); ++i) out[i] = b[i];
}
inline uint16_t CRC16_MODBUS_RTU_calc(mgpu_ctx* ctx, const int* data, int nItems) {
    std::vector<uint8_t> a(nItems);
    for (int i = 0; i < nItems; ++i) a[i] = uint8_t(data[i] & 0xff);
    uint16_t crc = 0;
    detail::check(mgpu_crc16_modbus_rtu(ctx, a.data(), 1, nItems, &crc), ctx, "CRC16_MODBUS_RTU_calc");
    return crc;
}

}  // namespace mgpu
