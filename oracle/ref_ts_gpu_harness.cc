// TEST INFRASTRUCTURE. The drop-in, executed from the reference's side ("Next round" item 1; INTEGRATION.md sections
// 1.2 / 1.3b in compiled form).
//
// oracle/_ref/libmercury_ref_ts_gpu.so holds the SAME unmodified reference objects as libmercury_ref_ts.so (telecom_system.cc, ofdm.cc,
// psk.cc, ldpc.cc, ... compiled from /root/reference by oracle/Makefile), except that oracle/interpose.sh has made the methods SURVEY.md
// section 8b lists WEAK in the objects that define them and has given the replaced machine code a second name (mref_orig_*). This file
// defines those methods again - as members of the reference's own classes, from the reference's own headers - and forwards each to the
// C-ABI of mercury_amd/libmercury_gpu.so (include/mercury_stages.h, mercury_gpu.h, mercury_rxloop.h), writing the members the original
// writes (estimated_channel[].value / .status, the receive_stats fields). The reference's callers - baseband_test_EsN0
// (telecom_system.cc:95-229), the hot span of receive_byte (:1132-1345), get_pre_equalization_channel (:3108-3145), RX_RAND_process_main
// (:2102-2190), receive_bit (:636-644) - are NOT touched and NOT recompiled differently: their object code calls the symbol, the linker
// binds it here. No reference text is stored in this repository and no reference source is patched.
//
// Per cl_telecom_system object a binding chooses what the replaced methods do (mreftsgpu_create / mreftsgpu_set_mode):
//   bit 0 (1)  STAGES   the per-method entry points run on the GPU (mercury_stages.h, mgpu_ldpc_batch)
//   bit 1 (2)  SHADOW   with STAGES: the original machine code runs as well, on copies, and every output is compared bit for bit
//                       (counters per method: calls, calls on the GPU, calls whose outputs differed)
//   bit 2 (4)  WHOLE    cl_telecom_system::receive_byte as a whole is mgpu_receive_byte_batch (mercury_rxloop.h) with W = 1
//   bit 3 (8)  MIRROR   cl_telecom_system::receive_byte as a whole is mgpu::cl_rx_phy::receive_byte (include/mercury_gpu.hpp, the product's C++
//                       mirror, through oracle/ref_ts_gpu_mirror.cc) running on its OWN receive_stats from call to call
//   0                   everything falls through to the original code: the object behaves as libmercury_ref_ts.so's
// A method called on an object that has no binding (another cl_psk inside cl_ofdm, a second cl_telecom_system) runs the original.
// The product never links this file; it links the product.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <complex>
#include <unistd.h>
#include <fcntl.h>

#include "physical_layer/telecom_system.h"
#include "gui/gui_state.h"

#include "../include/mercury_gpu.h"
#include "../include/mercury_stages.h"
#include "../include/mercury_rxloop.h"

typedef std::complex<double> cd;

// ---- the replaced machine code under its second name (oracle/interpose.sh); `this` travels as the first argument (Itanium C++ ABI) ----
extern "C" {
void mref_orig_ofdm_symbol_demod(cl_ofdm*, cd*, cd*);
void mref_orig_ofdm_deframer(cl_ofdm*, cd*, cd*);
void mref_orig_ofdm_ZF_channel_estimator(cl_ofdm*, cd*);
void mref_orig_ofdm_LS_channel_estimator(cl_ofdm*, cd*);
void mref_orig_ofdm_restore_channel_amplitude(cl_ofdm*);
void mref_orig_ofdm_automatic_gain_control(cl_ofdm*, cd*);
double mref_orig_ofdm_measure_variance(cl_ofdm*, cd*);
void mref_orig_ofdm_channel_equalizer(cl_ofdm*, cd*, cd*);
void mref_orig_ofdm_channel_equalizer_without_amplitude_restoration(cl_ofdm*, cd*, cd*);
void mref_orig_psk_demod(cl_psk*, const cd*, int, float*, float);
int mref_orig_ldpc_decode(cl_ldpc*, const float*, int*);
void mref_orig_deinterleaver_f32(float*, float*, int, int);
void mref_orig_deinterleaver_c128(cd*, cd*, int, int);
void mref_orig_bit_energy_dispersal(int*, int*, int*, int);
void mref_orig_bit_to_byte(int*, int*, int);
uint16_t mref_orig_crc16(int*, int);
// st_receive_stats is returned in memory: the hidden result pointer is the first argument, `this` the second
void mref_orig_ts_receive_byte(st_receive_stats*, cl_telecom_system*, double*, int*);
}

// oracle/ref_ts_gpu_mirror.cc
extern "C" {
struct mmirror_stats {
    int iterations_done, delay, delay_of_last_decoded_message, sync_trials, message_decoded, crc, all_zeros, mfsk_search_raw, frame_overflow_symbols;
    double freq_offset, freq_offset_of_last_decoded_message, SNR, signal_stregth_dbm, coarse_metric;
};
void* mmirror_create(int cfg, int max_iters);
void mmirror_destroy(void* h);
int mmirror_receive_byte(void* h, const double* data, int* out, double carrier_hz, int time_sync_trials_max, int use_last_good_time_sync,
                         int use_last_good_freq_offset, int coarse_freq_sync_enabled, int ctrl_mode, int nUnder_processing_events, int* mfsk_fixed_delay,
                         int delay_of_last_decoded_message, double freq_offset_of_last_decoded_message, int mfsk_search_raw, mmirror_stats* held);
}

namespace {

enum Method { M_SYMBOL_DEMOD, M_AGC, M_ESTIMATOR, M_RESTORE_AMPLITUDE, M_EQUALIZER, M_EQUALIZER_WAR, M_VARIANCE, M_DEFRAMER, M_DEINT_C128,
              M_DEINT_F32, M_PSK_DEMOD, M_LDPC_DECODE, M_DISPERSAL, M_BIT_TO_BYTE, M_CRC16, M_RECEIVE_BYTE, N_METHODS };
enum { STAGES = 1, SHADOW = 2, WHOLE = 4, MIRROR = 8 };

struct Binding {
    cl_telecom_system* ts = nullptr;
    mgpu_ctx* ctx = nullptr;        // the mode's context (receive_byte semantics: agc = 1, variance on the equalised grid)
    void* mirror = nullptr;         // mode MIRROR: an mgpu::cl_rx_phy (ref_ts_gpu_mirror.cc)
    mgpu_ctx* ctx_ctrl = nullptr;   // ROBUST_0 / ROBUST_1: the short control frames' context (set_mfsk_ctrl_mode), made on first use
    mgpu_info info{};
    int cfg = -1, mode = 0;
    long calls[N_METHODS] = {}, gpu[N_METHODS] = {}, differ[N_METHODS] = {};
    char error[256] = {};
};

std::vector<Binding*> g_bindings;
Binding* g_current = nullptr;        // the free functions (deinterleaver, bit_to_byte, ...) carry no object: the binding used last

Binding* binding_of(const void* member, size_t offset) {
    for (Binding* b : g_bindings)
        if (reinterpret_cast<const char*>(b->ts) + offset == reinterpret_cast<const char*>(member)) return g_current = b;
    return nullptr;
}
#define BOUND(member) binding_of(this, offsetof(cl_telecom_system, member))

bool fail(Binding* b, Method m, int rc) {
    if (rc == MGPU_OK) return false;
    snprintf(b->error, sizeof b->error, "method %d: rc %d: %s", int(m), rc, mgpu_last_error(b->ctx));
    fprintf(stderr, "[ref_ts_gpu] %s\n", b->error);
    return true;
}
// counts the call; true when the binding asks for the GPU (the caller may still decline: a method the C-ABI has no stage for in this mode)
bool on_gpu(Binding* b, Method m, bool supported = true) {
    if (!b) return false;
    b->calls[m]++;
    if (!(b->mode & STAGES) || !b->ctx || !supported) return false;
    b->gpu[m]++;
    return true;
}
void compare(Binding* b, Method m, const void* x, const void* y, size_t bytes) {
    if (memcmp(x, y, bytes) != 0) b->differ[m]++;
}
bool ofdm_mode(const Binding* b) { return b->info.mfsk_M == 0; }

struct Silence {   // the reference prints diagnostics from inside the replaced methods (ofdm.cc:1484) and around them
    int saved;
    Silence() { fflush(stdout); saved = dup(1); const int nul = open("/dev/null", O_WRONLY); dup2(nul, 1); close(nul); }
    ~Silence() { fflush(stdout); dup2(saved, 1); close(saved); }
};

}  // namespace

// =====================================================================================================================================
// cl_ofdm — include/physical_layer/ofdm.h:133-146
// =====================================================================================================================================

// ofdm.cc:862-867 (gi_remover + fft + zero_depadder): one OFDM symbol, Nofdm samples in, Nc carriers out
void cl_ofdm::symbol_demod(cd* in, cd* out) {
    Binding* b = BOUND(ofdm);
    if (!on_gpu(b, M_SYMBOL_DEMOD)) { mref_orig_ofdm_symbol_demod(this, in, out); return; }
    if (fail(b, M_SYMBOL_DEMOD, mgpu_symbol_demod(b->ctx, reinterpret_cast<const double*>(in), 1, reinterpret_cast<double*>(out)))) return;
    if (b->mode & SHADOW) {
        std::vector<cd> ref(Nc);
        mref_orig_ofdm_symbol_demod(this, in, ref.data());
        compare(b, M_SYMBOL_DEMOD, ref.data(), out, size_t(Nc) * 16);
    }
}

// ofdm.cc:1467-1498: in place over the Nsymb x Nc grid
void cl_ofdm::automatic_gain_control(cd* in) {
    Binding* b = BOUND(ofdm);
    if (!on_gpu(b, M_AGC, b && ofdm_mode(b))) { mref_orig_ofdm_automatic_gain_control(this, in); return; }
    const int G = Nsymb * Nc;
    std::vector<cd> ref;
    if (b->mode & SHADOW) { ref.assign(in, in + G); mref_orig_ofdm_automatic_gain_control(this, ref.data()); }
    if (fail(b, M_AGC, mgpu_automatic_gain_control(b->ctx, reinterpret_cast<double*>(in), 1))) return;
    if (b->mode & SHADOW) compare(b, M_AGC, ref.data(), in, size_t(G) * 16);
}

namespace {
// LS_channel_estimator (ofdm.cc:1315-1451) / ZF_channel_estimator (:1266-1313): estimated_channel[].value over the whole grid, .status MEASURED
// at the pilots and INTERPOLATED elsewhere (interpolator.cc:101,130,158)
void estimator(cl_ofdm* o, cd* in, bool ls) {
    Binding* b = binding_of(o, offsetof(cl_telecom_system, ofdm));
    const bool mode_is_ls = o->channel_estimator == LEAST_SQUARE;
    if (!on_gpu(b, M_ESTIMATOR, b && ofdm_mode(b) && ls == mode_is_ls)) {      // mgpu_channel_estimator runs the estimator the mode uses
        if (ls) mref_orig_ofdm_LS_channel_estimator(o, in); else mref_orig_ofdm_ZF_channel_estimator(o, in);
        return;
    }
    const int G = o->Nsymb * o->Nc;
    std::vector<st_channel_complex> ref;
    if (b->mode & SHADOW) {
        if (ls) mref_orig_ofdm_LS_channel_estimator(o, in); else mref_orig_ofdm_ZF_channel_estimator(o, in);
        ref.assign(o->estimated_channel, o->estimated_channel + G);
    }
    std::vector<cd> H(G);
    if (fail(b, M_ESTIMATOR, mgpu_channel_estimator(b->ctx, reinterpret_cast<const double*>(in), 1, reinterpret_cast<double*>(H.data())))) return;
    for (int i = 0; i < G; i++) {
        o->estimated_channel[i].value = H[i];
        o->estimated_channel[i].status = o->ofdm_frame[i].type == PILOT ? MEASURED : INTERPOLATED;
    }
    if (b->mode & SHADOW) {
        bool same = true;
        for (int i = 0; i < G && same; i++)
            same = memcmp(&ref[i].value, &o->estimated_channel[i].value, 16) == 0 && ref[i].status == o->estimated_channel[i].status;
        if (!same) b->differ[M_ESTIMATOR]++;
    }
}
}  // namespace
void cl_ofdm::LS_channel_estimator(cd* in) { estimator(this, in, true); }
void cl_ofdm::ZF_channel_estimator(cd* in) { estimator(this, in, false); }

// ofdm.cc:1453-1466: estimated_channel is saved (value and status) and replaced by unit-amplitude values of the same phase
void cl_ofdm::restore_channel_amplitude() {
    Binding* b = BOUND(ofdm);
    if (!on_gpu(b, M_RESTORE_AMPLITUDE, b && ofdm_mode(b))) { mref_orig_ofdm_restore_channel_amplitude(this); return; }
    const int G = Nsymb * Nc;
    std::vector<st_channel_complex> before(estimated_channel, estimated_channel + G), ref;
    if (b->mode & SHADOW) {
        mref_orig_ofdm_restore_channel_amplitude(this);
        ref.assign(estimated_channel, estimated_channel + G);
    }
    std::vector<cd> H(G);
    for (int i = 0; i < G; i++) { H[i] = before[i].value; estimated_channel_without_amplitude_restoration[i] = before[i]; }
    if (fail(b, M_RESTORE_AMPLITUDE, mgpu_restore_channel_amplitude(b->ctx, reinterpret_cast<double*>(H.data()), 1))) return;
    for (int i = 0; i < G; i++) { estimated_channel[i].value = H[i]; estimated_channel[i].status = before[i].status; }
    if (b->mode & SHADOW) {
        bool same = true;
        for (int i = 0; i < G && same; i++)
            same = memcmp(&ref[i].value, &estimated_channel[i].value, 16) == 0 && ref[i].status == estimated_channel[i].status;
        if (!same) b->differ[M_RESTORE_AMPLITUDE]++;
    }
}

namespace {
void equalizer(cl_ofdm* o, cd* in, cd* out, bool war) {
    const Method m = war ? M_EQUALIZER_WAR : M_EQUALIZER;
    Binding* b = binding_of(o, offsetof(cl_telecom_system, ofdm));
    st_channel_complex* ch = war ? o->estimated_channel_without_amplitude_restoration : o->estimated_channel;
    if (!on_gpu(b, m, b && ofdm_mode(b))) {
        if (war) mref_orig_ofdm_channel_equalizer_without_amplitude_restoration(o, in, out); else mref_orig_ofdm_channel_equalizer(o, in, out);
        return;
    }
    const int G = o->Nsymb * o->Nc;
    std::vector<cd> H(G), ref;
    for (int i = 0; i < G; i++) H[i] = ch[i].value;
    if (b->mode & SHADOW) {
        ref.resize(G);
        if (war) mref_orig_ofdm_channel_equalizer_without_amplitude_restoration(o, in, ref.data()); else mref_orig_ofdm_channel_equalizer(o, in, ref.data());
    }
    if (fail(b, m, mgpu_channel_equalizer(b->ctx, reinterpret_cast<const double*>(in), reinterpret_cast<const double*>(H.data()), 1,
                                          reinterpret_cast<double*>(out)))) return;
    if (!war) for (int i = 0; i < G; i++) ch[i].status = UNKNOWN;            // ofdm.cc:1644
    if (b->mode & SHADOW) compare(b, m, ref.data(), out, size_t(G) * 16);
}
}  // namespace
void cl_ofdm::channel_equalizer(cd* in, cd* out) { equalizer(this, in, out, false); }
void cl_ofdm::channel_equalizer_without_amplitude_restoration(cd* in, cd* out) { equalizer(this, in, out, true); }

// ofdm.cc:1500-1521
double cl_ofdm::measure_variance(cd* in) {
    Binding* b = BOUND(ofdm);
    if (!on_gpu(b, M_VARIANCE, b && ofdm_mode(b))) return mref_orig_ofdm_measure_variance(this, in);
    double v = 0;
    if (fail(b, M_VARIANCE, mgpu_measure_variance(b->ctx, reinterpret_cast<const double*>(in), 1, &v))) return 0;
    if (b->mode & SHADOW) { const double ref = mref_orig_ofdm_measure_variance(this, in); compare(b, M_VARIANCE, &ref, &v, 8); }
    return v;
}

// ofdm.cc:837-852
void cl_ofdm::deframer(cd* in, cd* out) {
    Binding* b = BOUND(ofdm);
    if (!on_gpu(b, M_DEFRAMER, b && ofdm_mode(b))) { mref_orig_ofdm_deframer(this, in, out); return; }
    if (fail(b, M_DEFRAMER, mgpu_deframer(b->ctx, reinterpret_cast<const double*>(in), 1, reinterpret_cast<double*>(out)))) return;
    if (b->mode & SHADOW) {
        std::vector<cd> ref(b->info.nData);
        mref_orig_ofdm_deframer(this, in, ref.data());
        compare(b, M_DEFRAMER, ref.data(), out, size_t(b->info.nData) * 16);
    }
}

// =====================================================================================================================================
// free functions — interleaver.h:28-36, misc.h, crc16_modbus_rtu.h. They carry no object: the binding that was used last serves them
// =====================================================================================================================================

void deinterleaver(cd* in, cd* out, int nItems, int block_size) {
    Binding* b = g_current;
    if (!on_gpu(b, M_DEINT_C128)) { mref_orig_deinterleaver_c128(in, out, nItems, block_size); return; }
    if (fail(b, M_DEINT_C128, mgpu_deinterleaver_c128(b->ctx, reinterpret_cast<const double*>(in), 1, nItems, block_size, reinterpret_cast<double*>(out)))) return;
    if (b->mode & SHADOW) {
        std::vector<cd> ref(nItems);
        mref_orig_deinterleaver_c128(in, ref.data(), nItems, block_size);
        compare(b, M_DEINT_C128, ref.data(), out, size_t(nItems) * 16);
    }
}
void deinterleaver(float* in, float* out, int nItems, int block_size) {
    Binding* b = g_current;
    if (!on_gpu(b, M_DEINT_F32)) { mref_orig_deinterleaver_f32(in, out, nItems, block_size); return; }
    if (fail(b, M_DEINT_F32, mgpu_deinterleaver_f32(b->ctx, in, 1, nItems, block_size, out))) return;
    if (b->mode & SHADOW) {
        std::vector<float> ref(nItems);
        mref_orig_deinterleaver_f32(in, ref.data(), nItems, block_size);
        compare(b, M_DEINT_F32, ref.data(), out, size_t(nItems) * 4);
    }
}

// interleaver.cc:111-117. The C-ABI applies the mode's own sequence (data_container.bit_energy_dispersal_sequence, telecom_system.cc:1961-1966);
// a call with another sequence keeps the original. in and out may be the same array (telecom_system.cc:1313).
void bit_energy_dispersal(int* in, int* sequence, int* out, int nItems) {
    Binding* b = g_current;
    if (!on_gpu(b, M_DISPERSAL, b && sequence == b->ts->data_container.bit_energy_dispersal_sequence)) {
        mref_orig_bit_energy_dispersal(in, sequence, out, nItems);
        return;
    }
    std::vector<uint8_t> bits(nItems), res(nItems);
    std::vector<int> ref;
    for (int i = 0; i < nItems; i++) bits[i] = uint8_t(in[i]);
    if (b->mode & SHADOW) { ref.resize(nItems); mref_orig_bit_energy_dispersal(in, sequence, ref.data(), nItems); }
    if (fail(b, M_DISPERSAL, mgpu_bit_energy_dispersal(b->ctx, bits.data(), 1, nItems, res.data()))) return;
    for (int i = 0; i < nItems; i++) out[i] = res[i];
    if (b->mode & SHADOW) compare(b, M_DISPERSAL, ref.data(), out, size_t(nItems) * 4);
}

// misc.cc:107-130
void bit_to_byte(int* data_bit, int* data_byte, int nBits) {
    Binding* b = g_current;
    if (!on_gpu(b, M_BIT_TO_BYTE)) { mref_orig_bit_to_byte(data_bit, data_byte, nBits); return; }
    const int nbytes = (nBits + 7) / 8;
    std::vector<uint8_t> bits(nBits), bytes(nbytes);
    for (int i = 0; i < nBits; i++) bits[i] = uint8_t(data_bit[i]);
    if (fail(b, M_BIT_TO_BYTE, mgpu_bit_to_byte(b->ctx, bits.data(), 1, nBits, bytes.data()))) return;
    std::vector<int> ref;
    if (b->mode & SHADOW) { ref.resize(nbytes); mref_orig_bit_to_byte(data_bit, ref.data(), nBits); }
    for (int i = 0; i < nbytes; i++) data_byte[i] = bytes[i];
    if (b->mode & SHADOW) compare(b, M_BIT_TO_BYTE, ref.data(), data_byte, size_t(nbytes) * 4);
}

// crc16_modbus_rtu.cc:25-45
uint16_t CRC16_MODBUS_RTU_calc(int* data_byte, int nItems) {
    Binding* b = g_current;
    if (!on_gpu(b, M_CRC16)) return mref_orig_crc16(data_byte, nItems);
    std::vector<uint8_t> bytes(nItems);
    for (int i = 0; i < nItems; i++) bytes[i] = uint8_t(data_byte[i] & 0xFF);
    uint16_t crc = 0;
    if (fail(b, M_CRC16, mgpu_crc16_modbus_rtu(b->ctx, bytes.data(), 1, nItems, &crc))) return 0;
    if (b->mode & SHADOW) { const uint16_t ref = mref_orig_crc16(data_byte, nItems); compare(b, M_CRC16, &ref, &crc, 2); }
    return crc;
}

// =====================================================================================================================================
// cl_psk::demod — psk.h:55, psk.cc:278-326.  cl_ldpc::decode — ldpc.h:90, ldpc.cc:266-278
// =====================================================================================================================================

void cl_psk::demod(const cd* in, int nItems, float* out, float variance) {
    Binding* b = BOUND(psk);
    if (!on_gpu(b, M_PSK_DEMOD, b && ofdm_mode(b) && nItems == b->info.nBits)) {
        mref_orig_psk_demod(this, in, nItems, out, variance);
        return;
    }
    if (fail(b, M_PSK_DEMOD, mgpu_psk_demod(b->ctx, reinterpret_cast<const double*>(in), 1, &variance, out))) return;
    if (b->mode & SHADOW) {
        std::vector<float> ref(nItems);
        mref_orig_psk_demod(this, in, nItems, ref.data(), variance);
        compare(b, M_PSK_DEMOD, ref.data(), out, size_t(nItems) * 4);
    }
}

int cl_ldpc::decode(const float* data, int* decoded_data) {
    Binding* b = BOUND(ldpc);
    if (!on_gpu(b, M_LDPC_DECODE)) return mref_orig_ldpc_decode(this, data, decoded_data);
    std::vector<uint8_t> bits(K);
    int iters = 0;
    if (fail(b, M_LDPC_DECODE, mgpu_ldpc_batch(b->ctx, data, 1, bits.data(), &iters))) return nIteration_max + 1;
    for (int i = 0; i < K; i++) decoded_data[i] = bits[i];                   // ldpc_decoder_SPA.cc:211-214: K hard decisions, one int each
    if (b->mode & SHADOW) {
        std::vector<int> ref(N);
        const int ref_iters = mref_orig_ldpc_decode(this, data, ref.data());
        if (ref_iters != iters || memcmp(ref.data(), decoded_data, size_t(K) * 4) != 0) b->differ[M_LDPC_DECODE]++;
    }
    return iters;
}

// =====================================================================================================================================
// cl_telecom_system::receive_byte as a whole — telecom_system.h:142, .cc:646-1503
// =====================================================================================================================================

st_receive_stats cl_telecom_system::receive_byte(double* data, int* out) {
    Binding* b = binding_of(this, 0);
    if (b) b->calls[M_RECEIVE_BYTE]++;
    if (b && (b->mode & MIRROR) && b->mirror) {
        b->gpu[M_RECEIVE_BYTE]++;
        mmirror_stats q{};
        int fixed = mfsk_fixed_delay;
        if (mmirror_receive_byte(b->mirror, data, out, carrier_frequency, time_sync_trials_max, use_last_good_time_sync, use_last_good_freq_offset,
                                 g_gui_state.coarse_freq_sync_enabled.load() ? 1 : 0, (mfsk_ctrl_mode && M == MOD_MFSK && ctrl_nsymb > 0) ? 1 : 0,
                                 data_container.nUnder_processing_events, &fixed, receive_stats.delay_of_last_decoded_message,
                                 receive_stats.freq_offset_of_last_decoded_message, receive_stats.mfsk_search_raw, &q) != 0) {
            snprintf(b->error, sizeof b->error, "mirror receive_byte failed");
            return receive_stats;
        }
        mfsk_fixed_delay = fixed;
        receive_stats.iterations_done = q.iterations_done; receive_stats.delay = q.delay;
        receive_stats.delay_of_last_decoded_message = q.delay_of_last_decoded_message; receive_stats.sync_trials = q.sync_trials;
        receive_stats.message_decoded = q.message_decoded; receive_stats.crc = q.crc; receive_stats.all_zeros = q.all_zeros;
        receive_stats.frame_overflow_symbols = q.frame_overflow_symbols; receive_stats.freq_offset = q.freq_offset;
        receive_stats.freq_offset_of_last_decoded_message = q.freq_offset_of_last_decoded_message; receive_stats.SNR = q.SNR;
        receive_stats.signal_stregth_dbm = q.signal_stregth_dbm; receive_stats.coarse_metric = q.coarse_metric;
        return receive_stats;
    }
    if (!b || !(b->mode & WHOLE) || !b->ctx) {
        st_receive_stats r;
        mref_orig_ts_receive_byte(&r, this, data, out);
        return r;
    }
    b->gpu[M_RECEIVE_BYTE]++;
    mgpu_ctx* ctx = b->ctx;
    if (mfsk_ctrl_mode && M == MOD_MFSK && ctrl_nsymb > 0) {                 // set_mfsk_ctrl_mode(true): the short control frames (:1572-1585)
        if (!b->ctx_ctrl) {
            mgpu_config gc{};
            gc.cfg = b->cfg; gc.max_iters = ldpc.nIteration_max; gc.decoder = MGPU_DEC_SPA; gc.agc = 1; gc.variance_source = 1; gc.max_batch = 1;
            gc.mfsk_ctrl_mode = 1;
            if (mgpu_create(&gc, &b->ctx_ctrl) != MGPU_OK) { fprintf(stderr, "[ref_ts_gpu] ctrl context: %s\n", mgpu_last_error(NULL)); return receive_stats; }
        }
        ctx = b->ctx_ctrl;
    }
    mgpu_receive_config rc{carrier_frequency, time_sync_trials_max, use_last_good_time_sync, use_last_good_freq_offset,
                           g_gui_state.coarse_freq_sync_enabled.load() ? 1 : 0};
    int search_start = receive_stats.mfsk_search_raw - data_container.nUnder_processing_events;      // :683-685
    if (search_start < 0) search_start = 0;
    const bool mfsk_mode = M == MOD_MFSK;
    mgpu_link_state ls{receive_stats.delay_of_last_decoded_message, receive_stats.freq_offset_of_last_decoded_message, search_start,
                       (mfsk_mode && mfsk_fixed_delay >= 0) ? mfsk_fixed_delay + 1 : 0};
    mgpu_receive_stats r{};
    mgpu_info info{};
    mgpu_get_info(ctx, &info);
    std::vector<uint8_t> bytes(info.payload_stride);
    const int rcode = mgpu_receive_byte_batch(ctx, data, 1, &rc, &ls, bytes.data(), &r);
    if (rcode != MGPU_OK) {
        snprintf(b->error, sizeof b->error, "receive_byte: rc %d: %s", rcode, mgpu_last_error(ctx));
        fprintf(stderr, "[ref_ts_gpu] %s\n", b->error);
        return receive_stats;
    }
    if (mfsk_mode) mfsk_fixed_delay = -1;                                     // used once (:663-672)
    // the members receive_byte writes, on the paths that write them (the same table as mgpu::detail::apply_receive_byte;
    // pinned by tests/test_receive_byte_stale_fields.py)
    if (r.iterations_done != -1)
        for (int i = 0; i < info.payload_bytes; i++) out[i] = bytes[i];                                              // :1329-1332
    receive_stats.message_decoded = r.message_decoded; receive_stats.frame_overflow_symbols = r.frame_overflow_symbols;   // :653-655, :710-712
    receive_stats.sync_trials = r.sync_trials;
    receive_stats.delay = r.delay; receive_stats.signal_stregth_dbm = r.signal_strength_dbm;                          // :668-692
    if (!mfsk_mode) receive_stats.coarse_metric = r.coarse_metric;                                                    // :693
    if (r.iterations_done != -1 || r.message_decoded) {                                                               // a trial reached the decoder
        receive_stats.iterations_done = r.iterations_done; receive_stats.crc = r.crc; receive_stats.all_zeros = r.all_zeros;   // :1310-1341
        receive_stats.SNR = r.snr_db;                                                                                 // :1347, :1362-1398
    }
    if (r.message_decoded && !mfsk_mode) receive_stats.freq_offset = r.freq_offset;                                   // :1421-1425
    receive_stats.delay_of_last_decoded_message = ls.delay_of_last_decoded_message;
    receive_stats.freq_offset_of_last_decoded_message = ls.freq_offset_of_last_decoded_message;
    return receive_stats;
}

// =====================================================================================================================================
// C entry points for the tests (on top of ref_ts_harness.cc's mrefts_*, which is linked into this library as well)
// =====================================================================================================================================
extern "C" {

void* mrefts_create(int cfg);       // ref_ts_harness.cc
void mrefts_destroy(void* h);

// A cl_telecom_system whose replaced methods are bound to a GPU context from the first call on: the context is made BEFORE
// load_configuration runs, so that the symbol_demod inside get_pre_equalization_channel (telecom_system.cc:3132) is served by it too.
void* mreftsgpu_create(int cfg, int mode, int max_iters) {
    Silence s;
    cl_telecom_system* t = new cl_telecom_system();
    Binding* b = new Binding();
    b->ts = t; b->cfg = cfg; b->mode = mode;
    if (mode != 0) {
        mgpu_config gc{};
        gc.cfg = cfg; gc.max_iters = max_iters > 0 ? max_iters : 50; gc.decoder = MGPU_DEC_SPA; gc.agc = 1; gc.variance_source = 1; gc.device = 0;
        gc.max_batch = 1;
        if (mgpu_create(&gc, &b->ctx) != MGPU_OK) {
            fprintf(stderr, "[ref_ts_gpu] mgpu_create(cfg %d): %s\n", cfg, mgpu_last_error(NULL));
            delete b; delete t;
            return nullptr;
        }
        mgpu_get_info(b->ctx, &b->info);
        if (mode & MIRROR) {
            b->mirror = mmirror_create(cfg, max_iters);
            if (!b->mirror) { mgpu_destroy(b->ctx); delete b; delete t; return nullptr; }
        }
    }
    g_bindings.push_back(b);
    t->operation_mode = RX_SHM;                      // main.cc:505
    if (max_iters > 0) t->default_configurations_telecom_system.ldpc_nIteration_max = max_iters;
    t->load_configuration(cfg);                      // main.cc:824
    return t;
}
void mreftsgpu_destroy(void* h) {
    Silence s;
    for (size_t i = 0; i < g_bindings.size(); i++)
        if (g_bindings[i]->ts == h) {
            Binding* b = g_bindings[i];
            if (g_current == b) g_current = nullptr;
            if (b->ctx) mgpu_destroy(b->ctx);
            if (b->ctx_ctrl) mgpu_destroy(b->ctx_ctrl);
            if (b->mirror) mmirror_destroy(b->mirror);
            g_bindings.erase(g_bindings.begin() + i);
            delete b;
            break;
        }
    delete static_cast<cl_telecom_system*>(h);
}
int mreftsgpu_set_mode(void* h, int mode) {
    for (Binding* b : g_bindings) if (b->ts == h) { const int old = b->mode; b->mode = mode; return old; }
    return -1;
}
// counters: [N_METHODS][3] = calls, calls served by the GPU, calls whose GPU output differed from the original's (SHADOW); returns N_METHODS
int mreftsgpu_counters(void* h, long* out, int reset) {
    for (Binding* b : g_bindings) if (b->ts == h) {
        for (int m = 0; m < N_METHODS; m++) { out[3 * m] = b->calls[m]; out[3 * m + 1] = b->gpu[m]; out[3 * m + 2] = b->differ[m]; }
        if (reset) for (int m = 0; m < N_METHODS; m++) b->calls[m] = b->gpu[m] = b->differ[m] = 0;
        return N_METHODS;
    }
    return -1;
}
const char* mreftsgpu_error(void* h) {
    for (Binding* b : g_bindings) if (b->ts == h) return b->error;
    return "no such object";
}

// cl_telecom_system::RX_RAND_process_main (telecom_system.cc:2102-2190) on one capture window, as the main loop runs it once the capture
// thread has filled passband_delayed_data and raised data_ready (main.cc RX_RAND branch): the function's own stdout (the decoded bytes and
// the statistics line it prints) is captured into `text`; returns its length. frames_to_read is the loop's own cross-call member and is
// left as the function sets it (in/out through *frames_to_read).
int mreftsgpu_rx_rand_process_main(void* h, const double* passband, int* frames_to_read, char* text, int text_cap) {
    cl_telecom_system* t = static_cast<cl_telecom_system*>(h);
    const int n = t->data_container.Nofdm * t->data_container.buffer_Nsymb * t->data_container.interpolation_rate;
    memcpy(t->data_container.passband_delayed_data, passband, size_t(n) * sizeof(double));
    t->data_container.frames_to_read = *frames_to_read;
    t->data_container.data_ready = 1;
    fflush(stdout);
    std::cout.flush();
    char path[] = "/tmp/mreftsgpu_stdout_XXXXXX";
    const int fd = mkstemp(path);
    const int saved = dup(1);
    dup2(fd, 1);
    t->RX_RAND_process_main();
    fflush(stdout);
    std::cout.flush();
    dup2(saved, 1);
    close(saved);
    const off_t len = lseek(fd, 0, SEEK_END);
    lseek(fd, 0, SEEK_SET);
    int got = 0;
    if (text && text_cap > 0) {
        got = int(read(fd, text, size_t(len < text_cap - 1 ? len : text_cap - 1)));
        if (got < 0) got = 0;
        text[got] = 0;
    }
    close(fd);
    unlink(path);
    *frames_to_read = t->data_container.frames_to_read;
    return got;
}

// receive_stats as the object holds it now (every field a caller can read, telecom_system.h:63-82):
// ints: iterations_done delay delay_of_last_decoded_message sync_trials message_decoded crc all_zeros mfsk_search_raw frame_overflow_symbols
// doubles: freq_offset freq_offset_of_last_decoded_message SNR signal_stregth_dbm coarse_metric
void mreftsgpu_receive_stats(void* h, int* ints, double* doubles) {
    const st_receive_stats& q = static_cast<cl_telecom_system*>(h)->receive_stats;
    ints[0] = q.iterations_done; ints[1] = q.delay; ints[2] = q.delay_of_last_decoded_message; ints[3] = q.sync_trials; ints[4] = q.message_decoded;
    ints[5] = q.crc; ints[6] = q.all_zeros; ints[7] = q.mfsk_search_raw; ints[8] = q.frame_overflow_symbols;
    doubles[0] = q.freq_offset; doubles[1] = q.freq_offset_of_last_decoded_message; doubles[2] = q.SNR; doubles[3] = q.signal_stregth_dbm;
    doubles[4] = q.coarse_metric;
}

}  // extern "C"
