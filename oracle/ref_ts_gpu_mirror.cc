// TEST INFRASTRUCTURE. include/mercury_gpu.hpp's mgpu::cl_rx_phy - the product's C++ mirror of cl_telecom_system's receive side - behind plain
// C functions, so that oracle/ref_ts_gpu_harness.cc can put it underneath the reference's own cl_telecom_system::receive_byte (mode MIRROR).
// A translation unit of its own because the mirror and the reference's headers cannot meet in one: the reference #defines YES / NO
// (physical_defines.h), the mirror declares them as enumerators.
#include <cstdio>
#include <exception>

#include "../include/mercury_gpu.hpp"

extern "C" {

struct mmirror_stats {     // mgpu::st_receive_stats, field for field
    int iterations_done, delay, delay_of_last_decoded_message, sync_trials, message_decoded, crc, all_zeros, mfsk_search_raw, frame_overflow_symbols;
    double freq_offset, freq_offset_of_last_decoded_message, SNR, signal_stregth_dbm, coarse_metric;
};

void* mmirror_create(int cfg, int max_iters) {
    try {
        mgpu::cl_rx_phy* p = new mgpu::cl_rx_phy();
        if (max_iters > 0) p->ldpc_nIteration_max = max_iters;
        p->load_configuration(cfg);
        return p;
    } catch (const std::exception& e) {
        fprintf(stderr, "[ref_ts_gpu mirror] %s\n", e.what());
        return nullptr;
    }
}
void mmirror_destroy(void* h) { delete static_cast<mgpu::cl_rx_phy*>(h); }

// the members code outside receive_byte may have changed between two calls (the process loops adjust delay_of_last_decoded_message,
// telecom_system.cc:2156; the ARQ layer sets mfsk_search_raw / mfsk_fixed_delay; set_mfsk_ctrl_mode) go in, the call runs on the mirror's OWN
// receive_stats (so what it leaves unwritten is the mirror's own previous value), and the whole struct comes back
int mmirror_receive_byte(void* h, const double* data, int* out, double carrier_hz, int time_sync_trials_max, int use_last_good_time_sync,
                         int use_last_good_freq_offset, int coarse_freq_sync_enabled, int ctrl_mode, int nUnder_processing_events, int* mfsk_fixed_delay,
                         int delay_of_last_decoded_message, double freq_offset_of_last_decoded_message, int mfsk_search_raw, mmirror_stats* held) {
    mgpu::cl_rx_phy* p = static_cast<mgpu::cl_rx_phy*>(h);
    try {
        p->carrier_frequency = carrier_hz; p->time_sync_trials_max = time_sync_trials_max; p->use_last_good_time_sync = use_last_good_time_sync;
        p->use_last_good_freq_offset = use_last_good_freq_offset; p->coarse_freq_sync_enabled = coarse_freq_sync_enabled;
        if (bool(ctrl_mode) != p->mfsk_ctrl_mode) p->set_mfsk_ctrl_mode(ctrl_mode != 0);
        p->nUnder_processing_events = nUnder_processing_events; p->mfsk_fixed_delay = *mfsk_fixed_delay;
        p->receive_stats.delay_of_last_decoded_message = delay_of_last_decoded_message;
        p->receive_stats.freq_offset_of_last_decoded_message = freq_offset_of_last_decoded_message;
        p->receive_stats.mfsk_search_raw = mfsk_search_raw;
        const mgpu::st_receive_stats q = p->receive_byte(data, out);
        *mfsk_fixed_delay = p->mfsk_fixed_delay;
        *held = mmirror_stats{q.iterations_done, q.delay, q.delay_of_last_decoded_message, q.sync_trials, q.message_decoded, q.crc, q.all_zeros,
                              q.mfsk_search_raw, q.frame_overflow_symbols, q.freq_offset, q.freq_offset_of_last_decoded_message, q.SNR,
                              q.signal_stregth_dbm, q.coarse_metric};
        return 0;
    } catch (const std::exception& e) {
        fprintf(stderr, "[ref_ts_gpu mirror] %s\n", e.what());
        return 1;
    }
}

// gear shifts on the mirror: load_configuration(cfg) (cl_rx_phy keeps the pre-equalisation table it measured for a (modulation, preamble,
// carrier, seeds) key and re-installs it; an MFSK load drops the key as the reference's sticky reinit flag does) and the table it holds
int mmirror_load_configuration(void* h, int cfg) {
    try { static_cast<mgpu::cl_rx_phy*>(h)->load_configuration(cfg); return 0; }
    catch (const std::exception& e) { fprintf(stderr, "[ref_ts_gpu mirror] %s\n", e.what()); return 1; }
}
int mmirror_pre_equalization_channel(void* h, double* out) {
    mgpu::cl_rx_phy* p = static_cast<mgpu::cl_rx_phy*>(h);
    for (size_t j = 0; j < p->pre_equalization_channel.size(); ++j) { out[2 * j] = p->pre_equalization_channel[j].real(); out[2 * j + 1] = p->pre_equalization_channel[j].imag(); }
    return int(p->pre_equalization_channel.size());
}

}  // extern "C"
