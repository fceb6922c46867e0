// Host-language (C++) test of the drop-in boundary: drives include/mercury_gpu.hpp the way
// telecom_system.cc drives the reference classes, on inputs written by tests/test_cpp_shim.py, and
// writes the results back for comparison with the CPU oracle.
//   usage: shim_test <cfg> <nframes> <baseband.bin> <llr.bin> <out.bin> [<passband_windows.bin> <nwindows>]
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "mercury_gpu.hpp"

template <typename T>
static std::vector<T> slurp(const char* path, size_t n) {
    std::vector<T> v(n);
    FILE* f = fopen(path, "rb");
    if (!f || fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc != 6 && argc != 8 && argc != 9) return 2;
    const int cfg = atoi(argv[1]), F = atoi(argv[2]);
    try {
        mgpu::cl_rx_phy phy;
        phy.max_batch = F;
        phy.load_configuration(cfg);                     // telecom_system.cc:824 / :2487
        phy.load_configuration(cfg);                     // second call is a no-op, as in the reference (:2489-2492)
        const int fs = phy.info.frame_samples, pb = phy.get_frame_size_bytes();
        auto bb = slurp<std::complex<double>>(argv[3], size_t(F) * fs);
        auto llr = slurp<float>(argv[4], size_t(F) * 1600);
        FILE* out = fopen(argv[5], "wb");
        // 1) frame at a time, the way receive_byte is called
        for (int f = 0; f < F; ++f) {
            std::vector<int> bytes(pb);
            mgpu::st_receive_stats st = phy.receive_frame(&bb[size_t(f) * fs], bytes.data());
            int rec[4] = {st.iterations_done, st.crc, st.all_zeros, st.message_decoded};
            fwrite(rec, sizeof(int), 4, out);
            fwrite(bytes.data(), sizeof(int), pb, out);
        }
        // 2) one batched call
        std::vector<uint8_t> payload(size_t(F) * phy.info.payload_stride);
        std::vector<mgpu::st_receive_stats> stats;
        phy.receive_batch(bb.data(), F, payload.data(), stats);
        for (int f = 0; f < F; ++f) {
            int rec[4] = {stats[f].iterations_done, stats[f].crc, stats[f].all_zeros, stats[f].message_decoded};
            fwrite(rec, sizeof(int), 4, out);
        }
        fwrite(payload.data(), 1, payload.size(), out);
        // 3) cl_ldpc alone, exactly the reference's call sequence (ldpc.init(); ldpc.decode(llr, bits))
        mgpu::cl_ldpc ldpc;
        ldpc.rate = float(phy.info.K) / 1600.0f;
        ldpc.nIteration_max = 50;
        ldpc.init();
        for (int f = 0; f < F; ++f) {
            std::vector<int> bits(ldpc.K);
            int it = ldpc.decode(&llr[size_t(f) * 1600], bits.data());
            fwrite(&it, sizeof(int), 1, out);
            fwrite(bits.data(), sizeof(int), ldpc.K, out);
        }
        {   // cl_ldpc::encode then decode of the noiseless word: 0 iterations, the data back
            std::vector<int> data(ldpc.K), enc(ldpc.N), back(ldpc.K);
            for (int i = 0; i < ldpc.K; ++i) data[i] = (i * 7 + i / 3) & 1;
            ldpc.encode(data.data(), enc.data());
            std::vector<float> l(ldpc.N);
            for (int i = 0; i < ldpc.N; ++i) l[i] = enc[i] ? -4.0f : 4.0f;
            if (ldpc.decode(l.data(), back.data()) != 0 || back != data) return 9;
        }
        fclose(out);
        // 4) the whole receive_byte on passband capture windows, one call per window like RX_SHM_process_main
        //    (optional 7th argument: file of W windows; results appended to <out>.rb)
        if (argc >= 8) {
            const int W = atoi(argv[7]), n = phy.capture_window_samples();
            auto pass = slurp<double>(argv[6], size_t(W) * n);
            FILE* rb = fopen((std::string(argv[5]) + ".rb").c_str(), "wb");
            for (int w = 0; w < W; ++w) {
                std::vector<int> bytes(pb);
                mgpu::st_receive_stats st = phy.receive_byte(&pass[size_t(w) * n], bytes.data());
                int rec[6] = {st.iterations_done, st.crc, st.message_decoded, st.delay, st.sync_trials, st.delay_of_last_decoded_message};
                fwrite(rec, sizeof(int), 6, rb);
                fwrite(bytes.data(), sizeof(int), pb, rb);
            }
            fclose(rb);
        }
        // 4b) transmit_byte, three consecutive SINGLE_MESSAGE calls and one NO_FILTER_MESSAGE like the ARQ batch sender
        //     (optional 9th argument: file of 4 messages, pb ints each; samples appended to <out>.tx)
        if (argc >= 9) {
            auto msgs = slurp<int>(argv[8], size_t(4) * pb);
            const int total = phy.total_frame_size();
            std::vector<double> audio(total);
            FILE* tx = fopen((std::string(argv[5]) + ".tx").c_str(), "wb");
            for (int m = 0; m < 4; ++m) {
                if (!phy.transmit_byte(&msgs[size_t(m) * pb], m == 1 ? pb / 2 : pb, audio.data(), m == 3 ? MGPU_NO_FILTER_MESSAGE : MGPU_SINGLE_MESSAGE)) return 4;
                fwrite(audio.data(), sizeof(double), total, tx);
            }
            // transmit_bit (telecom_system.cc:384): the frame's data bits as they are — the bits transmit_byte would have made of message 0
            // (CRC appended, computed here the way :365-372 does) must give message 0's audio again, the carrier running on
            {
                const int nReal = phy.info.nReal;
                std::vector<int> bits(nReal, 0);
                unsigned crc = 0xffff;
                for (int j = 0; j < pb; ++j) {
                    crc ^= unsigned(msgs[j]) & 0xff;
                    for (int i = 0; i < 8; ++i) crc = (crc & 1) ? ((crc >> 1) ^ 0xA001) : (crc >> 1);
                }
                for (int j = 0; j < pb; ++j) for (int i = 0; i < 8; ++i) bits[j * 8 + i] = (msgs[j] >> i) & 1;
                for (int i = 0; i < 8; ++i) { bits[pb * 8 + i] = (crc >> i) & 1; bits[(pb + 1) * 8 + i] = (crc >> (8 + i)) & 1; }
                phy.transmit_bit(bits.data(), audio.data(), MGPU_SINGLE_MESSAGE);
                fwrite(audio.data(), sizeof(double), total, tx);
            }
            fclose(tx);
            // receive_bit (telecom_system.cc:636): the first capture window again, bits instead of bytes
            if (argc >= 8 && atoi(argv[7]) > 0) {
                const int n = phy.capture_window_samples();
                auto pass = slurp<double>(argv[6], size_t(n));
                std::vector<int> rbits((phy.info.nReal / 8) * 8, -1);
                mgpu::cl_rx_phy fresh;                    // a receiver without the history of the calls above
                fresh.load_configuration(cfg);
                const mgpu::st_receive_stats st = fresh.receive_bit(pass.data(), rbits.data());
                FILE* rb = fopen((std::string(argv[5]) + ".rbits").c_str(), "wb");
                const int dec = st.message_decoded;
                fwrite(&dec, sizeof(int), 1, rb);
                fwrite(rbits.data(), sizeof(int), rbits.size(), rb);
                fclose(rb);
                // return_to_last_configuration (telecom_system.cc:3027-3034): back and forth between two modes
                const int other = cfg >= 100 ? 101 : 5;
                fresh.load_configuration(other);
                if (fresh.current_configuration != other || fresh.last_configuration != cfg) return 9;
                fresh.return_to_last_configuration();
                // the reference's own bookkeeping (telecom_system.cc:3027-3034): the two members read as before the call, the loaded mode is the former last
                if (fresh.current_configuration != other || fresh.last_configuration != cfg || fresh.info.cfg != cfg) return 10;
            }
            std::vector<int> too_long(pb + 1, 0);
            if (phy.transmit_byte(too_long.data(), pb + 1, audio.data(), MGPU_SINGLE_MESSAGE)) return 5;     // "message too long.. not sent."
            // default_configurations_telecom_system (physical_config.cc:35-65), set before load_configuration like the reference's callers do:
            // another pilot boost / LS window / seeds give another waveform, and a receiver configured the same way takes it back
            if (cfg < 100 && argc >= 8 && atoi(argv[7]) > 0) {
                mgpu::cl_rx_phy other;
                other.default_configurations_telecom_system.ofdm_pilot_configurator_pilot_boost = 1.7f;
                other.default_configurations_telecom_system.ofdm_LS_window_width = 10;
                other.default_configurations_telecom_system.ofdm_pilot_configurator_seed = 5;
                other.default_configurations_telecom_system.bit_energy_dispersal_seed = 7;
                other.default_configurations_telecom_system.ofdm_preamble_configurator_seed = 3;
                other.load_configuration(cfg);
                if (other.info.ls_window != 11) return 11;
                std::vector<double> a2(total);
                if (!other.transmit_byte(&msgs[0], pb, a2.data(), MGPU_SINGLE_MESSAGE)) return 12;
                std::vector<double> a1(total);
                if (!phy.transmit_byte(&msgs[0], pb, a1.data(), MGPU_SINGLE_MESSAGE)) return 12;
                if (a1 == a2) return 13;
                const int nw = other.capture_window_samples();
                std::vector<double> win(nw, 0.0);
                for (int i = 0; i < total && 20000 + i < nw; ++i) win[20000 + i] = a2[i];
                std::vector<int> got(pb, -1);
                const mgpu::st_receive_stats st2 = other.receive_byte(win.data(), got.data());
                if (!st2.message_decoded) return 14;
                for (int j = 0; j < pb; ++j) if (got[j] != (msgs[j] & 0xff)) return 15;
                const mgpu::st_receive_stats st1 = phy.receive_byte(win.data(), got.data());     // the default receiver must not take it
                if (st1.message_decoded) return 16;
            }
            // ofdm_pilot_density = LOW_DENSITY as cl_telecom_system::init resolves it (telecom_system.cc:1828-1836, :1857-1865): Dy 5 and 40 / 20 / 10
            // symbols for BPSK / QPSK / 16QAM - another frame length and another pilot lattice; the frame comes back through a receiver
            // configured the same way and not through the default one
            if (cfg < 100 && argc >= 8 && atoi(argv[7]) > 0 && (phy.info.M == 2 || phy.info.M == 4 || phy.info.M == 16)) {
                mgpu::cl_rx_phy low;
                low.default_configurations_telecom_system.ofdm_pilot_configurator_Dy = 5;
                low.default_configurations_telecom_system.ofdm_Nsymb = phy.info.M == 2 ? 40 : phy.info.M == 4 ? 20 : 10;
                low.load_configuration(cfg);
                if (low.info.Nsymb != low.default_configurations_telecom_system.ofdm_Nsymb || low.info.nBits != 1600 || low.info.nPilots * 5 != low.info.Nsymb * 50) return 17;
                const int lt = low.total_frame_size();
                if (lt >= total) return 18;                                     // fewer symbols per frame than with HIGH_DENSITY pilots
                std::vector<double> al(lt);
                if (!low.transmit_byte(&msgs[0], pb, al.data(), MGPU_SINGLE_MESSAGE)) return 19;
                const int nw = low.capture_window_samples();
                std::vector<double> win(nw, 0.0);
                for (int i = 0; i < lt && 20000 + i < nw; ++i) win[20000 + i] = al[i];
                std::vector<int> got(pb, -1);
                const mgpu::st_receive_stats sl = low.receive_byte(win.data(), got.data());
                if (!sl.message_decoded) return 20;
                for (int j = 0; j < pb; ++j) if (got[j] != (msgs[j] & 0xff)) return 21;
                const int nwh = phy.capture_window_samples();
                std::vector<double> winh(nwh, 0.0);
                for (int i = 0; i < lt && 20000 + i < nwh; ++i) winh[20000 + i] = al[i];
                const mgpu::st_receive_stats sh = phy.receive_byte(winh.data(), got.data());
                if (sh.message_decoded) return 22;
            }
            // 4c) the signalling calls the ARQ layer makes: ACK pattern out and back in, signal level, control-frame mode
            const int n = phy.capture_window_samples(), na = phy.ack_pattern_passband_samples();
            std::vector<double> buf(n, 0.0), ack(na);
            if (phy.generate_ack_pattern_passband(ack.data()) != na) return 6;
            for (int i = 0; i < na; ++i) buf[5000 + i] = ack[i];
            int m_ack = 0, m_brk = 0;
            const double metric = phy.detect_ack_pattern_from_passband(buf.data(), n, &m_ack);
            phy.detect_break_pattern_from_passband(buf.data(), n, &m_brk);
            const double dbm = phy.measure_signal_only(buf.data());
            const int data_nsymb = phy.get_active_nsymb();
            phy.set_mfsk_ctrl_mode(true);
            const int ctrl_nsymb = phy.get_active_nsymb();
            phy.set_mfsk_ctrl_mode(false);
            const double sig[6] = {metric, double(m_ack), double(m_brk), dbm, double(data_nsymb), double(ctrl_nsymb)};
            FILE* sg = fopen((std::string(argv[5]) + ".sig").c_str(), "wb");
            fwrite(sig, sizeof(double), 6, sg);
            fclose(sg);
            if (phy.get_active_nsymb() != data_nsymb) return 7;
            // 4d) the two BER self-simulations through their mirrors, at points where every frame decodes
            const mgpu_error_rate eb = phy.baseband_test_EsN0(cfg >= 100 ? 10.0f : 20.0f, 4);
            const mgpu_error_rate ep = phy.passband_test_EsN0(cfg >= 100 ? 10.0f : 30.0f, 3);
            const double ber[6] = {double(eb.Frames_total), double(eb.Error_frames_total), double(eb.Bits_total), double(ep.Frames_total),
                                   double(ep.Error_frames_total), double(ep.crc_ok_frames)};
            FILE* bf = fopen((std::string(argv[5]) + ".ber").c_str(), "wb");
            fwrite(ber, sizeof(double), 6, bf);
            fclose(bf);
        }
        // get_configuration (telecom_system.cc:3036-3108): spot checks on both sides of a few thresholds
        if (mgpu::cl_rx_phy::get_configuration(13.0) != 15 || mgpu::cl_rx_phy::get_configuration(12.5) != 14 || mgpu::cl_rx_phy::get_configuration(0.6) != 8 ||
            mgpu::cl_rx_phy::get_configuration(0.5) != 7 || mgpu::cl_rx_phy::get_configuration(-7.5) != 0 || mgpu::cl_rx_phy::get_configuration(-7.4) != 1) return 8;
        // 5) error behaviour: a wrong code rate throws instead of exit(1)
        mgpu::cl_ldpc bad;
        bad.rate = 7.0f / 16.0f;
        try { bad.init(); return 3; } catch (const std::runtime_error&) {}
    } catch (const std::exception& e) {
        fprintf(stderr, "shim_test: %s\n", e.what());
        return 1;
    }
    return 0;
}
