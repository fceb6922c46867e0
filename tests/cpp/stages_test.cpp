// receive_byte's front half written the way telecom_system.cc:1132-1298 writes it — one cl_ofdm / cl_psk method after the
// other — against include/mercury_gpu.hpp; the per-stage results are dumped for comparison with the CPU oracle.
//   usage: stages_test <cfg> <baseband.bin> <out.bin>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mercury_gpu.hpp"

typedef std::complex<double> cd;

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    try {
        mgpu::cl_rx_phy phy;
        phy.load_configuration(atoi(argv[1]));
        const mgpu_info& I = phy.info;
        mgpu::cl_ofdm ofdm(phy.context(), I);
        mgpu::cl_psk psk(phy.context(), I);
        const int G = I.Nsymb * I.Nc;
        std::vector<cd> baseband(I.frame_samples), demod(G), eq(G), eq_noamp(G), deframed(I.nData), tf(I.nData);
        std::vector<float> demodulated(I.nBits), deinterleaved(I.nBits);
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(baseband.data(), sizeof(cd), baseband.size(), f) != baseband.size()) return 2;
        fclose(f);
        // telecom_system.cc:1135-1138
        for (int i = 0; i < I.Nsymb; i++) ofdm.symbol_demod(&baseband[i * I.Nofdm], &demod[i * I.Nc]);
        ofdm.automatic_gain_control(demod.data());                                               // :1197
        if (ofdm.channel_estimator == MGPU_EST_ZF) ofdm.ZF_channel_estimator(demod.data());      // :1215-1222
        else ofdm.LS_channel_estimator(demod.data());
        if (ofdm.channel_estimator_amplitude_restoration == mgpu::YES) ofdm.restore_channel_amplitude();   // :1282-1287
        std::vector<cd> H = ofdm.estimated_channel;
        ofdm.channel_equalizer(demod.data(), eq.data());                                         // :1289
        float variance = ofdm.measure_variance(eq.data());                                       // :1291 (double -> float member)
        ofdm.deframer(eq.data(), deframed.data());                                               // :1293
        mgpu::deinterleaver(phy.context(), deframed.data(), tf.data(), I.nData, I.tf_blk);       // :1294
        psk.demod(tf.data(), I.nBits, demodulated.data(), variance);                             // :1296
        mgpu::deinterleaver(phy.context(), demodulated.data(), deinterleaved.data(), I.nBits, I.bit_blk);   // :1298
        // the integer tail on a known pattern (telecom_system.cc:1313-1341): descramble, pack, CRC self-check
        const int nReal = I.nReal;
        std::vector<int> bits(nReal), desc(nReal), bytes((nReal + 7) / 8);
        for (int i = 0; i < nReal; i++) bits[i] = (i * 7 + i / 3) & 1;
        mgpu::bit_energy_dispersal(phy.context(), bits.data(), nullptr, desc.data(), nReal);
        mgpu::bit_to_byte(phy.context(), desc.data(), bytes.data(), nReal);
        const int crc = mgpu::CRC16_MODBUS_RTU_calc(phy.context(), bytes.data(), nReal / 8);
        FILE* o = fopen(argv[3], "wb");
        fwrite(demod.data(), sizeof(cd), G, o);
        fwrite(H.data(), sizeof(cd), G, o);
        fwrite(eq.data(), sizeof(cd), G, o);
        fwrite(&variance, sizeof(float), 1, o);
        fwrite(tf.data(), sizeof(cd), I.nData, o);
        fwrite(demodulated.data(), sizeof(float), I.nBits, o);
        fwrite(deinterleaved.data(), sizeof(float), I.nBits, o);
        fwrite(desc.data(), sizeof(int), nReal, o);
        fwrite(bytes.data(), sizeof(int), bytes.size(), o);
        fwrite(&crc, sizeof(int), 1, o);
        fclose(o);
    } catch (const std::exception& e) {
        fprintf(stderr, "stages_test: %s\n", e.what());
        return 1;
    }
    return 0;
}
