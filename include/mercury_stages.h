/* mercury_stages.h — the RX path one reference method at a time (SURVEY.md §8b "signatures to keep").
 *
 * The production entry points (mercury_gpu.h) run the whole span fused; these expose each cl_ofdm / cl_psk method
 * and the free deinterleaver on its own, so host code that keeps the reference's call sequence
 * (telecom_system.cc:1132-1298) — or a test that wants one stage — can call them in place of the originals. F frames per
 * call, host buffers, blocking; the values are the same as the corresponding taps of mgpu_rx_batch_taps.
 *
 *   mgpu_symbol_demod                void cl_ofdm::symbol_demod(complex<double>* in, complex<double>* out)      ofdm.h:133
 *   mgpu_automatic_gain_control      void cl_ofdm::automatic_gain_control(complex<double>* in)                   ofdm.h:146
 *   mgpu_channel_estimator           void cl_ofdm::LS_channel_estimator / ZF_channel_estimator(complex<double>* in)  ofdm.h:137-138
 *                                    (whichever the mode uses; estimated_channel[].value is returned, incl. interpolation)
 *   mgpu_restore_channel_amplitude   void cl_ofdm::restore_channel_amplitude()                                   ofdm.h:139
 *   mgpu_channel_equalizer           void cl_ofdm::channel_equalizer(complex<double>* in, complex<double>* out)  ofdm.h:140
 *   mgpu_measure_variance            double cl_ofdm::measure_variance(complex<double>* in)                       ofdm.h:142
 *   mgpu_deframer                    void cl_ofdm::deframer(complex<double>* in, complex<double>* out)           ofdm.h:136
 *   mgpu_deinterleaver_c128 / _f32   void deinterleaver(T* in, T* out, int nItems, int block_size)               interleaver.h:28-34
 *   mgpu_psk_demod                   void cl_psk::demod(const complex<double>* in, int nItems, float* out, float variance)  psk.h:55
 *   mgpu_bit_energy_dispersal        void bit_energy_dispersal(int* in, int* seq, int* out, int nItems)          interleaver.h:36 (seq = the mode's)
 *   mgpu_bit_to_byte                 void bit_to_byte(int* in, int* out, int nItems)                            misc.h
 *   mgpu_crc16_modbus_rtu            uint16_t CRC16_MODBUS_RTU_calc(int* data, int nItems)                      crc16_modbus_rtu.h
 *                                    (bits and byte values travel as one uint8_t each instead of one int)
 * G = Nsymb*Nc cells per frame grid; complex arrays are interleaved (re, im) doubles. OFDM modes only.
 */
#ifndef MERCURY_STAGES_H
#define MERCURY_STAGES_H

#include "mercury_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

int mgpu_symbol_demod(mgpu_ctx* ctx, const double* in_c128 /*[n][Nofdm]*/, int n_symbols, double* out_c128 /*[n][Nc]*/);
int mgpu_automatic_gain_control(mgpu_ctx* ctx, double* grid_c128 /*[F][G], in place*/, int F);
int mgpu_channel_estimator(mgpu_ctx* ctx, const double* grid_c128 /*[F][G]*/, int F, double* H_c128 /*[F][G]*/);
int mgpu_restore_channel_amplitude(mgpu_ctx* ctx, double* H_c128 /*[F][G], in place*/, int F);
int mgpu_channel_equalizer(mgpu_ctx* ctx, const double* grid_c128, const double* H_c128, int F, double* out_c128 /*[F][G]*/);
int mgpu_measure_variance(mgpu_ctx* ctx, const double* grid_c128 /*[F][G]*/, int F, double* variance /*[F]*/);
int mgpu_deframer(mgpu_ctx* ctx, const double* grid_c128 /*[F][G]*/, int F, double* data_c128 /*[F][nData]*/);
int mgpu_deinterleaver_c128(mgpu_ctx* ctx, const double* in, int F, int nItems, int block_size, double* out);
int mgpu_deinterleaver_f32(mgpu_ctx* ctx, const float* in, int F, int nItems, int block_size, float* out);
int mgpu_psk_demod(mgpu_ctx* ctx, const double* syms_c128 /*[F][nData]*/, int F, const float* variance /*[F]*/, float* llr /*[F][nBits]*/);
int mgpu_bit_energy_dispersal(mgpu_ctx* ctx, const uint8_t* bits /*[F][n]*/, int F, int n, uint8_t* out /*[F][n]*/);
int mgpu_bit_to_byte(mgpu_ctx* ctx, const uint8_t* bits /*[F][nbits]*/, int F, int nbits, uint8_t* bytes /*[F][ceil(nbits/8)]*/);
int mgpu_crc16_modbus_rtu(mgpu_ctx* ctx, const uint8_t* bytes /*[F][n]*/, int F, int n, uint16_t* crc /*[F]*/);

#ifdef __cplusplus
}
#endif
#endif /* MERCURY_STAGES_H */
