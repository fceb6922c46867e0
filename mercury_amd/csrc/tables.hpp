// Host-side construction of every constant table the RX kernels consume.
//
// These are the init-time computations of the reference (done once per load_configuration),
// restated so that the device sees bit-identical constants:
//   mode table ............ telecom_system.cc:2506-2654, :1806-1869, data_container.cc:90-99
//   MFSK parameters ....... mfsk.cc:48-78, telecom_system.cc:1810-1816, :1940-1946, :2968-2989
//   PRNG .................. source/common/os_interop.cc:192-283 (glibc TYPE_3 random())
//   pilot lattice/values .. ofdm.cc:904-952, :976-1064
//   constellation ......... psk.cc:65-256
//   FFT twiddles .......... ofdm.cc:256-290
//   interleavers .......... interleaver.cc:77-109 ; re-pack telecom_system.cc:1300-1308
//   scrambler ............. telecom_system.cc:1961-1966
//   preamble .............. ofdm.cc:1127-1232 ; MFSK mfsk.cc:82-95, :172-193
//   transmit filters ...... fir_filter.cc:45-162, physical_config.cc:103-113
//   LDPC graph ............ mercury_normal_*_16.cc via mercury_ldpc_tables.bin (derived data)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mgpu {

struct Cplx { double re, im; };

// glibc TYPE_3 additive-feedback generator; reference carries its own copy so pilots and the
// scrambler are identical on every libc (os_interop.cc:157-170 holds the default state).
class GlibcRandom {
public:
    explicit GlibcRandom(unsigned seed) { reseed(seed); }
    void reseed(unsigned seed);
    int32_t next();
private:
    int32_t st_[31];
    int f_, r_;
};

// decodes an MGPU_CFG_EXPLICIT id (include/mercury_gpu.h); false if cfg is not one
bool explicit_mode_row(int cfg, int* M, int* rate16, int* preamble, int* estimator);

struct LdpcGraph {
    int K = 0, P = 0, N = 1600, E = 0, Cwidth = 0, Vwidth = 0;
    std::vector<uint32_t> cptr;    // [P+1] first edge of each check (edges are check-major, reference row order)
    std::vector<uint16_t> cvar;    // [E]   variable of edge e
    std::vector<uint32_t> vptr;    // [N+1] first slot of each variable
    std::vector<uint16_t> vedge;   // [E]   edge index of (variable, slot) in the reference's slot order
    // wave-private layout for the sum-product kernel: whole checks bin-packed (first-fit decreasing)
    // into 64-slot bins so that one wavefront owns every edge of the checks it updates
    int S = 0;                     // padded slot count = 64 * bins
    std::vector<uint32_t> vinfo;   // [N][8] (6 used) variable | deg<<11, then 10 u16 padded-slot indices; rows sorted by degree (descending); host only: vinfo2 is its device form
    // grouped layout of the fp32 sum-product kernel (ldpc.hip, spa_fast): every check sits in an ALIGNED group of 2, 4, 8, 16, 32 or 64 lanes
    // (its degree rounded up to a power of two), a bin holds groups of one size, so the product over a check is a handful of DPP
    // steps inside the wavefront instead of a walk through LDS
    int Sg = 0;                    // slots = 64 * bins
    std::vector<uint32_t> gdesc;   // [(rounds+1)*512] per slot: 0x80000000 | variable, 0 for padding lanes (rounds of 8 bins: 512-thread workgroups)
    std::vector<uint32_t> gkind;   // [(rounds+1)*8] per bin: log2 of its group size (1..6), 0 for an empty bin
    std::vector<uint64_t> gkpack;  // [8] per wavefront w: gkind[w + 8 r] in bits 3r .. 3r+2 (at most 21 rounds): one scalar register pair per wavefront
    std::vector<uint32_t> vinfo_g; // [N][8] like vinfo, slot indices in the grouped layout; the rows' order is the grouped layout's own (bank-aware, tables.cpp)
    double bank_model[2] = {0, 0}; // modelled LDS cycles per 32-lane gather group (1.0 = conflict-free) of the check pass's posterior reads / the variable update's message reads
    // the fp64 sum-product kernel's own tables (ldpc.hip, spa_decode): LDS byte offsets instead of indices, and the
    // product walk's execution masks tabulated per bin and step instead of compared per lane
    // register-layout limits of individual kernels, recorded here and enforced where a context selects its decoder (api.hip: ctx_alloc), so
    // that a graph one decoder cannot hold still loads for the others (GBF walks the plain lists and has no such limit)
    std::string fp64_limit, fp32_limit;   // empty = fits; otherwise what the fp64 / the fp32 sum-product + min-sum kernels cannot hold
    int maxdeg = 0;                // largest check degree
    int DM = 0;                    // mask row length: the largest check degree rounded up to a multiple of 8 (the walk's group size)
    std::vector<uint32_t> sadr;    // [(rounds+1)*1024][2] per padded slot: LDS byte offset of its variable's posterior (variable*8), LDS byte address of its check's first message (8*N + check_start*8); 0 for padding
    std::vector<uint64_t> bhead;   // [(rounds+1)*16 bins][4] lanes in use, lanes holding the last edge of a check, largest check degree in the bin, 0
    std::vector<uint64_t> bmask;   // [rounds*16 bins][DM] lanes of the bin that multiply factor j in: position != j and degree > j (0 past the bin's largest degree)
    std::vector<uint32_t> vinfo2;  // [N][8] like vinfo with byte offsets (slot*8) in the 10 u16 fields
};

// The parameters physical_config.cc:30-65 gives every mode and telecom_system.cc:2772-2811 copies into the DSP objects; a context may
// override them (include/mercury_gpu.h: mgpu_explicit_params). Defaults = the reference's.
struct ExplicitParams {
    float pilot_boost = 1.33f;     // ofdm_pilot_configurator_pilot_boost (physical_config.h:53: float)
    int ls_window = 20;            // ofdm_LS_window_width = _hight; an even value is incremented (telecom_system.cc:2802-2809)
    unsigned pilot_seed = 0;       // ofdm_pilot_configurator_seed
    unsigned scrambler_seed = 0;   // bit_energy_dispersal_seed
    unsigned preamble_seed = 1;    // ofdm_preamble_configurator_seed
    int Nsymb = 0;                 // ofdm_Nsymb; 0 = what init() selects from the modulation for HIGH_DENSITY pilots (telecom_system.cc:1810-1826)
    int Dy = 3;                    // ofdm_pilot_configurator_Dy: 3 = HIGH_DENSITY (every mode's default), 5 = the reference's LOW_DENSITY option (telecom_system.cc:1848-1869)
};

struct ModeTables {
    ExplicitParams xp;
    int cfg = 0, M = 0, bps = 0, K = 0, P = 0, N = 1600;
    int Nsymb = 0, Nc = 50, Nfft = 256, Ngi = 16, Nofdm = 272;
    int nData = 0, nBits = 0, nPilots = 0, nVirtual = 0, nReal = 0;
    int bit_blk = 0, tf_blk = 0, preamble = 0, estimator = 1, amp_restore = 0, lsw = 21;
    int payload_bytes = 0, payload_stride = 0, frame_samples = 0;
    // MFSK modes (cfg 100..102 = ROBUST_0..2): cl_mfsk::init mfsk.cc:48-78; control frames telecom_system.cc:2968-2989
    int mfsk_M = 0, mfsk_nbits = 0, mfsk_nstreams = 0, mfsk_hop = 0, mfsk_off[4] = {0, 0, 0, 0};
    int ctrl_nbits = 0, ctrl_nsymb = 0;
    int active_nbits = 0, active_nsymb = 0; // bits / symbols on the air (== nBits / Nsymb unless mfsk_ctrl_mode)
    double mfsk_amp = 0;                    // sqrt(Nc / nStreams), mfsk.cc:243
    std::vector<uint16_t> llr_dst;          // [nBits] MFSK only: decoder input position of demodulated LLR i (inverse of llr_src)
    double pilot_boost = 0;                 // (double)(float)1.33
    std::vector<uint8_t> cell_type;         // [G] 0 DATA / 1 PILOT
    std::vector<double> pilot_val;          // [G] real pilot value (0 at data cells)
    std::vector<Cplx> constellation;        // [M]
    std::vector<uint8_t> scrambler;         // [1600]
    std::vector<Cplx> twiddle;              // [128] exp(-2 pi i k/256)
    std::vector<uint16_t> sym_src;          // [nData] grid cell feeding de-interleaved symbol k
    std::vector<uint16_t> llr_src;          // [1600] index into the demod LLR vector feeding decoder input p
    std::vector<double> fir_time_sync, fir_data;   // receive FIRs (fir_filter.cc:45-131, parameters physical_config.cc:90-98)
    std::vector<double> ls_weight;          // [lsw*lsw+1] boost/sum_n(boost^2) per window population n
    // TX-side permutations for the synthetic generator
    std::vector<uint16_t> bit_il;           // [nBits] interleaved position <- encoded index: out[bit_il[i]] = in[i]
    std::vector<uint16_t> tf_inv;           // [nData] symbol k with tf-deinterleave source i (inverse of the RX gather)
    std::vector<uint16_t> data_cell;        // [nData] grid cell of de-framed position i (deframer order)
    std::vector<uint16_t> sym_cell;         // [nData] grid cell of modulated symbol k (tf-interleave + framer)
    std::vector<Cplx> preamble_carriers;    // [preamble][Nc] known symbols sent in front of the frame (ofdm.cc:1127-1232; MFSK: mfsk.cc:172-193)
    LdpcGraph graph;
};

// Throws std::runtime_error on a bad cfg or unreadable/corrupt table blob.
ModeTables build_mode_tables(int cfg, int mfsk_ctrl_mode, const uint8_t* ldpc_blob, size_t ldpc_blob_size, const ExplicitParams& xp = ExplicitParams());

uint16_t crc16_modbus(const uint8_t* bytes, int n);

// pre_equalization_channel of a freshly loaded configuration for a given carrier (telecom_system.cc:3108-3145): [Nc]
std::vector<Cplx> pre_equalization_channel(const ModeTables& t, double carrier_hz);

// transmit filters for a given carrier: which 0 = FIR_tx1 (HPF, Hamming), 1 = FIR_tx2 (LPF, Blackman); physical_config.cc:103-113
std::vector<double> design_tx_fir(int which, double carrier_hz);

}  // namespace mgpu
