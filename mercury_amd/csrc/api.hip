// C-ABI of the MI355X-native Mercury RX physical layer (declared in include/mercury_gpu.h).
// Host side only: context, constant-table upload, workspace management, kernel launches.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <atomic>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "numa.hpp"

typedef void (*fe_kernel_t)(MgpuDev, const double*, int, float*, float*, float*, double*, MgpuTapsDev);
static fe_kernel_t fe_kernel(int threads) { return threads == 1024 ? mgpu_frontend_kernel_t1024 : mgpu_frontend_kernel; }

thread_local std::string g_create_error;

namespace mgpu_detail {

void ctx_alloc(mgpu_ctx* c) {
    const auto& t = c->tab;
    MgpuDev& d = c->dev;
    std::vector<uint16_t> pilot_cell;
    for (int i = 0; i < t.Nsymb * t.Nc; ++i) if (t.cell_type[i]) pilot_cell.push_back(uint16_t(i));
    std::vector<double> cons, tw;
    for (auto& z : t.constellation) { cons.push_back(z.re); cons.push_back(z.im); }
    for (auto& z : t.twiddle) { tw.push_back(z.re); tw.push_back(z.im); }
    d.cell_type = c->keep(upload(t.cell_type));
    d.pilot_val = c->keep(upload(t.pilot_val));
    d.pilot_cell = c->keep(upload(pilot_cell));
    d.constellation = c->keep(upload(cons));
    d.twiddle = c->keep(upload(tw));
    d.sym_src = c->keep(upload(t.sym_src));
    d.llr_src = c->keep(upload(t.llr_src));
    d.ls_weight = c->keep(upload(t.ls_weight));
    d.scrambler = c->keep(upload(t.scrambler));
    d.llr_dst = c->keep(upload(t.llr_dst));
    d.bit_il = c->keep(upload(t.bit_il));
    c->d_fir[0] = c->keep(upload(t.fir_time_sync));
    c->d_fir[1] = c->keep(upload(t.fir_data));
    d.tf_inv = c->keep(upload(t.tf_inv));
    d.data_cell = c->keep(upload(t.data_cell));
    d.cell_lerp = nullptr;
    if (t.mfsk_M == 0) {
        // The two pilot rows a data cell (i, j) interpolates between and their pilots' indices, tabulated from the lattice itself the way
        // interpolate_linear_col walks a column (interpolator.cc:163-254): between two measured rows the nearest one above and the nearest
        // one below; above the column's first measured row the first two, below its last one the last two (extrapolation). Every column
        // needs two pilots (a column with fewer is refused: the reference's walk degenerates there). The kernel used to derive all of this
        // per cell from divisions by 50 and 3.
        std::vector<int> pilot_of_cell(size_t(t.Nsymb) * t.Nc, -1);
        for (size_t p = 0; p < pilot_cell.size(); ++p) pilot_of_cell[pilot_cell[p]] = int(p);
        std::vector<uint32_t> tab(size_t(t.nData) * 2, 0u);
        bool ok = t.Nsymb >= 2 && t.Nsymb * t.Nc < 4096 && pilot_cell.size() < 1024 && t.Nsymb < 256;
        std::vector<std::vector<int>> col_rows(size_t(t.Nc));
        for (int i = 0; i < t.Nsymb; ++i)
            for (int j = 0; j < t.Nc; ++j) if (pilot_of_cell[size_t(i) * t.Nc + j] >= 0) col_rows[size_t(j)].push_back(i);
        for (int k = 0; k < t.nData && ok; ++k) {
            const int cell = t.data_cell[k], i = cell / t.Nc, j = cell - i * t.Nc;
            const std::vector<int>& rows = col_rows[size_t(j)];
            if (rows.size() < 2) { ok = false; break; }
            int a, b;
            if (i < rows.front()) { a = rows[0]; b = rows[1]; }
            else if (i > rows.back()) { a = rows[rows.size() - 2]; b = rows.back(); }
            else {
                const size_t hi = size_t(std::upper_bound(rows.begin(), rows.end(), i) - rows.begin());     // a data cell is never a measured row itself
                a = rows[hi - 1]; b = rows[hi];
            }
            tab[2 * size_t(k)] = uint32_t(cell) | uint32_t(pilot_of_cell[size_t(a) * t.Nc + j]) << 12 | uint32_t(pilot_of_cell[size_t(b) * t.Nc + j]) << 22;
            tab[2 * size_t(k) + 1] = uint32_t(a) | uint32_t(b) << 8 | uint32_t(i) << 16;
        }
        if (!ok) throw std::runtime_error("pilot lattice / frame geometry outside what the front-end kernel's interpolation table covers");
        d.cell_lerp = c->keep(upload(tab));
    }
    d.cptr = c->keep(upload(t.graph.cptr));
    d.cvar = c->keep(upload(t.graph.cvar));
    d.S = t.graph.S;
    d.M = t.M; d.bps = t.bps; d.K = t.K; d.P = t.P; d.N = t.N; d.E = t.graph.E;
    d.Nsymb = t.Nsymb; d.G = t.Nsymb * t.Nc; d.nData = t.nData; d.nBits = t.nBits; d.nPilots = t.nPilots;
    d.nVirtual = t.nVirtual; d.nReal = t.nReal;
    d.estimator = t.estimator; d.amp_restore = t.amp_restore; d.lsw = t.lsw;
    d.payload_bytes = t.payload_bytes; d.payload_stride = t.payload_stride; d.frame_samples = t.frame_samples;
    d.agc = c->cfg.agc; d.var_eq = c->cfg.variance_source; d.max_iters = c->cfg.max_iters;
    d.pilot_boost = t.pilot_boost;
    d.staircase = 1;
    for (int q = 0; q < t.P && d.staircase; ++q) {
        int others = 0;
        for (uint32_t e = t.graph.cptr[q]; e < t.graph.cptr[q + 1]; ++e) {
            const int v = t.graph.cvar[e];
            if (v >= t.K && v != t.K + q) { ++others; if (v != t.K + q - 1) d.staircase = 0; }
        }
        if (others != (q == 0 ? 0 : 1)) d.staircase = 0;
    }
    d.regular_lattice = 1;
    for (int r = 0; r < t.Nsymb; ++r)
        for (int q = 0; q < t.Nc; ++q)
            if ((t.cell_type[size_t(r) * t.Nc + q] != 0) != (((r - q) % 3 + 3) % 3 == 0)) d.regular_lattice = 0;
    if (t.mfsk_M == 0 && t.Nc != 50) throw std::runtime_error("the front-end kernel is specialised for 50 carriers");
    // regular_lattice == 0 (an explicit Dy other than 3: include/mercury_gpu.h mgpu_explicit_params): the estimator takes its general path - the
    // reference's own walk over the window's cells (frontend.hip) - everything else is table-driven and does not care
    if (d.regular_lattice && std::min(t.lsw / 2 + 1, t.Nc) >= 9) d.regular_lattice = 2;   // and every (clipped) window row holds >= 3 pilots of each column residue
    d.minsum_alpha = c->cfg.minsum_alpha > 0 ? c->cfg.minsum_alpha : 0.8f;
    d.mfsk_M = t.mfsk_M; d.mfsk_nbits = t.mfsk_nbits; d.mfsk_nstreams = t.mfsk_nstreams; d.mfsk_hop = t.mfsk_hop;
    d.mfsk_off0 = t.mfsk_off[0]; d.mfsk_off1 = t.mfsk_off[1];
    d.active_nsymb = t.active_nsymb; d.active_nbits = t.active_nbits; d.mfsk_amp = t.mfsk_amp;
    d.puncture_from = (c->cfg.test_puncture_nBits > 0 && c->cfg.test_puncture_nBits < t.active_nbits) ? c->cfg.test_puncture_nBits : t.active_nbits;
    LdpcDev& l = c->ldev;
    l.scrambler = d.scrambler;
    l.cptr = d.cptr; l.cvar = d.cvar;
    {   // the CRC as a sum of per-bit constants (crc16_modbus_rtu.cc:25-45 is linear over GF(2) up to the register's initial value)
        const int full = d.nReal / 8;
        std::vector<uint8_t> msg(size_t(full > 0 ? full : 1), 0);
        const uint16_t zero = mgpu::crc16_modbus(msg.data(), full);
        std::vector<uint16_t> tab(size_t(full > 0 ? full : 1) * 8, 0);
        for (int b = 0; b < full; ++b)
            for (int j = 0; j < 8; ++j) {
                msg[b] = uint8_t(1u << j);
                tab[size_t(b) * 8 + j] = uint16_t(mgpu::crc16_modbus(msg.data(), full) ^ zero);
                msg[b] = 0;
            }
        l.crc_tab = c->keep(upload(tab));
        l.crc_init = zero;
    }
    l.gdesc = c->keep(upload(t.graph.gdesc));
    l.gkpack = c->keep(upload(t.graph.gkpack));
    l.vinfo_g = c->keep(upload(t.graph.vinfo_g));
    l.Sg = t.graph.Sg;
    l.sadr = c->keep(upload(t.graph.sadr));
    l.bhead = c->keep(upload(t.graph.bhead));
    l.bmask = c->keep(upload(t.graph.bmask));
    l.vinfo2 = c->keep(upload(t.graph.vinfo2));
    l.DM = t.graph.DM;
    l.S = d.S; l.N = d.N; l.P = d.P; l.K = d.K; l.E = d.E; l.nReal = d.nReal; l.payload_stride = d.payload_stride;
    l.max_iters = d.max_iters; l.minsum_alpha = d.minsum_alpha;
    l.hard_frames = reinterpret_cast<unsigned long long*>(c->keep(upload(std::vector<uint64_t>(64, 0))));
    {   // fp64 decoder, a frame's first iterations (ldpc.hip "adaptive"): from how many odd checks on - estimated from the 16 bins a judged look
        // samples - the next iteration's posteriors (the next two iterations') are looked at inside the following check pass instead of by a
        // pass of their own. Two pairs of weights: for the look at the channel's hard decisions (the first iteration removes far more errors
        // than any later one) and for the later looks; defaults from tests/tools/unsat_profile.py and profiles/r06_ab_spec*.txt. Results do
        // not depend on them. MERCURY_SPA_SPEC_WEIGHT="first:1,first:2,later:1,later:2" for experiments (0 = always, a huge value = never).
        int w[4] = {100, 230, 45, 150};
        if (const char* e = getenv("MERCURY_SPA_SPEC_WEIGHT")) {
            const int n = sscanf(e, "%d,%d,%d,%d", &w[0], &w[1], &w[2], &w[3]);
            if (n == 1) { w[1] = w[2] = w[3] = w[0]; }
            else if (n == 2) { w[2] = w[0]; w[3] = w[1]; }
            else if (n == 3) { w[3] = w[2]; }
        }
        const int nbins = d.S / 64 > 0 ? d.S / 64 : 1;
        auto sample_min = [&](int weight) {
            const long long m = (static_cast<long long>(weight) * 16 + nbins - 1) / nbins;
            return weight <= 0 ? 0 : (m > 0x7fffff ? 0x7fffff : int(m));
        };
        auto byte = [&](int weight) { const int m = sample_min(weight); return unsigned(m > 255 ? 255 : m); };      // (a wavefront's first bin holds at most 64 checks, 16 wavefronts: "never" is any value above 1024 - 255 stands for it, see ldpc.hip)
        l.spec_sample_pack = byte(w[0]) | byte(w[1]) << 8 | byte(w[2]) << 16 | byte(w[3]) << 24;
    }
    HIPCK(hipStreamCreate(&c->stream));
    for (auto& q : c->ev) for (auto& e : q) HIPCK(hipEventCreate(&e));
    for (auto& e : c->sync_ev) HIPCK(hipEventCreate(&e));

    const bool mfsk = t.mfsk_M > 0;      // the MFSK front-end keeps no frame grid in LDS (csrc/mfsk.hip)
    {
        const char* e = getenv("MERCURY_FE_THREADS");
        const size_t lds512 = mfsk ? 0 : mgpu_frontend_lds_bytes(d.G, d.nPilots, d.nBits, 512);
        c->fe_threads = e ? atoi(e) : (lds512 > size_t(160) * 1024 / 2 ? 1024 : 512);
        if (c->fe_threads != 512 && c->fe_threads != 1024) throw std::invalid_argument("MERCURY_FE_THREADS must be 512 or 1024");
        c->lds_fe = mfsk ? 0 : mgpu_frontend_lds_bytes(d.G, d.nPilots, d.nBits, c->fe_threads);
    }
    c->lds_tx = mgpu_txgen_lds_bytes(mfsk ? 0 : d.G);
    if (!mfsk) HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(fe_kernel(c->fe_threads)), hipFuncAttributeMaxDynamicSharedMemorySize, int(c->lds_fe)));
    HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(mgpu_txgen_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(c->lds_tx)));
    {
        int geo[6];
        mgpu_tsync_fine_geometry(geo);
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(mgpu_tsync_metric_fine_kernel_r4), hipFuncAttributeMaxDynamicSharedMemorySize, geo[2]));
        HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(mgpu_tsync_metric_fine_kernel_r8), hipFuncAttributeMaxDynamicSharedMemorySize, geo[5]));
    }
    switch (c->cfg.decoder) {
        case MGPU_DEC_SPA:
            if (!t.graph.fp64_limit.empty()) throw std::runtime_error(t.graph.fp64_limit);
            c->lds_dec = mgpu_spa_lds_bytes(d.S, d.N);
            {
                const int ne = std::max(4, (d.S + 1023) / 1024);      // rounds of 16 bins; the smallest instance runs 4 (tables sized to match)
                if (ne > 8) throw std::runtime_error("graph too large for the sum-product kernel");
                if (t.graph.maxdeg > mgpu_spa_max_degree(ne)) throw std::runtime_error("check degree exceeds the sum-product kernel's unrolled product walk");
                switch (ne) {
                    case 4: c->spa_kernel = mgpu_ldpc_spa_kernel_ne4; break;
                    case 5: c->spa_kernel = mgpu_ldpc_spa_kernel_ne5; break;
                    case 6: c->spa_kernel = mgpu_ldpc_spa_kernel_ne6; break;
                    case 7: c->spa_kernel = mgpu_ldpc_spa_kernel_ne7; break;
                    default: c->spa_kernel = mgpu_ldpc_spa_kernel_ne8; break;
                }
            }
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(c->spa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(c->lds_dec)));
            break;
        case MGPU_DEC_GBF:
            c->lds_dec = mgpu_gbf_lds_bytes(d.N);
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(mgpu_ldpc_gbf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(c->lds_dec)));
            break;
        case MGPU_DEC_MINSUM: {
            if (!t.graph.fp32_limit.empty()) throw std::runtime_error(t.graph.fp32_limit);
            c->lds_dec = mgpu_spa_fast_lds_bytes(c->ldev.Sg, d.N);
            c->dec_threads = 512;
            c->spa_kernel = mgpu_ldpc_minsum_kernel_t512;
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(c->spa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(c->lds_dec)));
            break;
        }
        case MGPU_DEC_SPA_FAST: {
            if (!t.graph.fp32_limit.empty()) throw std::runtime_error(t.graph.fp32_limit);
            c->lds_dec = mgpu_spa_fast_lds_bytes(c->ldev.Sg, d.N);
            c->dec_threads = 512;            // 8 wavefronts per barrier domain: the kernel is bound by waits, not by issue (1024 threads: 1.4x slower)
            c->spa_kernel = mgpu_ldpc_spa_fast_kernel_t512;
            HIPCK(hipFuncSetAttribute(reinterpret_cast<const void*>(c->spa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(c->lds_dec)));
            break;
        }
        default: throw std::runtime_error("unknown decoder");
    }
}

// Workspaces sized by max_batch are created on first use, so a context that only ever runs e.g. the
// decoder on caller-owned device buffers (the 10^7-codeword soak) does not pin tens of GB it never touches.
void ensure_workspaces(mgpu_ctx* c, unsigned what) {
    const auto& t = c->tab;
    const size_t B = size_t(c->max_batch);
    if ((what & WS_FRONTEND) && !c->d_variance) {
        HIPCK(hipMalloc(&c->d_variance, B * sizeof(float)));
        HIPCK(hipMalloc(&c->d_snrvar, B * sizeof(float)));
        if (t.estimator == MGPU_EST_ZF) HIPCK(hipMalloc(&c->d_eqdata, B * t.nData * 16));
    }
    if ((what & WS_LLR) && !c->d_llr) HIPCK(hipMalloc(&c->d_llr, B * t.N * sizeof(float)));
    if ((what & WS_OUT) && !c->d_payload) {
        HIPCK(hipMalloc(&c->d_payload, B * t.payload_stride));
        HIPCK(hipMalloc(&c->d_stats, B * sizeof(MgpuStatsDev)));
    }
    if ((what & WS_BITS) && !c->d_bits) {
        HIPCK(hipMalloc(&c->d_bits, B * t.K));
        HIPCK(hipMalloc(&c->d_iters, B * sizeof(int)));
    }
}


void launch_frontend(mgpu_ctx* c, const double* d_bb, int F, float* d_llr, float* d_var, float* d_snrvar,
                     const MgpuTapsDev& taps, hipStream_t s, int frame_stride, int frame0) {
    const int slot = c->ev_count % mgpu_ctx::kEvRing;
    const auto& t = c->tab;
    MgpuDev dev = c->dev;                        // kernel argument; the frame stride can differ from the frame length
    if (frame_stride > 0) dev.frame_samples = frame_stride;
    const size_t stride = size_t(dev.frame_samples);
    if (c->timing) { HIPCK(hipEventRecord(c->ev[slot][0], s)); c->ev_fe[slot] = true; }
    if (t.mfsk_M > 0) {
        // MFSK modes: workgroups of (frame, run of symbols); keep gridDim * blockDim below 2^32
        const int per = mgpu_mfsk_syms_per_block(), chunks = (t.active_nsymb + per - 1) / per;
        if (!((t.mfsk_M == 32 && t.mfsk_nstreams == 1) || (t.mfsk_M == 16 && t.mfsk_nstreams == 2)) || t.mfsk_off[0] != 9 ||
            (t.mfsk_nstreams == 2 && t.mfsk_off[1] != 25) || t.Nc != 50)
            throw std::runtime_error("MFSK tone plan differs from the one the kernel is specialised for");
        const int max_frames = (1 << 23) / chunks;
        for (int off = 0; off < F; off += max_frames) {
            const int n = F - off < max_frames ? F - off : max_frames;
            if (off && (taps.grid || taps.llr_demod || taps.variance || taps.agc_gain))
                throw std::invalid_argument("stage taps are limited to one launch per call");
            hipLaunchKernelGGL(t.mfsk_M == 32 ? mgpu_mfsk_frontend_kernel_m32 : mgpu_mfsk_frontend_kernel_m16x2, dim3(unsigned(n) * chunks), dim3(256), 0, s, dev,
                               d_bb + size_t(off) * stride * 2, n, chunks, d_llr + size_t(off) * t.N, d_var + off, at(d_snrvar, off), taps);
            HIPCK(hipGetLastError());
        }
        if (c->timing) HIPCK(hipEventRecord(c->ev[slot][1], s));
        return;
    }
    for (int off = 0; off < F; off += kMaxFramesPerLaunch) {
        const int n = F - off < kMaxFramesPerLaunch ? F - off : kMaxFramesPerLaunch;
        if (off && (taps.grid || taps.H || taps.eq || taps.syms || taps.llr_demod || taps.variance || taps.agc_gain || taps.mean_H))
            throw std::invalid_argument("stage taps are limited to 2^21 frames per call");
        hipLaunchKernelGGL(fe_kernel(c->fe_threads), dim3(n), dim3(c->fe_threads), c->lds_fe, s, dev, d_bb + size_t(off) * stride * 2, n,
                           d_llr + size_t(off) * t.N, d_var + off, at(d_snrvar, off), at(c->d_eqdata, (size_t(frame0) + off) * t.nData * 2), taps);
        HIPCK(hipGetLastError());
    }
    if (c->timing) HIPCK(hipEventRecord(c->ev[slot][1], s));
}

// zero-forcing modes: SNR from the re-encoded decision (telecom_system.cc:1374-1396); needs the payload and
// the de-framed equalised symbols the front-end kept.
void select_peak(const double* cand_vals, int ncand, int step, int size, int location_to_return, int nTrials_max, int* delay, double* corr) {
    // The reference fills vals[k*step] = metric of candidate k, leaves every other entry 0, and for j = 0..nTrials_max-1
    // sets loc[j] = j and scans i = j+1..size-1 replacing (vals[j], loc[j]) by any strictly larger vals[i] — nothing is
    // swapped out, so later passes see the same maximum again. Emulated without the size-long arrays: a pass only
    // meets candidates and zeros, and a zero matters only the first time it is met while the running value is negative.
    if (location_to_return >= nTrials_max) location_to_return = nTrials_max - 1;
    const int j = location_to_return;
    auto original = [&](int i) { return (i % step == 0 && i / step < ncand) ? cand_vals[i / step] : 0.0; };
    double cur = j < size ? original(j) : 0.0;
    int loc = j;
    int p = j + 1;                                   // next index the scan visits
    for (int k = (j + step) / step; k < ncand; ++k) {               // candidates with index k*step > j
        const int ci = k * step;
        if (ci <= j) continue;
        if (p < ci && cur < 0) { cur = 0.0; loc = p; }               // a non-candidate (zero) entry comes first
        if (cand_vals[k] > cur) { cur = cand_vals[k]; loc = ci; }
        p = ci + 1;
    }
    if (p < size && cur < 0) { cur = 0.0; loc = p; }
    *delay = loc;
    *corr = cur;
}

const double* mixer_table(mgpu_ctx* c, double carrier_hz, size_t count, hipStream_t s) {
    if (c->mix_carrier == carrier_hz && c->mix_count >= count) return c->d_mix_cs;
    std::vector<double> cs(2 * count);
    const double Ts = 1.0 / kSampleRate;
    // ofdm.cc:2331-2332 evaluates cos and sin of one phase; the reference's compiler merges the pair into a single sincos() call, and glibc's
    // sincos is not bit-for-bit its cos + sin, so the same call is made here
    for (size_t i = 0; i < count; ++i) ::sincos(2 * M_PI * carrier_hz * double(int(i)) * Ts, &cs[2 * i + 1], &cs[2 * i]);
    HIPCK(hipStreamSynchronize(s));
    if (c->mix_cap < cs.size()) {
        (void)hipFree(c->d_mix_cs);
        c->d_mix_cs = nullptr; c->mix_cap = 0;
        HIPCK(hipMalloc(reinterpret_cast<void**>(&c->d_mix_cs), cs.size() * 8));
        c->mix_cap = cs.size();
    }
    HIPCK(hipMemcpy(c->d_mix_cs, cs.data(), cs.size() * 8, hipMemcpyHostToDevice));
    c->mix_carrier = carrier_hz; c->mix_count = count;
    return c->d_mix_cs;
}

// The streaming coarse kernel (sync.hip: mgpu_tsync_metric_stream_kernel) when the geometry fits it and there are enough windows to give
// every SIMD a wavefront; false = use the staged kernel. MERCURY_TSYNC_STREAM=0 / 1 forces the choice (tests compare the two).
static bool tsync_stream_launch(const double* d_bb, int stride, const int* d_start, const int* d_widx, const int* d_ncand, int ncand_max, int n, int step,
                                int pre_nsymb, int ngi_i, int nfft_i, double* d_vals, hipStream_t s, int variant) {
    static const int env = [] { const char* e = getenv("MERCURY_TSYNC_STREAM"); return e ? atoi(e) : -1; }();
    const int force = variant >= 0 ? variant : env;
    if (force == 0 || (force < 0 && n < 32)) return false;                   // measured: 16 windows 0.21 vs 0.17 ms staged, 64 windows 0.26 vs 0.34 ms
    int g[6];
    mgpu_tsync_stream_geometry(g);
    if (step != g[0] || ngi_i != g[1] || nfft_i != g[2] || pre_nsymb != g[3]) return false;      // the kernel is built for the reference's coarse search
    const int K = g[4], ring = g[5];
    const int sym = ngi_i + nfft_i, half = nfft_i / 2, PS = ngi_i + half, NP = pre_nsymb * PS;
    const int J = (NP + K - 1) / K;
    // the span the candidates in flight read, relative to step * (newest candidate): candidate `e` periods older is at pair rho + K*e
    int lo = 1 << 30, hi = -(1 << 30);
    for (int rho = 0; rho < K; rho += 4)
        for (int e = 0; e < J; ++e) {
            const int np = rho + K * e;
            if (np >= NP) break;
            const int l = np / PS, k = np % PS, a = l * sym + k, b = a + (k < ngi_i ? nfft_i : half);
            lo = std::min(lo, a - step * e);
            hi = std::max(hi, b + 4 - step * e);
        }
    if (hi - lo + 128 > ring) return false;
    // pieces of the candidate range per window: about one wavefront per SIMD (4 rings of 35 KB fit a CU's LDS), at least J candidates each
    int pieces = std::max(1, (1024 + n - 1) / n);               // ceil: 618 windows in two pieces each measured faster than one wavefront per window
    pieces = std::min(pieces, std::max(1, ncand_max / J));
    const int cpp = (ncand_max + pieces - 1) / pieces;
    pieces = (ncand_max + cpp - 1) / cpp;
    hipLaunchKernelGGL(mgpu_tsync_metric_stream_kernel, dim3(n, pieces), dim3(64), 0, s, d_bb, stride, d_start, d_widx, d_ncand, ncand_max, d_vals, lo, hi, cpp);
    return true;
}

void launch_tsync_metric(const double* d_bb, int stride, const int* d_start, const int* d_widx, const int* d_ncand, int ncand_max, int n, int step,
                         int pre_nsymb, int ngi_i, int nfft_i, double* d_vals, hipStream_t s, int variant) {
    if (ngi_i % 64 || (nfft_i / 2) % 64) {    // the staged kernels walk the preamble in chunks of 8 / 64 pairs
        hipLaunchKernelGGL(mgpu_tsync_metric_generic_kernel, dim3((ncand_max + 63) / 64, n), dim3(64), 0, s, d_bb, stride, d_start, d_widx, d_ncand,
                           ncand_max, step, pre_nsymb, ngi_i, nfft_i, d_vals);
    } else if (step == 1 && ngi_i == 64 * (ngi_i / 64) && (variant > 0 || (variant < 0 && n >= 32))) {
        // fine search over many windows: R adjacent candidates per lane share every sample's products (sync.hip). variant 1: R = 4, 2: R = 8;
        // -1 picks by the number of windows (ms per launch of 4352 candidates, dense / R = 4 / R = 8: 16 windows 0.18 / 0.18 / 0.29,
        // 64: 0.40 / 0.33 / 0.31, 256: 1.24 / 0.81 / 0.84, 1024: 5.13 / 2.91 / 2.71). A few windows keep the dense kernel (one candidate
        // per lane: four times the wavefronts, a quarter of the latency).
        if (variant < 0) variant = n >= 512 ? 2 : 1;
        int geo[6];
        mgpu_tsync_fine_geometry(geo);                               // the kernels' LDS limits are raised per device in mgpu_create
        const int g = variant == 2 ? 3 : 0;
        hipLaunchKernelGGL(variant == 2 ? mgpu_tsync_metric_fine_kernel_r8 : mgpu_tsync_metric_fine_kernel_r4, dim3((ncand_max + geo[g] - 1) / geo[g], n),
                           dim3(geo[g + 1]), size_t(geo[g + 2]), s, d_bb, stride, d_start, d_widx, d_ncand, ncand_max, pre_nsymb, ngi_i, nfft_i, d_vals);
    } else if (step > 4 && tsync_stream_launch(d_bb, stride, d_start, d_widx, d_ncand, ncand_max, n, step, pre_nsymb, ngi_i, nfft_i, d_vals, s, variant)) {
        // many windows: one wavefront streams each (piece of a) window through an LDS ring, see sync.hip
    } else {
        // The coarse search re-reads every sample ~44 times (overlapping candidates) and is bound by that traffic. Launching it over
        // 64 windows at a time keeps the windows in flight (95 MB) inside the 256 MB Infinity Cache instead of streaming 1.5 GB per
        // 1024 windows from HBM: 10.4 -> 7.1 ms per 1024 windows (MERCURY_TSYNC_SLICE overrides; 0 = one launch).
        static const int slice = [] { const char* e = getenv("MERCURY_TSYNC_SLICE"); return e ? atoi(e) : 64; }();
        const int per = (slice > 0 && step > 4) ? slice : n;
        for (int off = 0; off < n; off += per) {
            const int m = std::min(per, n - off);
            const int threads = step <= 4 ? 256 : mgpu_tsync_coarse_threads();
            const int nblk = (ncand_max + threads - 1) / threads;
            // coarse kernel: windows along x, so that with a multiple of 8 windows per launch all workgroups of a window land on one XCD
            hipLaunchKernelGGL(step <= 4 ? mgpu_tsync_metric_dense_kernel : mgpu_tsync_metric_kernel, step <= 4 ? dim3(nblk, m) : dim3(m, nblk), dim3(threads), 0, s,
                               d_widx ? d_bb : d_bb + size_t(off) * stride * 2, stride, at(d_start, size_t(off)), at(d_widx, size_t(off)), at(d_ncand, size_t(off)),
                               ncand_max, step, pre_nsymb, ngi_i, nfft_i, d_vals + size_t(off) * ncand_max);
        }
    }
    HIPCK(hipGetLastError());
}

// -1: pick (sliding-tap kernels where they apply), 0: always the generic kernel, 1: as -1; test hook mgpu_debug_p2b_variant
static std::atomic<int> g_p2b_variant{-1};
extern "C" int mgpu_debug_p2b_variant(int v) { const int old = g_p2b_variant.exchange(v); return old; }

void launch_p2b(const double* passband, int in_size, const double* d_carrier, const int* d_start, int start_all, int count, int decim, const double* d_taps,
                int ntaps, double* out, const int* widx, const double* cs, const int* out_row, int row_by_launch, int nwin, hipStream_t s) {
    constexpr double kFs = 48000.0, kAmp = 1.4142135623730951;
    if (g_p2b_variant.load() != 0 && ntaps == 33 && (decim == 1 || decim == 4)) {
        int geo[6];
        mgpu_p2b_slide_geometry(geo);
        const int g = decim == 1 ? 0 : 3;
        auto kernel = decim == 1 ? (cs ? mgpu_p2b_slide_d1_kernel : mgpu_p2b_slide_d1_sincos_kernel) : (cs ? mgpu_p2b_slide_d4_kernel : mgpu_p2b_slide_d4_sincos_kernel);
        hipLaunchKernelGGL(kernel, dim3((count + geo[g] - 1) / geo[g], unsigned(nwin)), dim3(geo[g + 1]),
                           size_t(geo[g + 2]), s, passband, in_size, d_carrier, d_start, start_all, count, d_taps, kFs, kAmp, out, widx, cs, out_row, row_by_launch);
    } else {
        const size_t lds = size_t(255 * decim + ntaps) * 16;
        need(lds <= 64 * 1024 && ntaps <= 64, "decimation too large for the staging buffer");
        hipLaunchKernelGGL(mgpu_p2b_kernel, dim3((count + 255) / 256, unsigned(nwin)), dim3(256), lds, s, passband, in_size, d_carrier, d_start, start_all, count, decim,
                           d_taps, ntaps, kFs, kAmp, out, widx, cs, out_row, row_by_launch);
    }
    HIPCK(hipGetLastError());
}

static constexpr int kTones32[4] = {4, 20, 12, 28}, kTones16[4] = {2, 10, 6, 14};      // mfsk.cc:82-95

// the same search on the device, for the energies of W windows lying in d_energy ([W][nslots][Nc]); d_search_start: [W] or null
void launch_mfsk_sync(mgpu_ctx* c, const double* d_energy, int W, int nslots, int size, const int* d_search_start, int* d_delay, hipStream_t s) {
    const auto& t = c->tab;
    need(t.preamble >= 1 && t.preamble <= 4 && t.mfsk_nstreams >= 1 && t.mfsk_nstreams <= 4, "MFSK preamble / stream count outside the kernel's tables");
    MgpuMfskSync P{};
    P.np = t.preamble; P.nstreams = t.mfsk_nstreams; P.Nc = t.Nc; P.sym_period = t.Nofdm * 4; P.tail = t.Ngi * 4 + t.Nfft * 4;
    for (int i = 0; i < 4; ++i) P.off[i] = t.mfsk_off[i];
    for (int p = 0; p < P.np; ++p) P.tones[p] = (t.mfsk_M == 32 ? kTones32 : kTones16)[p % P.np];
    hipLaunchKernelGGL(mgpu_mfsk_sync_kernel, dim3(W), dim3(256), size_t(P.np) * nslots * 8, s, d_energy, nslots, size, P, d_search_start, d_delay);
    HIPCK(hipGetLastError());
}

int mfsk_sync_from_energies(const mgpu::ModeTables& t, const double* E, int nslots, int size, int search_start_symb) {
    const int* tones = t.mfsk_M == 32 ? kTones32 : kTones16;
    const int sym_period = t.Nofdm * 4, np = t.preamble;
    double best_metric = -1;
    int best = 0;
    for (int s = search_start_symb > 0 ? search_start_symb : 0; s <= nslots - np; ++s) {
        double metric = 0;
        for (int p = 0; p < np; ++p) {
            if ((s + p) * sym_period + t.Ngi * 4 + t.Nfft * 4 > size) break;
            const double* e = &E[size_t(s + p) * t.Nc];
            double e_target = 0;
            for (int st = 0; st < t.mfsk_nstreams; ++st) e_target += e[t.mfsk_off[st] + tones[p % np]];
            double e_total = 0;
            for (int k = 0; k < t.Nc; ++k) e_total += e[k];
            if (e_total > 0) metric += e_target / e_total;
        }
        if (metric > best_metric) { best_metric = metric; best = s; }
    }
    return best * sym_period;
}

void launch_zf_snr(mgpu_ctx* c, int F, const uint8_t* d_payload, MgpuStatsDev* d_stats, hipStream_t s, int frame0, double* d_var_out) {
    const auto& t = c->tab;
    if (t.estimator != MGPU_EST_ZF || !d_payload || !d_stats) return;
    for (int off = 0; off < F; off += kMaxFramesPerLaunch) {
        const int n = F - off < kMaxFramesPerLaunch ? F - off : kMaxFramesPerLaunch;
        hipLaunchKernelGGL(mgpu_zf_snr_kernel, dim3(n), dim3(256), mgpu_zfsnr_lds_bytes(t.nData), s, c->dev,
                           d_payload + size_t(off) * t.payload_stride, c->d_eqdata + (size_t(frame0) + off) * t.nData * 2, n, d_stats + off,
                           d_var_out ? d_var_out + off : nullptr);
        HIPCK(hipGetLastError());
    }
}

void launch_decoder(mgpu_ctx* c, const float* d_llr, int F, uint8_t* d_bits, int* d_iters, uint8_t* d_payload,
                    MgpuStatsDev* d_stats, const float* d_var, const float* d_snrvar, hipStream_t s) {
    const int slot = c->ev_count % mgpu_ctx::kEvRing;
    const auto& t = c->tab;
    if (c->timing) HIPCK(hipEventRecord(c->ev[slot][2], s));
    for (int off = 0; off < F; off += kMaxFramesPerLaunch) {
        const int n = F - off < kMaxFramesPerLaunch ? F - off : kMaxFramesPerLaunch;
        const float* llr = d_llr + size_t(off) * t.N;
        uint8_t* bits = at(d_bits, size_t(off) * t.K);
        int* iters = at(d_iters, off);
        uint8_t* pay = at(d_payload, size_t(off) * t.payload_stride);
        MgpuStatsDev* st = at(d_stats, off);
        const float* var = at(d_var, off);
        const float* sv = at(d_snrvar, off);
        if (c->cfg.decoder == MGPU_DEC_GBF)
            hipLaunchKernelGGL(mgpu_ldpc_gbf_kernel, dim3(n), dim3(1024), c->lds_dec, s, c->ldev, llr, n, bits, iters, pay, st, var, sv);
        else   // sum-product or min-sum, the variant for this graph's round count
            hipLaunchKernelGGL(c->spa_kernel, dim3(n), dim3(c->dec_threads), c->lds_dec, s, c->ldev, llr, n, bits, iters, pay, st, var, sv);
        HIPCK(hipGetLastError());
    }
    if (c->timing) { HIPCK(hipEventRecord(c->ev[slot][3], s)); ++c->ev_count; c->ev_fe[c->ev_count % mgpu_ctx::kEvRing] = false; }
}

}  // namespace mgpu_detail

extern "C" {

int mgpu_create(const mgpu_config* cfg, mgpu_ctx** out) { return mgpu_create_explicit(cfg, nullptr, out); }

extern "C" void mgpu_internal_libm_notice();     // libm_check.cpp

int mgpu_create_explicit(const mgpu_config* cfg, const mgpu_explicit_params* xp_in, mgpu_ctx** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return MGPU_ERR_ARG; }
    *out = nullptr;
    mgpu_internal_libm_notice();                 // only with MERCURY_GPU_LIBM_CHECK=1: is the host's libm the one the device restates? (once per process; stderr only if not)
    mgpu::ExplicitParams xp;
    if (xp_in) {
        // the kernels are specialised for the reference's carrier count, transform length and pilot column step: those fields only confirm them
        if ((xp_in->Nc != 0 && xp_in->Nc != 50) || (xp_in->Nfft != 0 && xp_in->Nfft != 256) || (xp_in->Dx != 0 && xp_in->Dx != 1)) {
            g_create_error = "explicit parameters: Nc / Nfft / Dx other than 50 / 256 / 1 are not supported (the kernels are specialised for them)";
            return MGPU_ERR_UNSUPPORTED;
        }
        if (xp_in->Dy < 0 || xp_in->Dy > 255 || xp_in->Nsymb < 0 || xp_in->Nsymb > 255) { g_create_error = "explicit parameters: Dy / Nsymb must be 0 (the reference's default) .. 255"; return MGPU_ERR_ARG; }
        if (xp_in->Dy != 0) xp.Dy = xp_in->Dy;
        xp.Nsymb = xp_in->Nsymb;
        if (xp_in->pilot_boost != 0.0f) xp.pilot_boost = xp_in->pilot_boost;
        if (xp_in->ls_window != 0) xp.ls_window = xp_in->ls_window;
        if (xp_in->ls_window < 0 || xp_in->ls_window > 21) { g_create_error = "explicit parameters: ls_window must be 1..21 (0 = the reference's 20)"; return MGPU_ERR_ARG; }
        if (!(xp.pilot_boost > 0.0f) || !(xp.pilot_boost < 1e6f)) { g_create_error = "explicit parameters: pilot_boost must be positive and finite (0 = the reference's 1.33)"; return MGPU_ERR_ARG; }
        if (xp_in->seeds_set) { xp.pilot_seed = xp_in->pilot_seed; xp.scrambler_seed = xp_in->scrambler_seed; xp.preamble_seed = xp_in->preamble_seed; }
    }
    int em, er, ep, ee;
    if (!((cfg->cfg >= 0 && cfg->cfg <= 16) || (cfg->cfg >= 100 && cfg->cfg <= 102) || mgpu::explicit_mode_row(cfg->cfg, &em, &er, &ep, &ee))) {
        g_create_error = "cfg must be 0..16 (OFDM modes), 100..102 (ROBUST MFSK modes) or an MGPU_CFG_EXPLICIT id";
        return MGPU_ERR_ARG;
    }
    if (cfg->max_iters < 1 || cfg->max_iters > 1000) { g_create_error = "max_iters out of range"; return MGPU_ERR_ARG; }
    if (cfg->decoder < 0 || cfg->decoder > MGPU_DEC_SPA_FAST) { g_create_error = "unknown decoder"; return MGPU_ERR_ARG; }
    if (cfg->max_batch < 1) { g_create_error = "max_batch must be >= 1"; return MGPU_ERR_ARG; }
    if (cfg->test_puncture_nBits < 0) { g_create_error = "test_puncture_nBits must be >= 0"; return MGPU_ERR_ARG; }
    mgpu_ctx* c = new mgpu_ctx();
    c->cfg = *cfg;
    c->max_batch = cfg->max_batch;
    try {
        std::vector<uint8_t> file_blob;
        const uint8_t* blob = mgpu_ldpc_blob;
        size_t blob_size = mgpu_ldpc_blob_size;
        if (const char* p = std::getenv("MERCURY_LDPC_TABLES")) {
            std::ifstream f(p, std::ios::binary);
            if (!f) throw std::runtime_error(std::string("cannot open ") + p);
            file_blob.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
            blob = file_blob.data();
            blob_size = file_blob.size();
        }
        try {
            c->tab = mgpu::build_mode_tables(cfg->cfg, cfg->mfsk_ctrl_mode, blob, blob_size, xp);
        } catch (const std::exception& e) {
            g_create_error = e.what();
            delete c;
            return MGPU_ERR_TABLES;
        }
        int ndev = 0;
        HIPCK(hipGetDeviceCount(&ndev));
        if (ndev < 1) throw HipError("no HIP device visible (the MI355X path has no CPU fallback)");
        HIPCK(hipSetDevice(cfg->device));
        c->numa_node = mgpu_detail::device_numa_node(cfg->device);      // where the context's page-locked staging lives (-1: anywhere)
        mgpu_detail::ctx_alloc(c);
    } catch (const std::exception& e) {
        g_create_error = e.what();
        mgpu_destroy(c);
        return MGPU_ERR_DEVICE;
    }
    *out = c;
    return MGPU_OK;
}

void mgpu_destroy(mgpu_ctx* c) {
    if (!c) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != c->cfg.device && hipSetDevice(c->cfg.device) != hipSuccess) prev = -1;   // no device: nothing below was allocated
    for (void* p : c->owned) (void)hipFree(p);
    (void)hipFree(c->d_baseband); (void)hipFree(c->d_llr); (void)hipFree(c->d_variance); (void)hipFree(c->d_snrvar);
    (void)hipFree(c->d_payload); (void)hipFree(c->d_stats); (void)hipFree(c->d_bits); (void)hipFree(c->d_iters); (void)hipFree(c->d_eqdata);
    if (c->rxloop_ws && c->rxloop_ws_free) c->rxloop_ws_free(c->rxloop_ws);
    (void)hipFree(c->rb_stage);
    (void)hipFree(c->rb_compact);
    if (c->rb_stream) (void)hipStreamDestroy(c->rb_stream);
    if (c->tx_state && c->tx_state_free) c->tx_state_free(c->tx_state);
    (void)hipFree(c->d_mix_cs);
    if (c->one_frame_graph) (void)hipGraphExecDestroy(c->one_frame_graph);
    if (c->h_one_in) (void)hipHostFree(c->h_one_in);
    if (c->h_one_out) (void)hipHostFree(c->h_one_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (auto& q : c->ev) for (auto& e : q) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->sync_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->hp_ev) if (e) (void)hipEventDestroy(e);
    (void)hipFree(c->d_one_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    for (auto& p : c->pipe) {
        (void)hipFree(p.d_in);
        if (p.stream) (void)hipStreamDestroy(p.stream);
        if (p.done) (void)hipEventDestroy(p.done);
        if (p.copied) (void)hipEventDestroy(p.copied);
    }
    if (prev >= 0 && prev != c->cfg.device) (void)hipSetDevice(prev);
    delete c;
}

void* mgpu_alloc_host(size_t bytes) {
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

int mgpu_device_props_get(int device, mgpu_device_props* out) {
    if (!out) return MGPU_ERR_ARG;
    std::memset(out, 0, sizeof(*out));
    out->numa_node = -1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return MGPU_ERR_DEVICE;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) return MGPU_ERR_DEVICE;
    out->compute_units = p.multiProcessorCount;
    out->clock_khz = p.clockRate;
    out->memory_clock_khz = p.memoryClockRate;
    out->lds_bytes_per_cu = int(p.maxSharedMemoryPerMultiProcessor);
    out->wavefront_size = p.warpSize;
    out->hbm_bytes = p.totalGlobalMem;
    std::snprintf(out->name, sizeof(out->name), "%s", p.name);
    std::snprintf(out->gcn_arch, sizeof(out->gcn_arch), "%s", p.gcnArchName);
    if (hipDeviceGetPCIBusId(out->pci_bus_id, int(sizeof(out->pci_bus_id)), device) != hipSuccess) out->pci_bus_id[0] = 0;
    out->numa_node = mgpu_host_numa_node_of_pci(out->pci_bus_id);
    return MGPU_OK;
}

extern "C++" {
namespace mgpu_detail {
// page-locked host memory on `node` (the GPU's NUMA node) when there is one: hipHostMallocNumaUser makes the runtime honour the calling
// thread's memory policy, which prefers that node while the allocation (and the page-locking first touch) runs
hipError_t host_alloc_on_node(void** p, size_t bytes, int node) {
    if (node >= 0) {
        mgpu_numa::PreferNode scope(node);
        if (scope.active() && hipHostMalloc(p, bytes, hipHostMallocNumaUser) == hipSuccess) return hipSuccess;
        (void)hipGetLastError();
    }
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}
int device_numa_node(int device) {
    char id[32] = {0};
    if (hipDeviceGetPCIBusId(id, int(sizeof(id)), device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return mgpu_host_numa_node_of_pci(id);
}
}  // namespace mgpu_detail
}  // extern "C++"

void* mgpu_alloc_host_near(int device, size_t bytes) {
    void* p = nullptr;
    return mgpu_detail::host_alloc_on_node(&p, bytes ? bytes : 16, mgpu_detail::device_numa_node(device)) == hipSuccess ? p : nullptr;
}
void mgpu_free_host(void* p) { if (p) (void)hipHostFree(p); }

const char* mgpu_last_error(mgpu_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

static void fill_info(const mgpu::ModeTables& t, mgpu_info* i) {
    i->cfg = t.cfg; i->M = t.M; i->bits_per_symbol = t.bps; i->K = t.K; i->P = t.P; i->N = t.N;
    i->Nsymb = t.Nsymb; i->Nc = t.Nc; i->Nfft = t.Nfft; i->Ngi = t.Ngi; i->Nofdm = t.Nofdm;
    i->nData = t.nData; i->nBits = t.nBits; i->nPilots = t.nPilots; i->nVirtual = t.nVirtual; i->nReal = t.nReal;
    i->bit_blk = t.bit_blk; i->tf_blk = t.tf_blk; i->preamble_nsymb = t.preamble;
    i->estimator = t.estimator; i->amp_restore = t.amp_restore; i->ls_window = t.lsw;
    i->Cwidth = t.graph.Cwidth; i->Vwidth = t.graph.Vwidth; i->E = t.graph.E;
    i->payload_bytes = t.payload_bytes; i->payload_stride = t.payload_stride; i->frame_samples = t.frame_samples;
    i->mfsk_M = t.mfsk_M; i->mfsk_nStreams = t.mfsk_nstreams; i->active_nsymb = t.active_nsymb; i->active_nbits = t.active_nbits;
}

int mgpu_get_info(mgpu_ctx* c, mgpu_info* i) {
    if (!c || !i) return MGPU_ERR_ARG;
    fill_info(c->tab, i);
    return MGPU_OK;
}

// load_configuration's mode row + derived sizes (telecom_system.cc:2487-3025, :1818-1826, :2910-2911) without a device: the same table builder
// mgpu_create runs, so the CPU test suite can hold it against the reference's printed values (tests/golden/survey_mode_table.json)
int mgpu_host_mode_info(int cfg, int mfsk_ctrl_mode, mgpu_info* i) {
    if (!i) return MGPU_ERR_ARG;
    try {
        const mgpu::ModeTables m = mgpu::build_mode_tables(cfg, mfsk_ctrl_mode, mgpu_ldpc_blob, mgpu_ldpc_blob_size);
        fill_info(m, i);
        return MGPU_OK;
    } catch (const std::exception& e) { g_create_error = e.what(); return MGPU_ERR_ARG; }
}

// the fp32 decoders' bank-aware placement, as modelled on the host (tables.cpp): LDS cycles per 32-lane gather group, 1.0 = conflict-free;
// out[0] the check pass's posterior reads, out[1] the variable update's message reads, out[2] bins, out[3] slots in use / slots
int mgpu_host_layout_stats(int cfg, double out[4]) {
    if (!out) return MGPU_ERR_ARG;
    try {
        const mgpu::ModeTables m = mgpu::build_mode_tables(cfg, 0, mgpu_ldpc_blob, mgpu_ldpc_blob_size);
        out[0] = m.graph.bank_model[0]; out[1] = m.graph.bank_model[1];
        out[2] = m.graph.Sg / 64; out[3] = m.graph.Sg ? double(m.graph.E) / m.graph.Sg : 0;
        return MGPU_OK;
    } catch (const std::exception& e) { g_create_error = e.what(); return MGPU_ERR_ARG; }
}

int mgpu_enable_timing(mgpu_ctx* c, int on) {
    if (!c) return MGPU_ERR_ARG;
    c->timing = on != 0;
    c->ev_count = 0;
    for (auto& b : c->ev_fe) b = false;
    return MGPU_OK;
}

int mgpu_decoder_hard_frames(mgpu_ctx* c, long long* frames) {
    if (!c || !frames) return MGPU_ERR_ARG;
    return guard(c, [&] {
        HIPCK(hipStreamSynchronize(c->stream));
        unsigned long long v[64];
        HIPCK(hipMemcpy(v, c->ldev.hard_frames, sizeof(v), hipMemcpyDeviceToHost));
        long long sum = 0;
        for (unsigned long long x : v) sum += (long long)x;
        *frames = sum;
    });
}

int mgpu_last_kernel_ms(mgpu_ctx* c, float ms[2]) {
    int n = 0;
    return mgpu_kernel_ms_avg(c, ms, &n);
}

int mgpu_kernel_ms_avg(mgpu_ctx* c, float ms[2], int* n_launches) {
    if (!c || !ms) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(c->timing && c->ev_count > 0, "timing not enabled or nothing launched");
        const int n = c->ev_count < mgpu_ctx::kEvRing ? c->ev_count : mgpu_ctx::kEvRing;
        double fe = 0, dec = 0;
        int nfe = 0;
        for (int i = 0; i < n; ++i) {
            HIPCK(hipEventSynchronize(c->ev[i][3]));
            float t = 0;
            HIPCK(hipEventElapsedTime(&t, c->ev[i][2], c->ev[i][3]));
            dec += t;
            if (c->ev_fe[i]) { HIPCK(hipEventElapsedTime(&t, c->ev[i][0], c->ev[i][1])); fe += t; ++nfe; }
        }
        ms[0] = nfe ? float(fe / nfe) : 0.f;
        ms[1] = float(dec / n);
        if (n_launches) *n_launches = n;
    });
}

int mgpu_frontend_dev(mgpu_ctx* c, const void* d_bb, int F, void* d_llr, void* d_variance_f, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(d_bb && d_llr && F >= 0 && F <= c->max_batch, "bad argument (F must be <= max_batch)");
        if (F == 0) return;
        ensure_workspaces(c, WS_FRONTEND);
        MgpuTapsDev taps{};
        launch_frontend(c, static_cast<const double*>(d_bb), F, static_cast<float*>(d_llr),
                        d_variance_f ? static_cast<float*>(d_variance_f) : c->d_variance, c->d_snrvar, taps, static_cast<hipStream_t>(stream));
    });
}

int mgpu_ldpc_batch_dev(mgpu_ctx* c, const void* d_llr, int F, void* d_bits, void* d_iters, void* d_payload,
                        void* d_stats, const void* d_variance_f, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(d_llr && F >= 0 && F <= c->max_batch, "bad argument (F must be <= max_batch)");
        if (F == 0) return;
        launch_decoder(c, static_cast<const float*>(d_llr), F, static_cast<uint8_t*>(d_bits), static_cast<int*>(d_iters),
                       static_cast<uint8_t*>(d_payload), static_cast<MgpuStatsDev*>(d_stats),
                       static_cast<const float*>(d_variance_f), nullptr, static_cast<hipStream_t>(stream));
    });
}

int mgpu_rx_batch_dev(mgpu_ctx* c, const void* d_bb, int F, void* d_payload, void* d_stats, void* d_llr_opt, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(d_bb && d_payload && d_stats && F >= 0 && F <= c->max_batch, "bad argument (F must be <= max_batch)");
        if (F == 0) return;
        hipStream_t s = static_cast<hipStream_t>(stream);
        ensure_workspaces(c, WS_FRONTEND | (d_llr_opt ? 0u : unsigned(WS_LLR)));
        float* llr = d_llr_opt ? static_cast<float*>(d_llr_opt) : c->d_llr;
        MgpuTapsDev taps{};
        launch_frontend(c, static_cast<const double*>(d_bb), F, llr, c->d_variance, c->d_snrvar, taps, s);
        launch_decoder(c, llr, F, nullptr, nullptr, static_cast<uint8_t*>(d_payload), static_cast<MgpuStatsDev*>(d_stats),
                       c->d_variance, c->d_snrvar, s);
        launch_zf_snr(c, F, static_cast<uint8_t*>(d_payload), static_cast<MgpuStatsDev*>(d_stats), s);
    });
}

void* mgpu_device_malloc(mgpu_ctx* c, size_t bytes) {
    if (!c) return nullptr;
    void* p = nullptr;
    const int rc = guard(c, [&] { HIPCK(hipMalloc(&p, bytes ? bytes : 1)); });
    return rc == MGPU_OK ? p : nullptr;
}
void mgpu_device_free(mgpu_ctx* c, void* d_ptr) {
    if (c && d_ptr) (void)guard(c, [&] { HIPCK(hipFree(d_ptr)); });
}
void* mgpu_context_stream(mgpu_ctx* c) { return c ? static_cast<void*>(c->stream) : nullptr; }
int mgpu_synchronize(mgpu_ctx* c, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] { HIPCK(hipStreamSynchronize(stream ? static_cast<hipStream_t>(stream) : c->stream)); });
}
int mgpu_copy_to_host(mgpu_ctx* c, void* dst, const void* d_src, size_t bytes, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need((dst && d_src) || bytes == 0, "bad argument");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
        if (bytes) HIPCK(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}
int mgpu_copy_to_device(mgpu_ctx* c, void* d_dst, const void* src, size_t bytes, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need((d_dst && src) || bytes == 0, "bad argument");
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
        if (bytes) HIPCK(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_txgen_dev(mgpu_ctx* c, uint64_t seed, uint64_t frame0, int F, double noise_amp, int channel, void* d_bb,
                   void* d_payload_opt, void* stream) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(d_bb && F >= 0 && (channel == 0 || channel == 1), "bad argument");
        if (F == 0) return;
        for (int off = 0; off < F; off += kMaxFramesPerLaunch) {
            const int n = F - off < kMaxFramesPerLaunch ? F - off : kMaxFramesPerLaunch;
            hipLaunchKernelGGL(mgpu_txgen_kernel, dim3(n), dim3(256), c->lds_tx, static_cast<hipStream_t>(stream), c->dev, seed,
                               frame0 + uint64_t(off), n, noise_amp, channel, static_cast<double*>(d_bb) + size_t(off) * c->tab.frame_samples * 2,
                               at(static_cast<uint8_t*>(d_payload_opt), size_t(off) * c->tab.payload_stride),
                               static_cast<const uint8_t*>(nullptr), 0, static_cast<const int*>(nullptr), 0, 0);
            HIPCK(hipGetLastError());
        }
    });
}

// ---- synchroniser building blocks (host-buffer, blocking) ---------------------------------------
int mgpu_passband_to_baseband(mgpu_ctx* c, const double* passband, int W, int in_size, const double* carrier_hz, int filter,
                              const int* start, int count, int decimation, double* out_c128) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(passband && carrier_hz && out_c128 && W > 0 && in_size > 0 && count > 0 && decimation >= 1 && (filter == 0 || filter == 1),
             "bad argument");
        const auto& taps = filter ? c->tab.fir_data : c->tab.fir_time_sync;
        DevBuf d_in(size_t(W) * in_size * 8), d_fc(size_t(W) * 8), d_out(size_t(W) * count * 16), d_start(size_t(W) * 4);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_in.p, passband, size_t(W) * in_size * 8, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_fc.p, carrier_hz, size_t(W) * 8, hipMemcpyHostToDevice, s));
        if (start) HIPCK(hipMemcpyAsync(d_start.p, start, size_t(W) * 4, hipMemcpyHostToDevice, s));
        const int ntaps = int(taps.size());
        bool shared = true;                                           // one carrier for every window: the host-libm mixer table applies
        for (int w = 1; w < W; ++w) shared = shared && carrier_hz[w] == carrier_hz[0];
        const double* cs = shared ? mixer_table(c, carrier_hz[0], size_t(in_size), s) : nullptr;
        HIPCK(hipEventRecord(c->sync_ev[0], s));
        launch_p2b(d_in.as<double>(), in_size, d_fc.as<double>(), start ? d_start.as<int>() : nullptr, 0, count, decimation, c->d_fir[filter], ntaps,
                   d_out.as<double>(), nullptr, cs, nullptr, 0, W, s);
        HIPCK(hipEventRecord(c->sync_ev[1], s));
        HIPCK(hipMemcpyAsync(out_c128, d_out.p, size_t(W) * count * 16, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_time_sync_preamble(mgpu_ctx* c, const double* bb, int W, int size, int step, int location_to_return, int nTrials_max,
                            int* delay, double* correlation) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        const auto& t = c->tab;
        const int interp = 4, sym = t.Nofdm * interp, L = t.preamble * sym;
        need(bb && delay && W > 0 && size > L && step >= 1 && nTrials_max >= 1 && nTrials_max <= size, "bad argument");
        const int ncand = (size - L + step - 1) / step;
        DevBuf d_in(size_t(W) * size * 16), d_vals(size_t(W) * ncand * 8);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_in.p, bb, size_t(W) * size * 16, hipMemcpyHostToDevice, s));
        HIPCK(hipEventRecord(c->sync_ev[0], s));
        const int ngi_i = t.Ngi * interp, nfft_i = t.Nfft * interp;
        launch_tsync_metric(d_in.as<double>(), size, nullptr, nullptr, nullptr, ncand, W, step, t.preamble, ngi_i, nfft_i, d_vals.as<double>(), s);
        HIPCK(hipGetLastError());
        HIPCK(hipEventRecord(c->sync_ev[1], s));
        std::vector<double> cand(size_t(W) * ncand);
        HIPCK(hipMemcpyAsync(cand.data(), d_vals.p, cand.size() * 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        for (int w = 0; w < W; ++w) {
            double corr = 0;
            select_peak(&cand[size_t(w) * ncand], ncand, step, size, location_to_return, nTrials_max, &delay[w], &corr);
            if (correlation) correlation[w] = corr;
        }
    });
}

int mgpu_baseband_test_esn0(mgpu_ctx* c, const double* esn0_db, int npoints, long long frames_per_point, uint64_t seed, uint64_t frame0, int channel,
                            mgpu_error_rate* out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(esn0_db && out && npoints > 0 && frames_per_point > 0 && (channel == 0 || channel == 1), "bad argument");
        const auto& t = c->tab;
        const int B = int(std::min<long long>(frames_per_point, c->max_batch));
        ensure_workspaces(c, WS_FRONTEND | WS_LLR | WS_OUT);
        DevBuf d_bb(size_t(B) * t.frame_samples * 16), d_sent(size_t(B) * t.payload_stride), d_acc(4 * 8);
        hipStream_t s = c->stream;
        for (int p = 0; p < npoints; ++p) {
            const double noise_amp = std::pow(10.0, -esn0_db[p] / 20.0) / std::sqrt(2.0);      // per component, telecom_system.cc:100,147
            HIPCK(hipMemsetAsync(d_acc.p, 0, 32, s));
            for (long long done = 0; done < frames_per_point; done += B) {
                const int n = int(std::min<long long>(B, frames_per_point - done));
                const uint64_t first = frame0 + uint64_t(p) * uint64_t(frames_per_point) + uint64_t(done);
                for (int off = 0; off < n; off += kMaxFramesPerLaunch) {
                    const int m = std::min(n - off, kMaxFramesPerLaunch);
                    hipLaunchKernelGGL(mgpu_txgen_kernel, dim3(m), dim3(256), c->lds_tx, s, c->dev, seed, first + uint64_t(off), m, noise_amp, channel,
                                       d_bb.as<double>() + size_t(off) * t.frame_samples * 2, d_sent.as<uint8_t>() + size_t(off) * t.payload_stride,
                                       static_cast<const uint8_t*>(nullptr), 0, static_cast<const int*>(nullptr), 0, 0);
                    HIPCK(hipGetLastError());
                }
                MgpuTapsDev taps{};
                launch_frontend(c, d_bb.as<double>(), n, c->d_llr, c->d_variance, c->d_snrvar, taps, s);
                launch_decoder(c, c->d_llr, n, nullptr, nullptr, c->d_payload, c->d_stats, c->d_variance, c->d_snrvar, s);
                hipLaunchKernelGGL(mgpu_error_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_sent.as<uint8_t>(), c->d_payload, c->d_stats,
                                   t.payload_stride, t.nReal, n, d_acc.as<unsigned long long>());
                HIPCK(hipGetLastError());
            }
            unsigned long long acc[4];
            HIPCK(hipMemcpyAsync(acc, d_acc.p, 32, hipMemcpyDeviceToHost, s));
            HIPCK(hipStreamSynchronize(s));
            mgpu_error_rate& r = out[p];
            r.esn0_db = esn0_db[p];
            r.Frames_total = frames_per_point; r.Error_frames_total = (long long)acc[1];
            r.Bits_total = frames_per_point * t.nReal; r.Error_bits_total = (long long)acc[0];
            r.BER = double(r.Error_bits_total) / double(r.Bits_total);
            r.FER = double(r.Error_frames_total) / double(r.Frames_total);
            r.avg_iterations = double(acc[2]) / double(frames_per_point);
            r.crc_ok_frames = (long long)acc[3];
        }
    });
}

int mgpu_debug_select_peak(mgpu_ctx* c, const double* cand_vals, int n, int ncand_max, const int* ncand, const int* size, const int* loc, int step,
                           int nTrials_max, int* delay, double* corr) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(cand_vals && ncand && size && loc && delay && corr && n > 0 && ncand_max > 0 && step >= 1 && nTrials_max >= 1, "bad argument");
        DevBuf d_v(size_t(n) * ncand_max * 8), d_nc(size_t(n) * 4), d_sz(size_t(n) * 4), d_lc(size_t(n) * 4), d_d(size_t(n) * 4), d_c(size_t(n) * 8);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_v.p, cand_vals, size_t(n) * ncand_max * 8, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_nc.p, ncand, size_t(n) * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_sz.p, size, size_t(n) * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_lc.p, loc, size_t(n) * 4, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(mgpu_select_peak_kernel, dim3(n), dim3(64), 0, s, d_v.as<double>(), d_nc.as<int>(), ncand_max, step, d_sz.as<int>(), d_lc.as<int>(),
                           nTrials_max, n, d_d.as<int>(), d_c.as<double>());
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(delay, d_d.p, size_t(n) * 4, hipMemcpyDeviceToHost, s));
        HIPCK(hipMemcpyAsync(corr, d_c.p, size_t(n) * 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_debug_occupancy(mgpu_ctx* c, int which) {
    if (!c) return -1;
    int n = -1;
    guard(c, [&] {
        const auto& t = c->tab;
        if (which == 0 && t.mfsk_M == 0) HIPCK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fe_kernel(c->fe_threads), c->fe_threads, c->lds_fe));
        else if (which == 0) HIPCK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, t.mfsk_M == 32 ? mgpu_mfsk_frontend_kernel_m32 : mgpu_mfsk_frontend_kernel_m16x2, 256, 0));
        else if (which == 1) n = int(c->lds_fe);          // dynamic LDS bytes of the front-end workgroup
        else if (which == 2) n = int(c->lds_dec);         // ... of the decoder workgroup
        else if (which == 3) n = c->dec_threads;
        else n = -1;
    });
    return n;
}

int mgpu_debug_tsync_metric(mgpu_ctx* c, const double* bb, int W, int size, int step, int variant, const int* start, const int* sub_size, double* vals) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        const auto& t = c->tab;
        const int interp = 4, sym = t.Nofdm * interp, L = t.preamble * sym;
        need(bb && vals && W > 0 && size > L && step >= 1 && variant >= -1 && variant <= 2 && (!start == !sub_size), "bad argument");
        const int ncand = (size - L + step - 1) / step;
        std::vector<int> nc(W, ncand), st(W, 0), wi(W);
        for (int w = 0; w < W; ++w) {
            wi[w] = w;
            if (start) {
                need(start[w] >= 0 && sub_size[w] >= 0 && start[w] + sub_size[w] <= size, "sub-window outside the window");
                st[w] = start[w];
                nc[w] = sub_size[w] > L ? (sub_size[w] - L + step - 1) / step : 0;
            }
        }
        DevBuf d_in(size_t(W) * size * 16), d_vals(size_t(W) * ncand * 8), d_st(size_t(W) * 4), d_nc(size_t(W) * 4), d_wi(size_t(W) * 4);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_in.p, bb, size_t(W) * size * 16, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_st.p, st.data(), size_t(W) * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_nc.p, nc.data(), size_t(W) * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_wi.p, wi.data(), size_t(W) * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipMemsetAsync(d_vals.p, 0xff, size_t(W) * ncand * 8, s));
        HIPCK(hipEventRecord(c->sync_ev[0], s));
        launch_tsync_metric(d_in.as<double>(), size, start ? d_st.as<int>() : nullptr, start ? d_wi.as<int>() : nullptr, start ? d_nc.as<int>() : nullptr, ncand, W, step,
                            t.preamble, t.Ngi * interp, t.Nfft * interp, d_vals.as<double>(), s, variant);
        HIPCK(hipEventRecord(c->sync_ev[1], s));
        HIPCK(hipMemcpyAsync(vals, d_vals.p, size_t(W) * ncand * 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_debug_mfsk_sync(mgpu_ctx* c, const double* energy, int W, int nslots, int size, const int* search_start, int variant, int* delay) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        const auto& t = c->tab;
        need(t.mfsk_M > 0, "MFSK modes only (cfg 100..102)");
        need(energy && delay && W > 0 && nslots >= t.preamble && size > 0 && (variant == 0 || variant == 1), "bad argument");
        if (variant == 0) {
            for (int w = 0; w < W; ++w) delay[w] = mfsk_sync_from_energies(t, energy + size_t(w) * nslots * t.Nc, nslots, size, search_start ? search_start[w] : 0);
            return;
        }
        DevBuf d_e(size_t(W) * nslots * t.Nc * 8), d_ss(size_t(W) * 4), d_delay(size_t(W) * 4);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_e.p, energy, size_t(W) * nslots * t.Nc * 8, hipMemcpyHostToDevice, s));
        if (search_start) HIPCK(hipMemcpyAsync(d_ss.p, search_start, size_t(W) * 4, hipMemcpyHostToDevice, s));
        launch_mfsk_sync(c, d_e.as<double>(), W, nslots, size, search_start ? d_ss.as<int>() : nullptr, d_delay.as<int>(), s);
        HIPCK(hipMemcpyAsync(delay, d_delay.p, size_t(W) * 4, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_debug_span_energy(mgpu_ctx* c, const double* bb, int W, int size, const int* wv, const int* off, int n, int len, int variant, double* sum, int* cnt) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(bb && wv && off && sum && cnt && W > 0 && size > 0 && n > 0 && len > 0 && len <= 1088 && (variant == 0 || variant == 1), "bad argument");
        for (int j = 0; j < n; ++j) need(wv[j] >= 0 && wv[j] < W && off[j] >= 0, "span outside the windows");
        DevBuf d_in(size_t(W) * size * 16), d_wv(size_t(n) * 4), d_off(size_t(n) * 4), d_sum(size_t(n) * 8), d_cnt(size_t(n) * 4);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_in.p, bb, size_t(W) * size * 16, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_wv.p, wv, size_t(n) * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_off.p, off, size_t(n) * 4, hipMemcpyHostToDevice, s));
        HIPCK(hipEventRecord(c->sync_ev[0], s));
        if (variant)
            hipLaunchKernelGGL(mgpu_span_energy_many_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_in.as<double>(), size, d_wv.as<int>(), d_off.as<int>(), n, len,
                               d_sum.as<double>(), d_cnt.as<int>());
        else
            hipLaunchKernelGGL(mgpu_span_energy_kernel, dim3((n + 3) / 4), dim3(256), 0, s, d_in.as<double>(), size, d_wv.as<int>(), d_off.as<int>(), n, len,
                               d_sum.as<double>(), d_cnt.as<int>());
        HIPCK(hipGetLastError());
        HIPCK(hipEventRecord(c->sync_ev[1], s));
        HIPCK(hipMemcpyAsync(sum, d_sum.p, size_t(n) * 8, hipMemcpyDeviceToHost, s));
        HIPCK(hipMemcpyAsync(cnt, d_cnt.p, size_t(n) * 4, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

int mgpu_freq_sync(mgpu_ctx* c, const double* bb, int W, int stride, double* freq_offset_hz) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        const auto& t = c->tab;
        int pre_half = t.preamble / 2 == 0 ? 1 : t.preamble / 2;            // ofdm.cc:548-555
        need(bb && freq_offset_hz && W > 0 && stride >= pre_half * t.Nofdm, "bad argument");
        DevBuf d_in(size_t(W) * stride * 16), d_out(size_t(W) * 16);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_in.p, bb, size_t(W) * stride * 16, hipMemcpyHostToDevice, s));
        const double bandwidth = 48000.0 * 50.0 / 256 / 4;
        HIPCK(hipEventRecord(c->sync_ev[0], s));
        hipLaunchKernelGGL(mgpu_fsync_kernel, dim3(W), dim3(256), 0, s, d_in.as<double>(), stride, pre_half, c->dev.twiddle, d_out.as<double>());
        HIPCK(hipGetLastError());
        HIPCK(hipEventRecord(c->sync_ev[1], s));
        std::vector<double> mul(size_t(W) * 2);
        HIPCK(hipMemcpyAsync(mul.data(), d_out.p, size_t(W) * 16, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        for (int w = 0; w < W; ++w) freq_offset_hz[w] = moose_hz(mul[2 * w], mul[2 * w + 1], bandwidth / double(t.Nc));
    });
}

// ---- MFSK synchroniser / signalling blocks (host buffers, blocking) -----------------------------------
extern "C++" {
namespace {
constexpr int kInterp = 4;

// carrier energies of every symbol slot of W windows: [W][nslots][50] on the host
// `passband_carrier_hz` >= 0: `bb` is real passband audio ([W][size] doubles) that is first mixed down and filtered with
// FIR_rx_data on the device (detect_ack_pattern_from_passband, telecom_system.cc:1628-1640); otherwise it is interpolated
// baseband ([W][size] complex).
// d_e ([W][nslots][Nc] doubles on the device) receives the energies; want_host: also returned on the host
std::vector<double> slot_energies(mgpu_ctx* c, const double* bb, int W, int size, int nslots, double passband_carrier_hz, DevBuf& d_e, bool want_host) {
    const auto& t = c->tab;
    DevBuf d_in(size_t(W) * size * 16);
    hipStream_t s = c->stream;
    if (passband_carrier_hz >= 0) {
        DevBuf d_pass(size_t(W) * size * 8), d_fc(size_t(W) * 8);
        std::vector<double> fc(W, passband_carrier_hz);
        HIPCK(hipMemcpyAsync(d_pass.p, bb, size_t(W) * size * 8, hipMemcpyHostToDevice, s));
        HIPCK(hipMemcpyAsync(d_fc.p, fc.data(), size_t(W) * 8, hipMemcpyHostToDevice, s));
        const int ntaps = int(t.fir_data.size());
        const double* cs = mixer_table(c, passband_carrier_hz, size_t(size), s);
        launch_p2b(d_pass.as<double>(), size, d_fc.as<double>(), nullptr, 0, size, 1, c->d_fir[1], ntaps, d_in.as<double>(), nullptr, cs, nullptr, 0, W, s);
        HIPCK(hipStreamSynchronize(s));          // d_pass / d_fc go out of scope here
    } else {
        HIPCK(hipMemcpyAsync(d_in.p, bb, size_t(W) * size * 16, hipMemcpyHostToDevice, s));
    }
    HIPCK(hipMemsetAsync(d_e.p, 0, size_t(W) * nslots * t.Nc * 8, s));
    HIPCK(hipEventRecord(c->sync_ev[0], s));
    hipLaunchKernelGGL(mgpu_slot_energy_kernel, dim3((nslots + 3) / 4, W), dim3(256), 0, s, d_in.as<double>(), size, nslots, kInterp,
                       c->dev.twiddle, d_e.as<double>());
    HIPCK(hipGetLastError());
    HIPCK(hipEventRecord(c->sync_ev[1], s));
    std::vector<double> e;
    if (want_host) {
        e.resize(size_t(W) * nslots * t.Nc);
        HIPCK(hipMemcpyAsync(e.data(), d_e.p, e.size() * 8, hipMemcpyDeviceToHost, s));
    }
    HIPCK(hipStreamSynchronize(s));              // d_in goes out of scope here
    return e;
}
std::vector<double> slot_energies(mgpu_ctx* c, const double* bb, int W, int size, int nslots, double passband_carrier_hz = -1.0) {
    DevBuf d_e(size_t(W) * nslots * c->tab.Nc * 8);
    return slot_energies(c, bb, W, size, nslots, passband_carrier_hz, d_e, true);
}
}  // namespace
}  // extern "C++"

int mgpu_time_sync_mfsk(mgpu_ctx* c, const double* bb, int W, int size, int search_start_symb, int* delay) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        const auto& t = c->tab;
        need(t.mfsk_M > 0, "time_sync_mfsk needs an MFSK mode (cfg 100..102)");
        const int sym_period = t.Nofdm * kInterp, nslots = size / sym_period, np = t.preamble;
        need(bb && delay && W > 0 && nslots >= np, "bad argument");
        if (W == 1) {                                                // one window: 260 KB of energies, the search on the host
            const std::vector<double> E = slot_energies(c, bb, W, size, nslots);
            delay[0] = mfsk_sync_from_energies(t, E.data(), nslots, size, search_start_symb);
            return;
        }
        DevBuf d_e(size_t(W) * nslots * t.Nc * 8), d_ss(size_t(W) * 4), d_delay(size_t(W) * 4);
        slot_energies(c, bb, W, size, nslots, -1.0, d_e, false);
        const std::vector<int> ss(W, search_start_symb);
        HIPCK(hipMemcpyAsync(d_ss.p, ss.data(), size_t(W) * 4, hipMemcpyHostToDevice, c->stream));
        launch_mfsk_sync(c, d_e.as<double>(), W, nslots, size, d_ss.as<int>(), d_delay.as<int>(), c->stream);
        HIPCK(hipMemcpyAsync(delay, d_delay.p, size_t(W) * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
    });
}

static int detect_ack_impl(mgpu_ctx* c, const double* bb, int W, int size, int pattern, double passband_carrier_hz, double* metric_out,
                           int* matched_out);

int mgpu_detect_ack_pattern(mgpu_ctx* c, const double* bb, int W, int size, int pattern, double* metric_out, int* matched_out) {
    return detect_ack_impl(c, bb, W, size, pattern, -1.0, metric_out, matched_out);
}

int mgpu_detect_ack_pattern_from_passband(mgpu_ctx* c, const double* passband, int W, int size, double carrier_hz, int pattern,
                                          double* metric_out, int* matched_out) {
    if (!(carrier_hz >= 0)) return MGPU_ERR_ARG;
    return detect_ack_impl(c, passband, W, size, pattern, carrier_hz, metric_out, matched_out);
}

static int detect_ack_impl(mgpu_ctx* c, const double* bb, int W, int size, int pattern, double passband_carrier_hz, double* metric_out,
                           int* matched_out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        const auto& t = c->tab;
        const int sym_period = t.Nofdm * kInterp, nslots = size / sym_period;
        need(bb && metric_out && W > 0 && size > 0 && (pattern == 1 || pattern == 2) && t.Nc == 50, "bad argument");
        if (nslots < kAckNsymb) {                                      // ofdm.cc:2075
            for (int w = 0; w < W; ++w) { metric_out[w] = 0.0; if (matched_out) matched_out[w] = 0; }
            return;
        }
        const std::vector<double> E = slot_energies(c, bb, W, size, nslots, passband_carrier_hz);
        const int* tones = pattern == 2 ? kBreakTones : kAckTones;
        for (int w = 0; w < W; ++w) {                                  // ofdm.cc:2085-2178
            double best_metric = 0.0;
            int best_matched = 0;
            for (int s = 0; s <= nslots - kAckNsymb; ++s) {
                double metric = 0;
                int matched = 0;
                for (int p = 0; p < kAckNsymb; ++p) {
                    if ((s + p) * sym_period + t.Ngi * kInterp + t.Nfft * kInterp > size) break;
                    const double* e = &E[(size_t(w) * nslots + s + p) * t.Nc];
                    const int actual = (tones[p % kAckLen] + p * kAckHop) % kAckM;
                    const double e_expected = e[kAckOffset + actual];
                    double e_target = 0;
                    e_target += e_expected;
                    double peak_e = -1.0;
                    for (int q = 0; q < kAckM; ++q) if (e[kAckOffset + q] > peak_e) peak_e = e[kAckOffset + q];
                    if (!(e_expected >= peak_e)) continue;             // the expected tone must be the band's peak
                    ++matched;
                    double e_total = 0;
                    for (int k = 0; k < t.Nc; ++k) e_total += e[k];
                    if (e_total > 0) metric += e_target / e_total;
                }
                if (metric > best_metric) { best_metric = metric; best_matched = matched; }
            }
            metric_out[w] = best_metric;
            if (matched_out) matched_out[w] = best_matched;
        }
    });
}

int mgpu_last_sync_kernel_ms(mgpu_ctx* c, float* ms) {
    if (!c || !ms) return MGPU_ERR_ARG;
    return guard(c, [&] {
        HIPCK(hipEventSynchronize(c->sync_ev[1]));
        HIPCK(hipEventElapsedTime(ms, c->sync_ev[0], c->sync_ev[1]));
    });
}

int mgpu_debug_spa_math(mgpu_ctx* c, const double* in, int n, double* tanh_out, double* atanh_out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(in && tanh_out && atanh_out && n > 0, "bad argument");
        double *d_in = nullptr, *d_t = nullptr, *d_a = nullptr;
        HIPCK(hipMalloc(&d_in, size_t(n) * 8)); HIPCK(hipMalloc(&d_t, size_t(n) * 8)); HIPCK(hipMalloc(&d_a, size_t(n) * 8));
        HIPCK(hipMemcpy(d_in, in, size_t(n) * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(mgpu_spa_math_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_in, d_t, d_a, n);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(c->stream));
        HIPCK(hipMemcpy(tanh_out, d_t, size_t(n) * 8, hipMemcpyDeviceToHost));
        HIPCK(hipMemcpy(atanh_out, d_a, size_t(n) * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_in); (void)hipFree(d_t); (void)hipFree(d_a);
    });
}

int mgpu_debug_glibc_trig(mgpu_ctx* c, const double* in, int n, double* atan_out, double* sin_out, double* cos_out) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(in && atan_out && sin_out && cos_out && n > 0, "bad argument");
        DevBuf d_in(size_t(n) * 8), d_a(size_t(n) * 8), d_s(size_t(n) * 8), d_c(size_t(n) * 8);
        HIPCK(hipMemcpyAsync(d_in.p, in, size_t(n) * 8, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(mgpu_glibc_trig_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_in.as<double>(), d_a.as<double>(), d_s.as<double>(),
                           d_c.as<double>(), n);
        HIPCK(hipGetLastError());
        HIPCK(hipMemcpyAsync(atan_out, d_a.p, size_t(n) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipMemcpyAsync(sin_out, d_s.p, size_t(n) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipMemcpyAsync(cos_out, d_c.p, size_t(n) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCK(hipStreamSynchronize(c->stream));
    });
}

int mgpu_rx_batch_taps(mgpu_ctx* c, const double* bb, int F, uint8_t* payload, mgpu_frame_stats* stats, const mgpu_stage_taps* taps) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(bb && F >= 0 && F <= c->max_batch, "bad argument (F must be <= max_batch)");
        if (F == 0) return;
        const auto& t = c->tab;
        const size_t in_bytes = size_t(F) * t.frame_samples * 16;
        ensure_workspaces(c, WS_FRONTEND | WS_LLR | WS_OUT);
        if (c->baseband_cap < in_bytes) {
            (void)hipFree(c->d_baseband);
            c->d_baseband = nullptr; c->baseband_cap = 0;
            HIPCK(hipMalloc(&c->d_baseband, in_bytes));
            c->baseband_cap = in_bytes;
        }
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(c->d_baseband, bb, in_bytes, hipMemcpyHostToDevice, s));
        MgpuTapsDev dt{};
        std::vector<void*> tmp;
        const size_t G = size_t(t.Nsymb) * t.Nc;
        auto dalloc = [&](size_t bytes) { void* p = nullptr; HIPCK(hipMalloc(&p, bytes)); tmp.push_back(p); return p; };
        if (taps) {
            if (taps->grid) { dt.grid = static_cast<double*>(dalloc(F * G * 16)); if (t.mfsk_M > 0) HIPCK(hipMemsetAsync(dt.grid, 0, F * G * 16, s)); }
            need(t.mfsk_M == 0 || !(taps->H || taps->eq || taps->syms), "the MFSK modes have no channel estimate / equalised grid to tap");
            if (taps->H) dt.H = static_cast<double*>(dalloc(F * G * 16));
            if (taps->eq) dt.eq = static_cast<double*>(dalloc(F * G * 16));
            if (taps->syms) dt.syms = static_cast<double*>(dalloc(size_t(F) * t.nData * 16));
            if (taps->llr_demod) dt.llr_demod = static_cast<float*>(dalloc(size_t(F) * t.nBits * 4));
            if (taps->variance) dt.variance = static_cast<double*>(dalloc(size_t(F) * 8));
            if (taps->cycles) { dt.cycles = static_cast<long long*>(dalloc(16 * 8)); HIPCK(hipMemsetAsync(dt.cycles, 0, 16 * 8, s)); }
            if (taps->agc_gain) { dt.agc_gain = static_cast<double*>(dalloc(size_t(F) * 8)); HIPCK(hipMemsetAsync(dt.agc_gain, 0, size_t(F) * 8, s)); }
        }
        launch_frontend(c, c->d_baseband, F, c->d_llr, c->d_variance, c->d_snrvar, dt, s);
        launch_decoder(c, c->d_llr, F, nullptr, nullptr, c->d_payload, c->d_stats, c->d_variance, c->d_snrvar, s);
        launch_zf_snr(c, F, c->d_payload, c->d_stats, s);
        if (payload) HIPCK(hipMemcpyAsync(payload, c->d_payload, size_t(F) * t.payload_stride, hipMemcpyDeviceToHost, s));
        if (stats) HIPCK(hipMemcpyAsync(stats, c->d_stats, size_t(F) * sizeof(MgpuStatsDev), hipMemcpyDeviceToHost, s));
        if (taps) {
            auto back = [&](void* h, void* d, size_t bytes) { if (h) HIPCK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s)); };
            back(taps->grid, dt.grid, F * G * 16); back(taps->H, dt.H, F * G * 16); back(taps->eq, dt.eq, F * G * 16);
            back(taps->syms, dt.syms, size_t(F) * t.nData * 16); back(taps->llr_demod, dt.llr_demod, size_t(F) * t.nBits * 4);
            back(taps->variance, dt.variance, size_t(F) * 8); back(taps->agc_gain, dt.agc_gain, size_t(F) * 8);
            back(taps->llr_ldpc, c->d_llr, size_t(F) * t.N * 4);
            back(taps->cycles, dt.cycles, 16 * 8);
        }
        HIPCK(hipStreamSynchronize(s));
        for (void* p : tmp) (void)hipFree(p);
    });
}

// One frame per call is how the reference's receive_byte uses this span, and at that size the call is bound by launch and
// copy submission, not by the kernels: the whole sequence (H2D, front-end, decoder, [ZF SNR], D2H x2) is captured once into
// a hipGraph over fixed page-locked staging buffers and replayed with a single launch.
static int rx_one_frame(mgpu_ctx* c, const double* bb, uint8_t* payload, mgpu_frame_stats* stats) {
    return guard(c, [&] {
        const auto& t = c->tab;
        const size_t in_bytes = size_t(t.frame_samples) * 16, out_bytes = size_t(t.payload_stride) + sizeof(MgpuStatsDev);
        hipStream_t s = c->stream;
        if (!c->one_frame_graph) {
            ensure_workspaces(c, WS_FRONTEND | WS_LLR | WS_OUT);
            // The graph bakes in every address it touches, so it reads from a device buffer and staging buffers of its own
            // that live as long as the context (d_baseband may be reallocated by a larger batch later; the max_batch-sized
            // workspaces never are). Nothing is published in the context until the whole graph exists.
            if (!c->d_one_in) HIPCK(hipMalloc(&c->d_one_in, in_bytes));
            if (!c->h_one_in) HIPCK(host_alloc_on_node(&c->h_one_in, in_bytes, c->numa_node));
            if (!c->h_one_out) HIPCK(host_alloc_on_node(&c->h_one_out, out_bytes, c->numa_node));
            hipGraph_t graph = nullptr;
            HIPCK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            try {
                HIPCK(hipMemcpyAsync(c->d_one_in, c->h_one_in, in_bytes, hipMemcpyHostToDevice, s));
                MgpuTapsDev dt{};
                launch_frontend(c, c->d_one_in, 1, c->d_llr, c->d_variance, c->d_snrvar, dt, s);
                launch_decoder(c, c->d_llr, 1, nullptr, nullptr, c->d_payload, c->d_stats, c->d_variance, c->d_snrvar, s);
                launch_zf_snr(c, 1, c->d_payload, c->d_stats, s);
                HIPCK(hipMemcpyAsync(c->h_one_out, c->d_payload, t.payload_stride, hipMemcpyDeviceToHost, s));
                HIPCK(hipMemcpyAsync(static_cast<char*>(c->h_one_out) + t.payload_stride, c->d_stats, sizeof(MgpuStatsDev), hipMemcpyDeviceToHost, s));
            } catch (...) {
                (void)hipStreamEndCapture(s, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            HIPCK(hipStreamEndCapture(s, &graph));
            hipGraphExec_t exec = nullptr;
            const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            HIPCK(e);
            c->one_frame_graph = exec;
        }
        std::memcpy(c->h_one_in, bb, in_bytes);
        HIPCK(hipGraphLaunch(c->one_frame_graph, s));
        HIPCK(hipStreamSynchronize(s));
        if (payload) std::memcpy(payload, c->h_one_out, t.payload_stride);
        if (stats) std::memcpy(stats, static_cast<char*>(c->h_one_out) + t.payload_stride, sizeof(MgpuStatsDev));
    });
}

// The blocking host-buffer entry point for F > 1 with no stage taps: classic double buffering. The batch goes through in
// chunks; a copy stream brings chunk i+1 (104 KB per mode-8 frame over PCIe) into the second input buffer while the kernel
// stream runs the front-end and the decoder of chunk i; the kernels stay in order on one stream (two decoder launches sharing
// the CUs only delay each other), events hand the buffers back and forth, and the small payload / stats copies ride behind each
// chunk into page-locked staging. The LLR / variance / payload / stats workspaces are the max_batch-sized ones, addressed by
// frame offset. Results are byte-identical to the one-launch path (frames are independent).
static void rx_batch_pipelined(mgpu_ctx* c, const double* bb, int F, uint8_t* payload, mgpu_frame_stats* stats) {
    const auto& t = c->tab;
    ensure_workspaces(c, WS_FRONTEND | WS_LLR | WS_OUT);
    const size_t frame_bytes = size_t(t.frame_samples) * 16;
    // chunk size: a decoder launch keeps the whole chip busy only from 2 workgroups per CU upwards (512 codewords on 256 CUs; a
    // smaller launch takes just as long), and a copy should carry a few MB; so chunks are multiples of that wave of workgroups
    // and a batch that is not larger than one chunk goes through in one piece.
    if (c->wave_of_wgs == 0) {           // per context: the devices of a pool need not be alike
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->cfg.device);
        c->wave_of_wgs = 2 * (cus > 0 ? cus : 256);
    }
    const int wave_of_wgs = c->wave_of_wgs;
    int chunk = wave_of_wgs;
    while (size_t(chunk) * frame_bytes < (size_t(8) << 20)) chunk += wave_of_wgs;
    if (const char* e = std::getenv("MERCURY_RX_CHUNK")) chunk = std::max(1, std::atoi(e));
    chunk = std::min(chunk, F);
    for (auto& p : c->pipe) {
        if (!p.stream) HIPCK(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        if (!p.done) HIPCK(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
        if (!p.copied) HIPCK(hipEventCreateWithFlags(&p.copied, hipEventDisableTiming));
        if (p.cap < size_t(chunk) * frame_bytes) {
            HIPCK(hipStreamSynchronize(p.stream));
            (void)hipFree(p.d_in);
            p.d_in = nullptr; p.cap = 0;
            HIPCK(hipMalloc(&p.d_in, size_t(chunk) * frame_bytes));
            p.cap = size_t(chunk) * frame_bytes;
        }
    }
    // Results come back through page-locked staging owned by the context: a device-to-host copy into the caller's pageable
    // arrays would block the host until the chunk's kernels have finished, i.e. before the next chunk's input copy could even
    // be queued, and nothing would overlap.
    const size_t out_bytes = size_t(c->max_batch) * (t.payload_stride + sizeof(MgpuStatsDev));
    if (!c->h_out) HIPCK(host_alloc_on_node(&c->h_out, out_bytes + 16, c->numa_node));
    uint8_t* h_payload = static_cast<uint8_t*>(c->h_out);
    MgpuStatsDev* h_stats = reinterpret_cast<MgpuStatsDev*>(h_payload + ((size_t(c->max_batch) * t.payload_stride + 15) & ~size_t(15)));
    MgpuTapsDev none{};
    // Two ways to overlap, chosen by the kind of host memory (measured on MI355X / PCIe Gen5, tools/bench_host_path.py):
    //  * page-locked input (mgpu_alloc_host, hipHostMalloc/Register): the copy is a true asynchronous DMA. One copy stream runs
    //    ahead into the other input buffer while ONE kernel stream keeps the launches in order (two decoder launches sharing the
    //    CUs only delay each other); events hand the buffers back and forth.
    //  * pageable input: the runtime stages the copy itself and holds the calling thread until it is done, which already paces
    //    the copies one behind the other; each chunk's copy and kernels then go to the stream that owns the chunk's input
    //    buffer (two streams alternating), so a copy waits exactly for the front-end that last read its buffer.
    hipPointerAttribute_t attr{};
    const bool pinned = hipPointerGetAttributes(&attr, bb) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    for (auto& e : c->hp_ev) if (!e) HIPCK(hipEventCreate(&e));
    const int nchunks = (F + chunk - 1) / chunk;
    int k = 0;
    for (int off = 0; off < F; off += chunk, ++k) {
        auto& p = c->pipe[k % mgpu_ctx::kPipes];                     // input buffer of this chunk
        hipStream_t cs = pinned ? c->pipe[0].stream : p.stream, ks = pinned ? c->pipe[1].stream : p.stream;
        const int n = std::min(chunk, F - off);
        if (pinned && k >= mgpu_ctx::kPipes) HIPCK(hipStreamWaitEvent(cs, p.done, 0));          // the front-end of chunk k-2 has consumed it
        if (k == 0) HIPCK(hipEventRecord(c->hp_ev[0], cs));
        HIPCK(hipMemcpyAsync(p.d_in, reinterpret_cast<const char*>(bb) + size_t(off) * frame_bytes, size_t(n) * frame_bytes, hipMemcpyHostToDevice, cs));
        if (k == 0) HIPCK(hipEventRecord(c->hp_ev[1], cs));          // fill: nothing can compute before the first chunk has landed
        if (k == nchunks - 1) HIPCK(hipEventRecord(c->hp_ev[2], cs)); // drain: what is left when the last input byte has landed
        if (pinned) {
            HIPCK(hipEventRecord(p.copied, cs));
            HIPCK(hipStreamWaitEvent(ks, p.copied, 0));
        }
        launch_frontend(c, p.d_in, n, c->d_llr + size_t(off) * t.N, c->d_variance + off, c->d_snrvar + off, none, ks, 0, off);
        if (pinned) HIPCK(hipEventRecord(p.done, ks));               // the front-end is the only reader of the input buffer
        launch_decoder(c, c->d_llr + size_t(off) * t.N, n, nullptr, nullptr, c->d_payload + size_t(off) * t.payload_stride, c->d_stats + off,
                       c->d_variance + off, c->d_snrvar + off, ks);
        launch_zf_snr(c, n, c->d_payload + size_t(off) * t.payload_stride, c->d_stats + off, ks, off);
        if (payload) HIPCK(hipMemcpyAsync(h_payload + size_t(off) * t.payload_stride, c->d_payload + size_t(off) * t.payload_stride,
                                          size_t(n) * t.payload_stride, hipMemcpyDeviceToHost, ks));
        if (stats) HIPCK(hipMemcpyAsync(h_stats + off, c->d_stats + off, size_t(n) * sizeof(MgpuStatsDev), hipMemcpyDeviceToHost, ks));
        if (k == nchunks - 1) HIPCK(hipEventRecord(c->hp_ev[3], ks));
    }
    for (auto& p : c->pipe) HIPCK(hipStreamSynchronize(p.stream));
    c->hp_chunk = chunk; c->hp_nchunks = nchunks;
    (void)hipEventElapsedTime(&c->hp_fill_ms, c->hp_ev[0], c->hp_ev[1]);
    (void)hipEventElapsedTime(&c->hp_drain_ms, c->hp_ev[2], c->hp_ev[3]);
    (void)hipEventElapsedTime(&c->hp_total_ms, c->hp_ev[0], c->hp_ev[3]);
    if (payload) std::memcpy(payload, h_payload, size_t(F) * t.payload_stride);
    if (stats) std::memcpy(stats, h_stats, size_t(F) * sizeof(MgpuStatsDev));
}

int mgpu_host_path_last(mgpu_ctx* c, int* chunk_frames, int* n_chunks, float* fill_ms, float* drain_ms, float* total_ms) {
    if (!c) return MGPU_ERR_ARG;
    if (chunk_frames) *chunk_frames = c->hp_chunk;
    if (n_chunks) *n_chunks = c->hp_nchunks;
    if (fill_ms) *fill_ms = c->hp_fill_ms;
    if (drain_ms) *drain_ms = c->hp_drain_ms;
    if (total_ms) *total_ms = c->hp_total_ms;
    return MGPU_OK;
}

int mgpu_rx_batch(mgpu_ctx* c, const double* bb, int F, uint8_t* payload, mgpu_frame_stats* stats, float* llr_opt) {
    if (c && bb && F == 1 && !llr_opt && !c->timing && c->max_batch >= 1 && !std::getenv("MERCURY_NO_GRAPH"))
        return rx_one_frame(c, bb, payload, stats);
    if (c && bb && F > 1 && F <= c->max_batch && !llr_opt && !c->timing && !std::getenv("MERCURY_NO_PIPELINE"))
        return guard(c, [&] { rx_batch_pipelined(c, bb, F, payload, stats); });
    mgpu_stage_taps taps{};
    taps.llr_ldpc = llr_opt;
    return mgpu_rx_batch_taps(c, bb, F, payload, stats, llr_opt ? &taps : nullptr);
}

int mgpu_ldpc_batch(mgpu_ctx* c, const float* llr, int F, uint8_t* bits, int* iters) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(llr && F >= 0 && F <= c->max_batch, "bad argument (F must be <= max_batch)");
        if (F == 0) return;
        const auto& t = c->tab;
        hipStream_t s = c->stream;
        ensure_workspaces(c, WS_LLR | WS_BITS);
        HIPCK(hipMemcpyAsync(c->d_llr, llr, size_t(F) * t.N * 4, hipMemcpyHostToDevice, s));
        launch_decoder(c, c->d_llr, F, c->d_bits, c->d_iters, nullptr, nullptr, nullptr, nullptr, s);
        if (bits) HIPCK(hipMemcpyAsync(bits, c->d_bits, size_t(F) * t.K, hipMemcpyDeviceToHost, s));
        if (iters) HIPCK(hipMemcpyAsync(iters, c->d_iters, size_t(F) * 4, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

// ---- host-side pieces of the library, callable without a GPU (the CPU test suite checks them against the oracle) ----------------
int mgpu_host_select_peak(const double* cand_vals, int ncand, int step, int size, int location_to_return, int nTrials_max, int* delay,
                          double* correlation) {
    if (!cand_vals || !delay || ncand < 0 || step < 1 || size < 1 || nTrials_max < 1 || nTrials_max > size) return MGPU_ERR_ARG;
    double corr = 0;
    select_peak(cand_vals, ncand, step, size, location_to_return, nTrials_max, delay, &corr);
    if (correlation) *correlation = corr;
    return MGPU_OK;
}

int mgpu_host_fir_taps(int which, double carrier_hz, double* taps, int* ntaps) {
    if (!taps || !ntaps || which < 0 || which > 3) return MGPU_ERR_ARG;
    try {
        std::vector<double> t;
        if (which >= 2) t = mgpu::design_tx_fir(which - 2, carrier_hz);
        else {
            const mgpu::ModeTables m = mgpu::build_mode_tables(8, 0, mgpu_ldpc_blob, mgpu_ldpc_blob_size);
            t = which ? m.fir_data : m.fir_time_sync;
        }
        std::copy(t.begin(), t.end(), taps);
        *ntaps = int(t.size());
        return MGPU_OK;
    } catch (const std::exception&) { return MGPU_ERR_ARG; }
}

int mgpu_host_preamble_carriers(int cfg, double* carriers_c128, int* n_symbols) {
    if (!carriers_c128 || !n_symbols) return MGPU_ERR_ARG;
    try {
        const mgpu::ModeTables m = mgpu::build_mode_tables(cfg, 0, mgpu_ldpc_blob, mgpu_ldpc_blob_size);
        std::memcpy(carriers_c128, m.preamble_carriers.data(), m.preamble_carriers.size() * 16);
        *n_symbols = m.preamble;
        return MGPU_OK;
    } catch (const std::exception&) { return MGPU_ERR_ARG; }
}

int mgpu_ldpc_encode_batch(mgpu_ctx* c, const uint8_t* bits, int F, uint8_t* encoded) {
    if (!c) return MGPU_ERR_ARG;
    return guard(c, [&] {
        need(bits && encoded && F >= 0, "bad argument");
        if (F == 0) return;
        const auto& t = c->tab;
        DevBuf d_in(size_t(F) * t.K), d_out(size_t(F) * t.N);
        hipStream_t s = c->stream;
        HIPCK(hipMemcpyAsync(d_in.p, bits, size_t(F) * t.K, hipMemcpyHostToDevice, s));
        for (int off = 0; off < F; off += kMaxFramesPerLaunch) {
            const int n = std::min(F - off, kMaxFramesPerLaunch);
            hipLaunchKernelGGL(mgpu_ldpc_encode_kernel, dim3(n), dim3(256), 0, s, c->dev, d_in.as<uint8_t>() + size_t(off) * t.K, n,
                               d_out.as<uint8_t>() + size_t(off) * t.N);
            HIPCK(hipGetLastError());
        }
        HIPCK(hipMemcpyAsync(encoded, d_out.p, size_t(F) * t.N, hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
    });
}

}  // extern "C"
