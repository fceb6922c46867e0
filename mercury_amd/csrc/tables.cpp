// Host-side table construction for the Mercury RX kernels — see tables.hpp for the reference map.
#include "tables.hpp"
#include "device_tables.h"      // kSpaMaxVarDegree: the degree bound the fp64 decoder is built (and its hard-frame shortcut proved) with

#include <algorithm>
#include <complex>
#include <cmath>
#include <cstring>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>

namespace mgpu {

// ---- glibc random() TYPE_3 (os_interop.cc:192-283) -------------------------------------------
void GlibcRandom::reseed(unsigned seed) {
    if (seed == 0) seed = 1;
    st_[0] = static_cast<int32_t>(seed);
    int32_t word = st_[0];
    for (int i = 1; i < 31; ++i) {
        // state[i] = (16807 * state[i-1]) % 2147483647 without overflowing 31 bits
        const long hi = word / 127773, lo = word % 127773;
        word = static_cast<int32_t>(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        st_[i] = word;
    }
    f_ = 3;
    r_ = 0;
    for (int k = 0; k < 310; ++k) (void)next();
}

int32_t GlibcRandom::next() {
    const uint32_t val = static_cast<uint32_t>(st_[f_]) + static_cast<uint32_t>(st_[r_]);
    st_[f_] = static_cast<int32_t>(val);
    if (++f_ >= 31) { f_ = 0; ++r_; }
    else if (++r_ >= 31) r_ = 0;
    return static_cast<int32_t>(val >> 1);
}

uint16_t crc16_modbus(const uint8_t* bytes, int n) {
    uint16_t crc = 0xffff;
    for (int j = 0; j < n; ++j) {
        crc ^= bytes[j];
        for (int i = 0; i < 8; ++i) crc = (crc & 1) ? static_cast<uint16_t>((crc >> 1) ^ 0xA001) : static_cast<uint16_t>(crc >> 1);
    }
    return crc;
}

namespace {

struct ModeRow { int M, rate16, preamble, estimator; };
// telecom_system.cc:2506-2624
constexpr ModeRow kModeTable[17] = {
    {2, 1, 4, 1},  {2, 2, 4, 1},  {2, 3, 4, 1},  {2, 4, 4, 1},  {2, 5, 4, 1},  {2, 6, 4, 1},
    {2, 8, 4, 1},  {4, 5, 4, 1},  {4, 6, 4, 1},  {4, 8, 4, 1},  {8, 6, 3, 1},  {8, 8, 3, 1},
    {4, 14, 3, 1}, {16, 8, 2, 1}, {8, 14, 2, 1}, {16, 14, 2, 0}, {32, 14, 1, 0},
};

// psk.cc:124-157 — the 32-point cross constellation has no closed form
constexpr int8_t kQam32[32][2] = {
    {-3, 5}, {-1, 5}, {-3, -5}, {-1, -5}, {-5, 3}, {-5, 1}, {-5, -3}, {-5, -1},
    {-1, 3}, {-1, 1}, {-1, -3}, {-1, -1}, {-3, 3}, {-3, 1}, {-3, -3}, {-3, -1},
    {3, 5},  {1, 5},  {3, -5},  {1, -5},  {5, 3},  {5, 1},  {5, -3},  {5, -1},
    {1, 3},  {1, 1},  {1, -3},  {1, -1},  {3, 3},  {3, 1},  {3, -3},  {3, -1}};

std::vector<Cplx> make_constellation(int M) {
    std::vector<Cplx> c(M);
    switch (M) {
        case 2: c = {{1, 0}, {-1, 0}}; break;
        case 4: c = {{-1, 1}, {-1, -1}, {1, 1}, {1, -1}}; break;
        case 8: {
            const double h = std::sqrt(2.0);  // complex * sqrt(2.0) / 2.0, component-wise (psk.cc:83-90)
            const double p = (1 * h) / 2.0, n = (-1 * h) / 2.0;
            c = {{n, n}, {-1, 0}, {0, 1}, {n, p}, {0, -1}, {p, n}, {p, p}, {1, 0}};
            break;
        }
        case 16:  // psk.cc:105-121: index b3b2b1b0 -> I = (b3 ? + : -)(b2 ? 1 : 3), Q = (b1 ? - : +)(b0 ? 1 : 3)
            for (int s = 0; s < 16; ++s)
                c[s] = {((s & 8) ? 1.0 : -1.0) * ((s & 4) ? 1.0 : 3.0), ((s & 2) ? -1.0 : 1.0) * ((s & 1) ? 1.0 : 3.0)};
            break;
        case 32:
            for (int s = 0; s < 32; ++s) c[s] = {double(kQam32[s][0]), double(kQam32[s][1])};
            break;
        default: throw std::runtime_error("unsupported constellation size");
    }
    // psk.cc:229-256: the normaliser is accumulated and stored in a FLOAT
    float pnv = 0;
    for (const Cplx& z : c) pnv = static_cast<float>(static_cast<double>(pnv) + (z.re * z.re + z.im * z.im));
    const float mean = pnv / static_cast<float>(M);
    pnv = static_cast<float>(1.0 / std::sqrt(static_cast<double>(mean)));
    for (Cplx& z : c) { z.re *= static_cast<double>(pnv); z.im *= static_cast<double>(pnv); }
    return c;
}

// cl_pilot_configurator::configure (ofdm.cc:976-1064) for Dx=1, all edge rows/cols DATA,
// last_col AUTO_SELLECT with the COPY_FIRST_COL fallback.
std::vector<uint8_t> make_pilot_lattice(int Nsymb, int Nc, int Dy) {
    const int Dx = 1;
    const int S = Nc > Nsymb ? Nc : Nsymb;
    std::vector<uint8_t> v(size_t(S) * S, 0);
    for (int x = 0, y = 0; x < S && y < S; x += Dx, ++y) {
        for (int j = y; j < S; j += Dy) v[size_t(j) * S + x] = 1;
        for (int j = y; j >= 0; j -= Dy) v[size_t(j) * S + x] = 1;
    }
    int in_last = 0;
    for (int j = 0; j < Nsymb; ++j) in_last += v[size_t(j) * S + Nc - 1];
    if (in_last < 2)
        for (int j = 0; j < S; ++j) v[size_t(j) * S + Nc - 1] = v[size_t(j) * S];
    std::vector<uint8_t> t(size_t(Nsymb) * Nc);
    for (int j = 0; j < Nsymb; ++j)
        for (int i = 0; i < Nc; ++i) t[size_t(j) * Nc + i] = v[size_t(j) * S + i];
    return t;
}

static LdpcGraph load_graph_uncached(int K, const uint8_t* blob, size_t size) {
    auto rd32 = [&](size_t off) { uint32_t x; if (off + 4 > size) throw std::runtime_error("LDPC table blob truncated"); std::memcpy(&x, blob + off, 4); return x; };
    if (size < 12 || std::memcmp(blob, "MLDP", 4) != 0 || rd32(4) != 1) throw std::runtime_error("LDPC table blob: bad magic/version");
    const uint32_t nrates = rd32(8);
    size_t off = 12;
    for (uint32_t r = 0; r < nrates; ++r) {
        const uint32_t k = rd32(off), P = rd32(off + 4), N = rd32(off + 8), E = rd32(off + 12), cw = rd32(off + 16), vw = rd32(off + 20);
        off += 24;
        const size_t need = size_t(P) + 2 * size_t(E) + N + 2 * size_t(E);
        if (off + need > size) throw std::runtime_error("LDPC table blob truncated");
        if (int(k) != K) { off += need; continue; }
        LdpcGraph g;
        if (N != 1600) throw std::runtime_error("LDPC table blob: N must be 1600 (the decoders' LDS layout is fixed to it)");
        g.K = K; g.P = P; g.N = N; g.E = E; g.Cwidth = cw; g.Vwidth = vw;
        const uint8_t* cdeg = blob + off;
        const uint8_t* cflat = cdeg + P;
        const uint8_t* vdeg = cflat + 2 * size_t(E);
        const uint8_t* vflat = vdeg + N;
        g.cptr.resize(P + 1); g.cvar.resize(E); g.vptr.resize(N + 1); g.vedge.resize(E);
        uint32_t e = 0;
        for (uint32_t c = 0; c < P; ++c) {
            g.cptr[c] = e;
            for (uint32_t j = 0; j < cdeg[c]; ++j, ++e) {
                uint16_t v; std::memcpy(&v, cflat + 2 * size_t(e), 2);
                g.cvar[e] = v;
            }
        }
        g.cptr[P] = e;
        uint32_t s = 0;
        for (uint32_t v = 0; v < N; ++v) {
            g.vptr[v] = s;
            for (uint32_t j = 0; j < vdeg[v]; ++j, ++s) {
                uint16_t c; std::memcpy(&c, vflat + 2 * size_t(s), 2);
                // slot j of variable v talks to check c: find v's edge inside check c (V_pos, ldpc_decoder_SPA.cc:81-104)
                uint32_t found = UINT32_MAX;
                for (uint32_t q = g.cptr[c]; q < g.cptr[c] + cdeg[c]; ++q)
                    if (g.cvar[q] == v) { found = q; break; }
                if (found == UINT32_MAX) throw std::runtime_error("LDPC table blob: C/V adjacency mismatch");
                g.vedge[s] = uint16_t(found);
            }
        }
        g.vptr[N] = s;
        if (s != E) throw std::runtime_error("LDPC table blob: edge count mismatch");
        // ---- wave-private bins (first-fit decreasing over check degrees) ----------------------
        std::vector<uint32_t> order(P);
        for (uint32_t c = 0; c < P; ++c) order[c] = c;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cdeg[a] > cdeg[b]; });
        std::vector<int> fill;                       // used slots per bin
        std::vector<std::vector<uint32_t>> members;  // checks per bin
        for (uint32_t c : order) {
            const int d = cdeg[c];
            if (d > 64) throw std::runtime_error("check degree exceeds a wavefront");
            size_t b = 0;
            for (; b < fill.size(); ++b) if (fill[b] + d <= 64) break;
            if (b == fill.size()) { fill.push_back(0); members.emplace_back(); }
            fill[b] += d;
            members[b].push_back(c);
        }
        g.S = int(fill.size()) * 64;
        if (g.S >= 8192) throw std::runtime_error("padded edge count exceeds the 13-bit slot field");
        std::vector<uint32_t> slot_of_edge(E);
        for (size_t b = 0; b < members.size(); ++b) {
            uint32_t p = uint32_t(b) * 64;
            for (uint32_t c : members[b]) {
                const uint32_t d = cdeg[c];
                for (uint32_t j = 0; j < d; ++j, ++p) slot_of_edge[g.cptr[c] + j] = p;
            }
        }
        // variable update order: by degree (descending) so the lanes of a wavefront share a trip count
        std::vector<uint32_t> vorder(N);
        for (uint32_t v = 0; v < N; ++v) vorder[v] = v;
        std::stable_sort(vorder.begin(), vorder.end(), [&](uint32_t a, uint32_t b) { return vdeg[a] > vdeg[b]; });
        g.vinfo.assign(size_t(N) * 8, 0u);   // per variable: v | deg<<11, then 10 u16 slot indices in 5 words; rows of 8 words (16-byte aligned)
        for (uint32_t i = 0; i < N; ++i) {
            const uint32_t v = vorder[i], d = vdeg[v];
            if (d > uint32_t(kSpaMaxVarDegree)) throw std::runtime_error("variable degree exceeds the unrolled update (and the bound the fp64 decoder's hard-frame shortcut is proved with: device_tables.h kSpaMaxVarDegree)");
            if (i >= 1024 && d > 4) g.fp64_limit = "variable degrees exceed the fp64 decoder's register layout (rows from 1024 on: at most 4 edges)";
            g.vinfo[size_t(i) * 8] = v | (d << 11);
            for (uint32_t j = 0; j < d; ++j) {
                const uint32_t slot = slot_of_edge[g.vedge[g.vptr[v] + j]];
                g.vinfo[size_t(i) * 8 + 1 + j / 2] |= slot << (16 * (j & 1));
            }
        }
        // ---- grouped layout for the fp32 sum-product kernel ---------------------------------------
        {
            std::vector<std::vector<uint32_t>> by_kind(7);           // checks per group-size exponent
            for (uint32_t c : order) {
                int k = 1;
                while ((1 << k) < int(cdeg[c])) ++k;
                by_kind[k].push_back(c);
            }
            std::vector<uint32_t> kinds;                              // per bin
            std::vector<std::vector<uint32_t>> bin_checks;
            for (int k = 6; k >= 1; --k) {
                const size_t per = size_t(64) >> k;
                for (size_t i = 0; i < by_kind[k].size(); i += per) {
                    kinds.push_back(uint32_t(k));
                    bin_checks.emplace_back(by_kind[k].begin() + i, by_kind[k].begin() + std::min(by_kind[k].size(), i + per));
                }
            }
            const size_t bins = kinds.size(), rounds = (bins + 7) / 8;
            g.Sg = int(bins) * 64;
            if (g.Sg >= 65536) throw std::runtime_error("grouped layout exceeds the 16-bit slot index of the variable records");
            if (rounds > 21) throw std::runtime_error("grouped layout exceeds the 21 rounds whose group sizes fit one 64-bit word per wavefront");
            g.gdesc.assign((rounds + 1) * 512, 0u);
            g.gkind.assign((rounds + 1) * 8, 0u);
            g.gkpack.assign(8, 0ull);
            // ---- bank-aware placement (round 4) --------------------------------------------------------------------------------------
            // The two scattered LDS accesses of these kernels are 4-byte gathers: the check pass reads the posterior of every slot's variable,
            // the variable update reads the messages of every row's edges. A wave64 ds_read_b32 is served as two groups of 32 lanes, one cycle
            // each when the 32 addresses fall into 32 different banks ((address / 4) mod 32) and one more per extra address on a bank; with
            // variables and slots where the graph happens to put them that is ~3.5 cycles per group (PMC round 3: two conflict cycles per LDS
            // instruction, a fifth of the kernels' time). Two things are free to choose, and neither changes any arithmetic:
            //  (A) where a variable's posterior lives. It is stored at its ROW (the order the variable update walks, sorted by degree) instead
            //      of at its number, which makes the update's own stores consecutive, and inside every run of 32 rows the variables are
            //      permuted so that the 32 variables a half-wavefront of the check pass gathers lie in different banks;
            //  (B) which lane of its check's group an edge sits on (the product over a group is an all-reduce: any order). The edges are
            //      permuted inside their group - never across the two halves of a bin - so that the j-th messages of 32 consecutive rows lie in
            //      different banks.
            // Both are descents on sum(count^2) over (access, bank) by pairwise swaps: deterministic, a few milliseconds per graph.
            std::vector<uint32_t> lane_of_edge(E), bin_of_edge(E), group_of_edge(E), gsize_of_edge(E);
            for (size_t b = 0; b < bins; ++b) {
                g.gkind[b] = kinds[b];
                g.gkpack[b & 7] |= uint64_t(kinds[b]) << (3 * (b >> 3));
                const uint32_t gsz = 1u << kinds[b];
                for (size_t q = 0; q < bin_checks[b].size(); ++q) {
                    const uint32_t c = bin_checks[b][q];
                    for (uint32_t j = 0; j < cdeg[c]; ++j) {
                        const uint32_t eo = g.cptr[c] + j;
                        lane_of_edge[eo] = uint32_t(q) * gsz + j;
                        bin_of_edge[eo] = uint32_t(b);
                        group_of_edge[eo] = c;
                        gsize_of_edge[eo] = gsz;
                    }
                }
            }
            // (A) rows: the degree order, permuted inside runs of 32
            std::vector<uint32_t> vorder_g = vorder, row_of(N);
            {
                // distinct variables per half-bin (two checks of a bin may share a variable: one address, a broadcast, no conflict)
                std::vector<std::vector<uint32_t>> halves_of_var(N);
                for (uint32_t c = 0; c < P; ++c)
                    for (uint32_t j = 0; j < cdeg[c]; ++j) {
                        const uint32_t eo = g.cptr[c] + j, h = bin_of_edge[eo] * 2 + (lane_of_edge[eo] >= 32 ? 1u : 0u);
                        auto& hv = halves_of_var[g.cvar[eo]];
                        if (std::find(hv.begin(), hv.end(), h) == hv.end()) hv.push_back(h);
                    }
                std::vector<int> cnt(bins * 2 * 32, 0);
                auto bump = [&](uint32_t v, uint32_t bank, int d) { long long delta = 0; for (uint32_t h : halves_of_var[v]) { int& x = cnt[size_t(h) * 32 + bank]; delta += d > 0 ? 2 * x + 1 : 1 - 2 * x; x += d; } return delta; };
                for (uint32_t i = 0; i < N; ++i) bump(vorder_g[i], i & 31, +1);
                uint32_t lcg = 12345u;                                   // sideways moves (no change of the objective) are taken on a fixed coin
                auto coin = [&] { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 16) & 1u; };
                // two rows may trade places when that keeps the update's register layout and its wavefront-uniform early exits: inside a run of 32
                // rows always, across runs when the degrees are equal and both rows belong to the same block of 512
                auto try_swap = [&](uint32_t ia, uint32_t ib) {
                    const uint32_t va = vorder_g[ia], vb = vorder_g[ib];
                    if ((ia & 31) == (ib & 31)) return false;
                    const long long d = bump(va, ia & 31, -1) + bump(vb, ib & 31, -1) + bump(va, ib & 31, +1) + bump(vb, ia & 31, +1);
                    if (d < 0 || (d == 0 && coin())) { std::swap(vorder_g[ia], vorder_g[ib]); return d < 0; }
                    bump(va, ib & 31, -1); bump(vb, ia & 31, -1); bump(va, ia & 31, +1); bump(vb, ib & 31, +1);
                    return false;
                };
                for (int sweep = 0, idle = 0; sweep < 200 && idle < 6; ++sweep) {
                    bool improved = false;
                    for (uint32_t ia = 0; ia < N; ++ia) {
                        const uint32_t lo = ia & ~31u, hi = std::min<uint32_t>(lo + 32, N);
                        for (uint32_t ib = ia + 1; ib < hi; ++ib) improved |= try_swap(ia, ib);
                        // same degree, same block of 512, another run
                        for (uint32_t ib = hi; ib < N && ib / 512 == ia / 512 && vdeg[vorder_g[ib]] == vdeg[vorder_g[ia]]; ++ib) improved |= try_swap(ia, ib);
                    }
                    idle = improved ? 0 : idle + 1;
                }
                for (uint32_t i = 0; i < N; ++i) row_of[vorder_g[i]] = i;
            }
            // (B) lanes: which access an edge belongs to = (run of 32 rows of its variable, its index in the variable's slot order)
            {
                std::vector<uint32_t> access_of_edge(E);
                for (uint32_t v = 0; v < N; ++v)
                    for (uint32_t j = 0; j < vdeg[v]; ++j) access_of_edge[g.vedge[g.vptr[v] + j]] = (row_of[v] / 32) * 9 + j;
                std::vector<int> cnt(size_t((N + 31) / 32) * 9 * 32, 0);
                auto at = [&](uint32_t e, uint32_t lane) -> int& { return cnt[size_t(access_of_edge[e]) * 32 + (lane & 31)]; };
                for (uint32_t e2 = 0; e2 < E; ++e2) ++at(e2, lane_of_edge[e2]);
                uint32_t lcg = 54321u;
                auto coin = [&] { lcg = lcg * 1664525u + 1013904223u; return (lcg >> 16) & 1u; };
                for (int sweep = 0, idle = 0; sweep < 200 && idle < 6; ++sweep) {
                    bool improved = false;
                    for (uint32_t c = 0; c < P; ++c) {
                        const uint32_t e0 = g.cptr[c], d = cdeg[c];
                        if (d == 0) continue;
                        const uint32_t gsz = gsize_of_edge[e0], gbase = lane_of_edge[e0] - (lane_of_edge[e0] % gsz);
                        // the lanes of the group, used or free; an edge may move to any lane of its group inside its half of the bin
                        std::vector<int> edge_at(gsz, -1);
                        for (uint32_t j = 0; j < d; ++j) edge_at[lane_of_edge[e0 + j] - gbase] = int(e0 + j);
                        for (uint32_t a = 0; a < gsz; ++a)
                            for (uint32_t b2 = a + 1; b2 < gsz; ++b2) {
                                if (((gbase + a) ^ (gbase + b2)) & 32) continue;
                                const int ea = edge_at[a], eb = edge_at[b2];
                                if (ea < 0 && eb < 0) continue;
                                long long dlt = 0;
                                auto mv = [&](int e2, uint32_t from, uint32_t to) { if (e2 < 0) return; int& x = at(uint32_t(e2), from); dlt += 1 - 2 * x; --x; int& y = at(uint32_t(e2), to); dlt += 2 * y + 1; ++y; };
                                mv(ea, gbase + a, gbase + b2); mv(eb, gbase + b2, gbase + a);
                                if (dlt < 0 || (dlt == 0 && coin())) {
                                    if (ea >= 0) lane_of_edge[ea] = gbase + b2;
                                    if (eb >= 0) lane_of_edge[eb] = gbase + a;
                                    std::swap(edge_at[a], edge_at[b2]);
                                    improved |= dlt < 0;
                                } else { mv(ea, gbase + b2, gbase + a); mv(eb, gbase + a, gbase + b2); }
                            }
                    }
                    // ... and two groups of a bin may trade places inside their half (a group of 8 lanes reaches only 8 of the 32 banks: when more
                    // than 8 of an access's edges sit in groups at the same offset, no order inside the groups separates them)
                    for (size_t b = 0; b < bins; ++b) {
                        const uint32_t gsz = 1u << kinds[b];
                        if (gsz >= 32) continue;
                        const uint32_t per_half = 32 / gsz;
                        std::vector<std::vector<uint32_t>> members(64 / gsz);          // edges per group position
                        for (uint32_t c : bin_checks[b])
                            for (uint32_t j = 0; j < cdeg[c]; ++j) members[lane_of_edge[g.cptr[c] + j] / gsz].push_back(g.cptr[c] + j);
                        for (uint32_t half = 0; half < 2; ++half)
                            for (uint32_t pa = 0; pa < per_half; ++pa)
                                for (uint32_t pb = pa + 1; pb < per_half; ++pb) {
                                    const uint32_t ga = half * per_half + pa, gb = half * per_half + pb;
                                    if (members[ga].empty() && members[gb].empty()) continue;
                                    const int shift = int(gb - ga) * int(gsz);
                                    long long dlt = 0;
                                    auto mvg = [&](const std::vector<uint32_t>& es, int sh) {
                                        for (uint32_t e2 : es) { int& x = at(e2, lane_of_edge[e2]); dlt += 1 - 2 * x; --x; }
                                        for (uint32_t e2 : es) { int& y = at(e2, uint32_t(int(lane_of_edge[e2]) + sh)); dlt += 2 * y + 1; ++y; }
                                    };
                                    mvg(members[ga], shift); mvg(members[gb], -shift);
                                    if (dlt < 0 || (dlt == 0 && coin())) {
                                        for (uint32_t e2 : members[ga]) lane_of_edge[e2] = uint32_t(int(lane_of_edge[e2]) + shift);
                                        for (uint32_t e2 : members[gb]) lane_of_edge[e2] = uint32_t(int(lane_of_edge[e2]) - shift);
                                        std::swap(members[ga], members[gb]);
                                        improved |= dlt < 0;
                                    } else {
                                        // undo the counts (the lanes were not changed)
                                        for (uint32_t e2 : members[ga]) { --at(e2, uint32_t(int(lane_of_edge[e2]) + shift)); ++at(e2, lane_of_edge[e2]); }
                                        for (uint32_t e2 : members[gb]) { --at(e2, uint32_t(int(lane_of_edge[e2]) - shift)); ++at(e2, lane_of_edge[e2]); }
                                    }
                                }
                    }
                    idle = improved ? 0 : idle + 1;
                }
            }
            std::vector<uint32_t> gslot_of_edge(E);
            for (uint32_t eo = 0; eo < E; ++eo) {
                gslot_of_edge[eo] = bin_of_edge[eo] * 64 + lane_of_edge[eo];
                g.gdesc[gslot_of_edge[eo]] = 0x80000000u | row_of[g.cvar[eo]];          // the posterior's index = the variable's row
            }
            {   // what the placement achieved, by the rule of the LDS (a 32-lane group costs one cycle + one per extra address on a bank)
                auto group_cycles = [](std::vector<uint32_t>& addr) {
                    std::sort(addr.begin(), addr.end());
                    addr.erase(std::unique(addr.begin(), addr.end()), addr.end());
                    int cnt[32] = {0}, m = 1;
                    for (uint32_t a : addr) m = std::max(m, ++cnt[a & 31]);
                    return m;
                };
                long cyc = 0, groups = 0;
                for (size_t b = 0; b < bins; ++b)
                    for (uint32_t half = 0; half < 2; ++half) {
                        std::vector<uint32_t> addr;
                        for (uint32_t l = 0; l < 32; ++l) { const uint32_t d = g.gdesc[b * 64 + half * 32 + l]; addr.push_back(d & 0x80000000u ? d & 0x7ffu : 0u); }
                        cyc += group_cycles(addr); ++groups;
                    }
                g.bank_model[0] = groups ? double(cyc) / groups : 0;
                cyc = groups = 0;
                for (uint32_t base = 0; base < N; base += 32)
                    for (uint32_t j = 0; j < 9; ++j) {
                        std::vector<uint32_t> addr;
                        for (uint32_t i = base; i < std::min<uint32_t>(base + 32, N); ++i)
                            if (j < vdeg[vorder_g[i]]) addr.push_back(gslot_of_edge[g.vedge[g.vptr[vorder_g[i]] + j]]);
                        if (!addr.empty()) { cyc += group_cycles(addr); ++groups; }
                    }
                g.bank_model[1] = groups ? double(cyc) / groups : 0;
            }
            g.vinfo_g.assign(size_t(N) * 8, 0u);
            for (uint32_t i = 0; i < N; ++i) {
                const uint32_t v = vorder_g[i], d = vdeg[v];
                // the fp32 decoders keep a lane's records (rows i, i + 512, ...) in 6 + 4 + 3 + 2 registers
                if (d > (i < 512 ? 9u : i < 1024 ? 6u : i < 1536 ? 4u : 2u)) g.fp32_limit = "variable degrees exceed the fp32 decoders' register layout (rows of 512: 9 / 6 / 4 / 2 edges)";
                g.vinfo_g[size_t(i) * 8] = v | (d << 11);
                for (uint32_t j = 0; j < d; ++j) {
                    const uint32_t slot = gslot_of_edge[g.vedge[g.vptr[v] + j]];
                    g.vinfo_g[size_t(i) * 8 + 1 + j / 2] |= slot << (16 * (j & 1));
                }
            }
        }
        // ---- the fp64 sum-product kernel's view: byte offsets and tabulated walk masks ----------
        {
            int maxdeg = 0;
            for (uint32_t c = 0; c < P; ++c) maxdeg = std::max(maxdeg, int(cdeg[c]));
            g.maxdeg = maxdeg;
            g.DM = ((maxdeg + 7) & ~7) + 8;      // the walk runs in groups of (two, four or) eight steps; an all-zero group ends it
            const size_t rounds = size_t(std::max(4, (g.S + 1023) / 1024));    // the smallest kernel instance runs 4 rounds
            g.sadr.assign((rounds + 1) * 1024 * 2, 0u);
            g.bhead.assign((rounds + 1) * 16 * 4, 0ull);
            g.bmask.assign(rounds * 16 * size_t(g.DM), 0ull);
            for (size_t b = 0; b < members.size(); ++b) {
                uint32_t p = uint32_t(b) * 64;
                for (uint32_t c : members[b]) {
                    const uint32_t cs = p, d = cdeg[c];
                    for (uint32_t j = 0; j < d; ++j, ++p) {
                        const uint32_t eo = g.cptr[c] + j;
                        g.sadr[size_t(p) * 2] = kSpaOnesBytes + g.cvar[eo] * 8u;           // LDS address of the slot's posterior
                        g.sadr[size_t(p) * 2 + 1] = kSpaOnesBytes + 8u * N + cs * 8u;       // LDS address of the first message of the slot's check
                        g.bhead[b * 4] |= 1ull << (p & 63);
                        if (j + 1 == d) g.bhead[b * 4 + 1] |= 1ull << (p & 63);
                        g.bhead[b * 4 + 2] = std::max<uint64_t>(g.bhead[b * 4 + 2], d);
                        for (uint32_t s2 = 0; s2 < d; ++s2)
                            if (s2 != j) g.bmask[b * size_t(g.DM) + s2] |= 1ull << (p & 63);
                    }
                }
            }
            g.vinfo2 = g.vinfo;
            for (uint32_t i = 0; i < N; ++i)
                for (int w = 1; w <= 5; ++w) {
                    const uint32_t x = g.vinfo[size_t(i) * 8 + w];
                    g.vinfo2[size_t(i) * 8 + w] = ((x & 0xffffu) * 8u) | ((x >> 16) * 8u) << 16;
                }
        }
        return g;
    }
    throw std::runtime_error("LDPC table blob has no graph for this rate");
}

// The layouts are functions of the blob alone and take a few hundred milliseconds to place (the bank-aware descent above): contexts of one
// process share them (a gear-shifting caller re-creates contexts of the same eight codes over and over).
// The lock covers the map only: the placement of a code runs outside it (one std::shared_future per (blob, K), so concurrent mgpu_create
// calls for DIFFERENT codes - pool creation, multi-threaded gear shifting - place in parallel and calls for the SAME code wait for one
// placement), and a hit copies the graph outside the lock from a shared_ptr<const LdpcGraph>.
LdpcGraph load_graph(int K, const uint8_t* blob, size_t size) {
    typedef std::shared_ptr<const LdpcGraph> Ptr;
    static std::mutex m;
    static std::map<std::pair<uint64_t, int>, std::shared_future<Ptr>> cache;
    uint64_t h = 1469598103934665603ull;                         // FNV-1a of the blob: a file-supplied table set (MERCURY_LDPC_TABLES) gets its own entries
    for (size_t i = 0; i < size; ++i) { h ^= blob[i]; h *= 1099511628211ull; }
    const std::pair<uint64_t, int> key(h ^ size, K);
    std::shared_future<Ptr> fut;
    std::promise<Ptr> mine;
    bool build = false;
    {
        std::lock_guard<std::mutex> lk(m);
        auto it = cache.find(key);
        if (it == cache.end()) { fut = mine.get_future().share(); cache.emplace(key, fut); build = true; }
        else fut = it->second;
    }
    if (build) {
        try { mine.set_value(std::make_shared<const LdpcGraph>(load_graph_uncached(K, blob, size))); }
        catch (...) {
            { std::lock_guard<std::mutex> lk(m); cache.erase(key); }        // a corrupt blob is not remembered: the next call reports it again
            mine.set_exception(std::current_exception());
        }
    }
    return *fut.get();
}

}  // namespace

// cl_FIR::design for type LPF with a Hamming window (fir_filter.cc:45-131)
static std::vector<double> design_lpf_hamming(double transition_bw, double cut, double fs) {
    int n = int(4.0 / (transition_bw / (fs / 2.0)));
    if (n % 2 == 0) ++n;
    std::vector<double> c(n);
    const double Ts = 1.0 / fs;
    c[n / 2] = 1;
    for (int i = 0; i < n / 2; ++i) {
        const double temp = 2 * M_PI * cut * double(n / 2 - i) * Ts;
        c[i] = std::sin(temp) / temp;
        c[n - i - 1] = c[i];
    }
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += c[i];
    for (int i = 0; i < n; ++i) c[i] /= sum;
    for (int i = 0; i < n; ++i) c[i] *= 0.54 - 0.46 * std::cos(2.0 * M_PI * double(i) / (n - 1));
    return c;
}

// cl_FIR::design (fir_filter.cc:45-162) for the two transmit filters, parameters of physical_config.cc:103-113:
// FIR_tx1 = high-pass at carrier - bandwidth/2 (spectral inversion of the low-pass), Hamming window;
// FIR_tx2 = low-pass at carrier + bandwidth/2, Blackman window; both with a 1 kHz transition band (97 taps at 48 kHz).
std::vector<double> design_tx_fir(int which, double carrier_hz) {
    const double bandwidth = 48000.0 * 50.0 / 256 / 4, fs = 48000.0, transition_bw = 1000.0;
    const bool hpf = which == 0;
    const double cut = hpf ? carrier_hz - bandwidth / 2 : carrier_hz + bandwidth / 2;
    int n = int(4.0 / (transition_bw / (fs / 2.0)));
    if (n % 2 == 0) ++n;
    std::vector<double> c(n);
    const double Ts = 1.0 / fs;
    c[n / 2] = 1;
    for (int i = 0; i < n / 2; ++i) {
        const double temp = 2 * M_PI * cut * double(n / 2 - i) * Ts;
        c[i] = std::sin(temp) / temp;
        c[n - i - 1] = c[i];
    }
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += c[i];
    for (int i = 0; i < n; ++i) c[i] /= sum;
    if (hpf) {
        for (int i = 0; i < n; ++i) c[i] *= -1;
        c[(n - 1) / 2] += 1;
        for (int i = 0; i < n; ++i) c[i] *= 0.54 - 0.46 * std::cos(2.0 * M_PI * double(i) / (n - 1));
    } else {
        for (int i = 0; i < n; ++i) c[i] *= 0.42 - 0.5 * std::cos(2.0 * M_PI * double(i) / n) + 0.08 * std::cos(4.0 * M_PI * double(i) / n);
    }
    return c;
}

// cl_telecom_system::get_pre_equalization_channel (telecom_system.cc:3108-3145) as init() reaches it (:1954-1958) in a process that
// has loaded this one configuration. An init-time table like the filters above, so it is computed on the host, statement by statement:
// 1000 x [Nc random symbols (the PRNG continues where cl_ofdm::init left it: __srandom(pilot seed 0), one draw per pilot, ofdm.cc:112-113,
// :940-951) -> psk.mod -> symbol_mod -> baseband_to_passband (phase origin 0, ofdm.cc:2293-2315) -> FIR_tx1 -> FIR_tx2 (fir_filter.cc:189-210)
// -> passband_to_baseband (FIR_rx_data, decimation 4, ofdm.cc:2316-2339) -> symbol_demod], mean of sent / received per carrier.
std::vector<Cplx> pre_equalization_channel(const ModeTables& t, double carrier_hz) {
    if (t.mfsk_M > 0) throw std::runtime_error("pre_equalization_channel: the OFDM modes only (telecom_system.cc:1954)");
    typedef std::complex<double> cd;
    const double fs = 48000.0, amplitude = std::sqrt(2.0), Ts = 1.0 / fs;       // carrier_amplitude: telecom_system.cc:69
    const int interp = 4, Nofdm = t.Nofdm, Nc = t.Nc, n = Nofdm * interp, nTries = 1000;
    const std::vector<double> f1 = design_tx_fir(0, carrier_hz), f2 = design_tx_fir(1, carrier_hz);
    const std::vector<double>& frx = t.fir_data;
    int bitrev[256];
    for (int i = 0; i < 256; ++i) { int r = 0; for (int b = 0; b < 8; ++b) if (i & (1 << b)) r |= 1 << (7 - b); bitrev[i] = r; }
    auto cmul = [](cd a, cd b) { return cd(a.real() * b.real() - a.imag() * b.imag(), a.real() * b.imag() + a.imag() * b.real()); };
    auto fft256 = [&](cd* v, bool inverse) {                                     // _fft_fast / _ifft_fast, ofdm.cc:310-377
        for (int i = 0; i < 256; ++i) if (i < bitrev[i]) std::swap(v[i], v[bitrev[i]]);
        for (int size = 2; size <= 256; size *= 2) {
            const int half = size / 2, step = 256 / size;
            for (int i = 0; i < 256; i += size)
                for (int j = 0; j < half; ++j) {
                    cd w(t.twiddle[j * step].re, t.twiddle[j * step].im);
                    if (inverse) w = std::conj(w);
                    const cd x = cmul(w, v[i + j + half]);
                    v[i + j + half] = v[i + j] - x;
                    v[i + j] = v[i + j] + x;
                }
        }
    };
    // std::complex operator/ in the reference's build = libgcc's __divdc3 (Smith's method; operands finite and normal here). Spelled out:
    // this file is compiled by clang, whose runtime divides complex numbers by another method that differs in the last ulp.
    auto cdiv = [](cd x, cd y) {
        const double a = x.real(), b = x.imag(), c = y.real(), d = y.imag();
        if (std::fabs(c) < std::fabs(d)) {
            const double ratio = c / d, denom = (c * ratio) + d;
            return cd(((a * ratio) + b) / denom, ((b * ratio) - a) / denom);
        }
        const double ratio = d / c, denom = (d * ratio) + c;
        return cd(((b * ratio) + a) / denom, (b - (a * ratio)) / denom);
    };
    auto lerp = [](cd a, double ax, cd b, double bx, double x) {                 // interpolate_linear, interpolator.cc:43-50
        const cd d = b - a;
        const double m = x - ax, q = bx - ax;
        cd r(d.real() * m, d.imag() * m);
        r = cd(r.real() / q, r.imag() / q);
        return a + r;
    };
    auto fir_real = [](const std::vector<double>& c, const double* in, double* out, int len) {   // cl_FIR::apply(double*), fir_filter.cc:189-210
        const int nt = int(c.size()), h = (nt - 1) / 2;
        for (int i = 0; i < len + nt - 1; ++i) {
            double acc = 0;
            for (int j = 0; j < nt; ++j) if (i - j >= 0 && i - j < len) acc += in[i - j] * c[j];
            if (i >= h && i < len + h) out[i - h] = acc;
        }
    };
    GlibcRandom rng(t.xp.pilot_seed);
    for (int i = 0; i < t.nPilots; ++i) (void)rng.next();
    std::vector<cd> acc(Nc, cd(0, 0)), mod(Nc), sym(Nofdm), bb(Nofdm), dem(Nc), mixed(n);
    std::vector<double> pb(n), t1(n), t2(n), cs(2 * size_t(n));
    // the carrier phases are the same in every try (phase origin 0 both ways). The reference's build evaluates cos and sin of one phase
    // as a single sincos() call (the compiler merges the pair); glibc's sincos is not bit-for-bit its cos + sin, so the same call here
    for (int i = 0; i < n; ++i) ::sincos(2 * M_PI * carrier_hz * double(i) * Ts, &cs[2 * i + 1], &cs[2 * i]);
    for (int tr = 0; tr < nTries; ++tr) {
        for (int i = 0; i < Nc; ++i) {                                           // psk.mod, psk.cc:259-272
            unsigned loc = 0;
            for (int j = 0; j < t.bps; ++j) { loc += unsigned(rng.next() % 2); loc <<= 1; }
            loc >>= 1;
            mod[i] = cd(t.constellation[loc].re, t.constellation[loc].im);
        }
        cd z[256];                                                               // symbol_mod, ofdm.cc:855-860
        for (auto& v : z) v = cd(0, 0);
        for (int j = 0; j < 25; ++j) z[j + 256 - 25] = mod[j];
        for (int j = 25; j < 50; ++j) z[j - 25 + 1] = mod[j];
        fft256(z, true);
        for (int j = 0; j < 256; ++j) sym[j + 16] = z[j];
        for (int j = 0; j < 16; ++j) sym[j] = z[j + 256 - 16];
        for (int i = 0; i < Nofdm; ++i)                                          // rational_resampler INTERPOLATION + mixer
            for (int j = 0; j < interp; ++j) {
                const cd v = i < Nofdm - 1 ? lerp(sym[i], 0, sym[i + 1], interp, j) : lerp(sym[Nofdm - 2], 0, sym[Nofdm - 1], interp, interp + j);
                const int k = i * interp + j;
                pb[k] = v.real() * amplitude * cs[2 * k];
                pb[k] += v.imag() * amplitude * cs[2 * k + 1];
            }
        fir_real(f1, pb.data(), t1.data(), n);
        fir_real(f2, t1.data(), t2.data(), n);
        for (int i = 0; i < n; ++i) mixed[i] = cd(t2[i] * amplitude * cs[2 * i], t2[i] * amplitude * cs[2 * i + 1]);
        {                                                                        // cl_FIR::apply(complex) fir_filter.cc:164-187 + DECIMATION ofdm.cc:2267-2278
            const int nt = int(frx.size()), h = (nt - 1) / 2;
            int index = 0;
            for (int m = 0; m < n; m += interp) {
                double ar = 0, ai = 0;
                const int i = m + h;
                for (int j = 0; j < nt; ++j)
                    if (i - j >= 0 && i - j < n) { ar += mixed[i - j].real() * frx[j]; ai += mixed[i - j].imag() * frx[j]; }
                bb[index++] = cd(ar, ai);
            }
        }
        for (int j = 0; j < 256; ++j) z[j] = bb[j + 16];                         // symbol_demod, ofdm.cc:862-867
        fft256(z, false);
        for (int j = 0; j < 256; ++j) z[j] = cd(z[j].real() / 256.0, z[j].imag() / 256.0);
        for (int j = 0; j < 25; ++j) dem[j] = z[j + 256 - 25];
        for (int j = 25; j < 50; ++j) dem[j] = z[j - 25 + 1];
        for (int i = 0; i < Nc; ++i) acc[i] += cdiv(mod[i], dem[i]);
    }
    std::vector<Cplx> out(Nc);
    for (int i = 0; i < Nc; ++i) out[i] = {acc[i].real() / double(nTries), acc[i].imag() / double(nTries)};
    return out;
}

// explicit (M, LDPC rate, preamble length, estimator) combinations outside the 17 rows of load_configuration: cfg id
// 1000 + (((log2(M) - 1) * 8 + rate_index) * 8 + (preamble_nSymb - 1)) * 2 + estimator, M in {2,4,8,16,32}, rate_index into
// {1,2,3,4,5,6,8,14}/16, preamble_nSymb 1..8, estimator 0 = ZERO_FORCE / 1 = LEAST_SQUARE; every other parameter as
// physical_config.cc / init() give it (Nc 50, Nfft 256, gi 1/16, Dx 1, Dy 3, LS window 21, seeds 0 / 1, pilot boost 1.33) (include/mercury_gpu.h: MGPU_CFG_EXPLICIT)
bool explicit_mode_row(int cfg, int* M, int* rate16, int* preamble, int* estimator) {
    static const int rates[8] = {1, 2, 3, 4, 5, 6, 8, 14};
    if (cfg < 1000 || cfg >= 1000 + 5 * 8 * 8 * 2) return false;
    const int v = cfg - 1000;
    *estimator = v & 1;
    *preamble = ((v >> 1) & 7) + 1;
    *rate16 = rates[(v >> 4) & 7];
    *M = 2 << (v >> 7);
    return true;
}

ModeTables build_mode_tables(int cfg, int mfsk_ctrl_mode, const uint8_t* blob, size_t blob_size, const ExplicitParams& xp) {
    const bool robust = cfg >= 100 && cfg <= 102;                             // common_defines.h:63-65
    ModeRow explicit_row = {0, 0, 0, 0};
    const bool is_explicit = explicit_mode_row(cfg, &explicit_row.M, &explicit_row.rate16, &explicit_row.preamble, &explicit_row.estimator);
    if (!robust && !is_explicit && (cfg < 0 || cfg > 16))
        throw std::runtime_error("cfg must be 0..16 (OFDM), 100..102 (ROBUST MFSK) or an MGPU_CFG_EXPLICIT id");
    const ModeRow robust_row = {200 /* MOD_MFSK, mfsk.h:28 */, cfg == 102 ? 4 : 1, 4, 1};   // telecom_system.cc:2625-2645
    const ModeRow& row = robust ? robust_row : is_explicit ? explicit_row : kModeTable[cfg];
    ModeTables t;
    t.xp = xp;
    t.lsw = xp.ls_window % 2 == 0 ? xp.ls_window + 1 : xp.ls_window;        // telecom_system.cc:2802-2809
    if (!(xp.pilot_boost > 0.0f) || !(xp.pilot_boost < 1e6f)) throw std::runtime_error("pilot_boost must be a positive finite number");
    if (t.lsw < 1 || t.lsw > 21) throw std::runtime_error("LS window must be 1..21 cells wide (the front-end reads at most 7 pilots of a window row)");
    t.cfg = cfg;
    t.M = row.M;
    t.preamble = row.preamble;
    t.estimator = row.estimator;
    t.amp_restore = (row.M == 2 || row.M == 4 || row.M == 8) ? 1 : 0;       // telecom_system.cc:2647-2654
    t.K = static_cast<int>(static_cast<float>(t.N) * (row.rate16 / 16.0f));  // ldpc.cc:65
    t.P = t.N - t.K;
    switch (row.M) {                                                         // telecom_system.cc:1818-1826
        case 2: t.Nsymb = 48; t.bps = 1; break;
        case 4: t.Nsymb = 24; t.bps = 2; break;
        case 8: t.Nsymb = 16; t.bps = 3; break;
        case 16: t.Nsymb = 12; t.bps = 4; break;
        default: t.Nsymb = 9; t.bps = 5; break;
    }
    if (xp.Nsymb != 0 || xp.Dy != 3) {                                        // telecom_system.cc:2775-2778: ofdm_Nsymb / ofdm_pilot_configurator_Dy as load_configuration copies them
        if (robust) throw std::runtime_error("explicit Nsymb / Dy: the MFSK modes take their frame length from the codeword (telecom_system.cc:1812-1816) and carry no pilots");
        if (xp.Nsymb < 0 || xp.Nsymb > 255 || xp.Dy < 1 || xp.Dy > 255) throw std::runtime_error("explicit Nsymb / Dy out of range");
        if (xp.Nsymb > 0) t.Nsymb = xp.Nsymb;
    }
    if (robust) {                                                            // mfsk.cc:48-78 via telecom_system.cc:2900-2907
        t.mfsk_M = cfg == 100 ? 32 : 16;
        t.mfsk_nstreams = cfg == 100 ? 1 : 2;
        t.mfsk_nbits = cfg == 100 ? 5 : 4;
        t.mfsk_hop = t.mfsk_M == 32 ? 13 : 7;                                // M is a power of two: the kernels reduce mod M with a mask
        const int global_offset = (t.Nc - t.mfsk_nstreams * t.mfsk_M) / 2;
        for (int k = 0; k < t.mfsk_nstreams; ++k) t.mfsk_off[k] = global_offset + k * t.mfsk_M;
        t.mfsk_amp = std::sqrt(double(t.Nc) / t.mfsk_nstreams);
        t.bps = t.mfsk_nbits * t.mfsk_nstreams;                              // M_eff = 2^bps, telecom_system.cc:1944-1946
        t.Nsymb = t.N / t.bps;                                               // telecom_system.cc:1812-1816
        t.ctrl_nbits = cfg == 100 ? 1200 : cfg == 101 ? 1400 : 0;            // telecom_system.cc:2973-2988
        t.ctrl_nsymb = t.ctrl_nbits / t.bps;
    }
    const int G = t.Nsymb * t.Nc;
    t.cell_type = robust ? std::vector<uint8_t>(G, 0) : make_pilot_lattice(t.Nsymb, t.Nc, xp.Dy);   // MFSK frames carry no pilots
    t.pilot_boost = static_cast<double>(xp.pilot_boost);                      // physical_config.h:53 (float)
    t.pilot_val.assign(G, 0.0);
    if (!robust) {   // DBPSK pilot sequence, ofdm.cc:940-951
        GlibcRandom rng(xp.pilot_seed);
        int last = 0;
        for (int c = 0; c < G; ++c) {
            if (!t.cell_type[c]) continue;
            const int pv = (rng.next() % 2) ^ last;
            t.pilot_val[c] = double(2 * pv - 1) * t.pilot_boost;
            last = pv;
            ++t.nPilots;
        }
    }
    t.nData = robust ? t.Nsymb : G - t.nPilots;                              // MFSK: one "symbol" per OFDM symbol period
    t.nBits = t.nData * t.bps;                                               // data_container.cc:93
    t.nVirtual = t.N - t.nBits;
    t.nReal = t.nBits - t.P;
    if (t.nVirtual < 0 || t.nVirtual > t.K || t.nReal < 24)
        throw std::runtime_error("frame geometry: the data cells hold " + std::to_string(t.nBits) + " bits - more than a codeword, fewer than its parity + a payload byte, or more virtual bits than information bits");
    t.bit_blk = t.nBits / 10;                                                // telecom_system.cc:2910-2911
    t.tf_blk = t.nData / 10;
    t.payload_bytes = (t.nReal - 16) / 8;                                    // telecom_system.cc:332-335
    t.payload_stride = (t.nReal + 7) / 8;
    {   // set_mfsk_ctrl_mode / get_active_nsymb / get_active_nbits, telecom_system.cc:1572-1585
        const bool on = mfsk_ctrl_mode && robust && t.ctrl_nbits > 0 && t.ctrl_nbits < t.nBits;
        t.active_nsymb = (on && t.ctrl_nsymb > 0) ? t.ctrl_nsymb : t.Nsymb;
        t.active_nbits = (on && t.ctrl_nbits > 0) ? t.ctrl_nbits : t.nBits;
    }
    t.frame_samples = t.active_nsymb * t.Nofdm;
    if (!robust) t.constellation = make_constellation(t.M);
    {   // telecom_system.cc:1961-1966
        GlibcRandom rng(xp.scrambler_seed);
        t.scrambler.resize(t.N);
        for (int i = 0; i < t.N; ++i) t.scrambler[i] = uint8_t(rng.next() % 2);
    }
    t.twiddle.resize(128);
    for (int k = 0; k < 128; ++k) {                                          // ofdm.cc:266-271
        const double angle = -2.0 * M_PI * k / 256;
        t.twiddle[k] = {std::cos(angle), std::sin(angle)};
    }
    // ---- RX gathers ---------------------------------------------------------------------
    std::vector<uint16_t> data_cell;                                         // deframer ofdm.cc:837-852
    for (int c = 0; c < G && int(data_cell.size()) < t.nData; ++c) if (!t.cell_type[c]) data_cell.push_back(uint16_t(c));
    auto deint_src = [](int n, int bs) {                                     // interleaver.cc:94-109
        std::vector<int> src(n);
        const int nb = n / bs;
        for (int i = 0; i < nb; ++i) for (int j = 0; j < bs; ++j) src[i * bs + j] = j * nb + i;
        for (int i = nb * bs; i < n; ++i) src[i] = i;
        return src;
    };
    const std::vector<int> tf = deint_src(t.nData, t.tf_blk);
    t.sym_src.resize(t.nData);
    t.tf_inv.resize(t.nData);
    for (int k = 0; k < t.nData; ++k) { t.sym_src[k] = data_cell[tf[k]]; t.tf_inv[tf[k]] = uint16_t(k); }
    t.data_cell = data_cell;
    const std::vector<int> bd = deint_src(t.nBits, t.bit_blk);
    // decoder input p <- de-interleaved index: telecom_system.cc:1300-1308
    t.llr_src.resize(t.N);
    for (int p = 0; p < t.N; ++p) {
        int d;                       // index into the bit-de-interleaved vector
        if (p < t.nReal) d = p;
        else if (p < t.nReal + t.nVirtual) d = p - t.nReal;       // virtual bits copy LLRs of bits [0,nVirtual)
        else d = p - t.nVirtual;                                  // parity shifted up by nVirtual
        t.llr_src[p] = uint16_t(bd[d]);
    }
    if (robust) {                    // nVirtual == 0: the gather is a permutation, the MFSK demapper scatters through its inverse
        if (t.nVirtual != 0) throw std::runtime_error("MFSK mode with virtual bits");
        t.llr_dst.assign(t.nBits, 0);
        for (int p = 0; p < t.N; ++p) t.llr_dst[t.llr_src[p]] = uint16_t(p);
    }
    // LS weights: x' = x / sum(x*x) with x = +-boost, sums accumulated sequentially (misc.cc:73-91)
    t.ls_weight.assign(size_t(t.lsw) * t.lsw + 1, 0.0);
    {
        double s = 0;
        for (size_t n = 1; n < t.ls_weight.size(); ++n) {
            s += t.pilot_boost * t.pilot_boost;
            t.ls_weight[n] = t.pilot_boost * (1.0 / s);
        }
    }
    // ---- TX permutations (synthetic generator only) ----------------------------------------
    t.bit_il.resize(t.nBits);
    for (int i = 0; i < t.nBits; ++i) t.bit_il[bd[i]] = uint16_t(i);  // interleaver is the inverse gather: out[bd[i]] = in[i]
    t.sym_cell.resize(t.nData);
    for (int k = 0; k < t.nData; ++k) t.sym_cell[k] = t.sym_src[k];   // symbol k lands where the RX reads it back
    // ---- preamble carriers (transmit path): the mode's known symbols in front of every frame ----
    t.preamble_carriers.assign(size_t(t.preamble) * t.Nc, Cplx{0.0, 0.0});
    if (t.mfsk_M > 0) {                 // cl_mfsk::generate_preamble, mfsk.cc:172-193; tones :82-95
        static constexpr int kTones32[4] = {4, 20, 12, 28}, kTones16[4] = {2, 10, 6, 14};
        const int* tones = t.mfsk_M == 32 ? kTones32 : kTones16;
        for (int s = 0; s < t.preamble; ++s)
            for (int st = 0; st < t.mfsk_nstreams; ++st) t.preamble_carriers[size_t(s) * t.Nc + t.mfsk_off[st] + tones[s % 4]] = Cplx{t.mfsk_amp, 0.0};
    } else {                            // cl_preamble_configurator::configure + init, ofdm.cc:1127-1232 (QPSK, seed 1, telecom_system.cc:2837-2841)
        GlibcRandom rng(xp.preamble_seed);
        const double s2 = std::sqrt(2.0);
        for (int i = 0; i < t.preamble; ++i)
            for (int j = 0; j < t.Nc; ++j) {
                const int bin = j < t.Nc / 2 ? j + t.Nfft - t.Nc / 2 : j - t.Nc / 2 + 1;    // zero_padder placement, start_shift = 1
                if (bin % 2 == 1) continue;                                               // odd FFT bins stay empty: two identical halves in time
                // std::complex<double>(2*(__random()%2)-1, 2*(__random()%2)-1): the reference's compiler draws the imaginary part first
                const int im = 2 * (rng.next() % 2) - 1;
                const int re = 2 * (rng.next() % 2) - 1;
                t.preamble_carriers[size_t(i) * t.Nc + j] = Cplx{double(re) / s2, double(im) / s2};
            }
    }
    {   // physical_config.cc:80,90-98 ; telecom_system.cc:1569,1910-1922
        const double bandwidth = 48000.0 * 50.0 / 256 / 4, fs = 48000.0;
        t.fir_time_sync = design_lpf_hamming(3000.0, 0.9 * bandwidth / 2, fs);
        t.fir_data = design_lpf_hamming(3000.0, 1.0 * bandwidth / 2, fs);
    }
    t.graph = load_graph(t.K, blob, blob_size);
    if (t.graph.P != t.P) throw std::runtime_error("LDPC graph does not match the mode");
    return t;
}

}  // namespace mgpu
