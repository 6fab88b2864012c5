// scl_big.h -- SCL / CA-SCL for shapes whose LLR levels do not fit a CU's LDS ("LLRs spill HBM": BASELINE
// config 5, N = 4096 L = 32; also N = 1024 L = 32 in f64).  Same algorithm and operation order as
// scl_generic.h (reference: SCLdecode, SCL_1024.c:560-674; CASCL, CASCL_1024_L8.c:613-755), other storage:
//
//   One codeword per wavefront (four independent wavefronts per workgroup, sharing the look-up tables);
//   lane = (path p, position pos), S = 64 / L lanes per path.
//   LLR level t <= TL (the 2^(TL+1) - 1 lowest values of a path): LDS, lowA[L][2^(TL+1)], level t at offset 2^t.
//   LLR level t > TL: per-wave slice of a global scratch buffer, hiA[L][N]; channel LLRs chg[N].
//     Written with plain stores (write-through to L2), read back with sc1 loads, vmcnt drained in between.
//   Partial sums, bit-packed (level t, element e <-> bit 2^t + e): bits < 32 in a register, levels 5..TB in
//     LDS, levels > TB in the scratch slice.
//   The decoder is a chain of dependent LDS / L2 round trips, so throughput follows the number of resident
//   wavefronts, which the LDS footprint sets: TL and TB trade a few more L2 round trips per frame for more
//   codewords per CU (profiles/README.md).
//
//   Both the LLR levels AND the saved left-child partial sums are shared lazily between a path and its clones
//   through per-level pointer tables (ptrA / ptrB, 5 bits per level): a fork copies two 64-bit words, never a
//   row.  (The reference clones the whole factor graph, SCL_1024.c:451-478.)
//
//   Levels > TL are evaluated with all 64 lanes on consecutive elements of one path (of 64 / 2^t paths below
//   level 6): coalesced rows; levels <= TL by the path's own S lanes.
//
//   Pruning: candidate (p, b) is owned by lane p*S + b.  Every owner counts #{m : c_m <= c_own} over the 2L
//   keys broadcast from LDS; "count <= L" is the reference's strict "< median" (SCL_1024.c:610-632).  The
//   m-th both-survivor is paired with the m-th dead slot through a rank table in LDS (:636-661).
#pragma once
#include "polar_math.h"
#include "polar_lut.h"
#include "scl_generic.h"

namespace polar {

// CHK of the steps with several independent elements per lane: the one-round-trip table form in f64 (config 5 with the
// 4 / 7 / 1 split: + 1.6 %, N = 1024 L = 32: +- 0; profiles/r03_ab_experiments.txt run 39 -- it had measured slower while
// levels 4 and 5 still went through the scratch, run 23), the compact two-round-trip form in f32
template <typename R>
__device__ __forceinline__ R chk_wide(R a, R b, const Lut<R> &L)
{
    if constexpr (sizeof(R) == 8) return chk_lut1<R>(a, b, L);
    else return chk_lut<R>(a, b, L);
}

__device__ __forceinline__ uint32_t ld_bypass(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename R, int LOGL, int TLv, int TBv, int RLv = 0>
struct BigCfg {
    static constexpr int L = 1 << LOGL;
    static constexpr int S = 64 / L;
    static constexpr int WAVES = 4;                   // codewords per workgroup; they share only the look-up tables
    static constexpr int TL = TLv;                    // highest LLR level kept in LDS (3..5)
    static constexpr int TB = TBv;                    // highest partial-sum level kept in LDS (5..7)
    static constexpr int RL = RLv;                    // LLR levels TL+1 .. TL+RL: registers of the path's own lanes (0..2)
    static_assert(RL >= 0 && RL <= 2 && TLv + RLv <= TBv && (RLv == 0 || (2 << TLv) >= 2 * S), "register levels");
    static constexpr int LOW = (2 << TL) + 1;         // row stride of lowA: odd, so that the paths fall into different banks
    static constexpr int WLU = 1 << (TB - 4);         // partial-sum words per path in LDS
    static constexpr int WL = WLU + 1;                // their row stride (odd)
    static_assert(TL >= 3 && TL <= 5 && TB >= 5 && TB <= 7, "storage split");
    struct __attribute__((aligned(16))) State { uint64_t ptrA, ptrB; R c1; uint32_t crc, bl0; };
    static constexpr size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }
    static constexpr size_t off_blw = align16(sizeof(R) * (size_t)L * LOW);
    static constexpr size_t off_cur = off_blw + sizeof(uint32_t) * (size_t)L * WL;
    static constexpr size_t off_cand = align16(off_cur + sizeof(uint32_t) * (size_t)L * WL);
    static constexpr size_t off_tbl = align16(off_cand + sizeof(R) * 2 * L);   // int[2L]: leader list / 32-bit metric keys
    static constexpr size_t off_st = align16(off_tbl + sizeof(int) * 2 * L);
    static constexpr size_t wave_bytes = align16(off_st + sizeof(State) * L);
    static constexpr size_t lds_bytes = wave_bytes * WAVES + Lut<R>::bytes;
    // scratch per wave, in bytes: chg[N] + hiA[L][N] reals, then gbl[L][N/32] + gcur[L][N/32] words
    static constexpr size_t scratch_bytes(int N) { return sizeof(R) * (size_t)(L + 1) * N + 2 * sizeof(uint32_t) * (size_t)L * (N / 32); }
};

__device__ __forceinline__ int metric_key_i(double x) { return __double2hiint(x); }
__device__ __forceinline__ int metric_key_i(float x) { return __float_as_int(x); }
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t big_sign(double x) { return (uint32_t)__double2hiint(x) >> 31; }
__device__ __forceinline__ uint32_t big_sign(float x) { return (uint32_t)__float_as_int(x) >> 31; }
template <int CTRL>
__device__ __forceinline__ int big_dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// "Every path keeps the branch its lambda favours": kf / ko = 32-bit key of the path's metric with the favoured / the other
// branch (the same value in all S lanes of a path).  True iff the largest favoured key of the wave is below the smallest other
// key: then the L favoured candidates are the L smallest of the 2L, all strictly below the median of SCL_1024.c:619-633, no
// path forks and none dies -- the ranking, its key exchange through LDS and the fork bookkeeping can be skipped.
template <int S>
__device__ __forceinline__ bool big_trivial_prune(int kf, int ko)
{
    int mx = kf;   // the largest favoured key of the wave, then ONE compare per lane: every other key must be above it
    if constexpr (S < 2) mx = max(mx, big_dpp<0x121>(mx));   // row_ror:1
    if constexpr (S < 4) mx = max(mx, big_dpp<0x122>(mx));
    if constexpr (S < 8) mx = max(mx, big_dpp<0x124>(mx));
    if constexpr (S < 16) mx = max(mx, big_dpp<0x128>(mx));
    if constexpr (S < 32) {
        auto a = __builtin_amdgcn_permlane16_swap((uint32_t)mx, (uint32_t)mx, false, false);
        mx = max((int)a[0], (int)a[1]);
    }
    {
        auto a = __builtin_amdgcn_permlane32_swap((uint32_t)mx, (uint32_t)mx, false, false);
        mx = max((int)a[0], (int)a[1]);
    }
    return __ballot(mx >= ko) == 0ull;
}

// everything a wave wrote (LDS and scratch) is visible to its own later loads; no other wave ever reads it
__device__ __forceinline__ void wave_sync() { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

#ifdef POLAR_STAMPS  // diagnostic build: s_memtime per section, summed per wave, added to P.dbg[]
#define BIG_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsec[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define BIG_STAMP(i) do { } while (0)
#endif
#ifdef POLAR_MARKS  // static instruction accounting (tools/count_marks.py): comments in the ISA
#define BIG_MARK(name) __asm__ volatile("; MARK " name)
#else
#define BIG_MARK(name) do { } while (0)
#endif

// One register level (RL = 1) is worth it only at four wavefronts per SIMD (128 VGPRs): the decoder is a chain of dependent
// round trips and its rate follows the number of resident wavefronts.
// CH = 1: the chains of f steps below a step at level >= 7 run in one pass (chain() below).  It needs more registers for its
// loads in flight, so the kernel then runs three wavefronts per SIMD; measured on N = 4096, L = 32 (BASELINE config 5): +9 %,
// a third less fabric read traffic; on N = 1024, L = 32 the four-wavefront kernel without it stays ahead.
template <typename R, typename IN, int LOGL, int TLv, int TBv, int RLv = 0, int CH = 0>
#ifndef POLAR_BIG_CH_WAVES
#define POLAR_BIG_CH_WAVES 3
#endif
__global__ __launch_bounds__(256, RLv == 1 ? (CH ? POLAR_BIG_CH_WAVES : 4) : 1) void k_scl_big(SclParams P)
{
#ifdef POLAR_STAMPS
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    using Cfg = BigCfg<R, LOGL, TLv, TBv, RLv>;
    using State = typename Cfg::State;
    constexpr int L = Cfg::L, S = Cfg::S, LOW = Cfg::LOW, WL = Cfg::WL, TL = Cfg::TL, TB = Cfg::TB, WAVES = Cfg::WAVES, RL = Cfg::RL;
    constexpr int TR = TL + RL;                         // highest level that is not in the scratch slice
    constexpr int PER1 = (RL >= 1) ? (2 << TL) / S : 1; // register level TL+1: element pos + S*k of the path in ra[k]
    constexpr int PER2 = (RL >= 2) ? (4 << TL) / S : 1; // register level TL+2: rb[k]
    static_assert(S >= 2, "one candidate per lane needs 2L <= 64");
    const int N = P.N, n = P.n, NW = N >> 5;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = lane / S, pos = lane % S;
    const uint64_t below = (1ull << (p * S)) - 1ull;   // lanes of lower-numbered paths

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *mine_mem = smem + (size_t)wave * Cfg::wave_bytes;
    R *lowA = reinterpret_cast<R *>(mine_mem);
    uint32_t *blw = reinterpret_cast<uint32_t *>(mine_mem + Cfg::off_blw);
    uint32_t *curw = reinterpret_cast<uint32_t *>(mine_mem + Cfg::off_cur);
    R *cand = reinterpret_cast<R *>(mine_mem + Cfg::off_cand);
    int *tbl = reinterpret_cast<int *>(mine_mem + Cfg::off_tbl);
    State *st = reinterpret_cast<State *>(mine_mem + Cfg::off_st);
    unsigned char *lut_mem = smem + (size_t)WAVES * Cfg::wave_bytes;
    Lut<R>::build(lut_mem, threadIdx.x, blockDim.x);
    Lut<R> lut;
    lut.bind(lut_mem);

    const int slot = blockIdx.x * WAVES + wave, nslots = gridDim.x * WAVES;
    unsigned char *slice = reinterpret_cast<unsigned char *>(P.scratch) + (size_t)slot * Cfg::scratch_bytes(N);
    R *chg = reinterpret_cast<R *>(slice);
    R *hiA = chg + N;
    uint32_t *gbl = reinterpret_cast<uint32_t *>(hiA + (size_t)L * N);
    uint32_t *gcur = gbl + (size_t)L * NW;
    __syncthreads();   // the tables are built; from here on the waves of a workgroup never meet again

    for (int frame = slot; frame < P.B; frame = next_job_wave(P.queue, frame, nslots, P.B)) {
        {   // channel LLRs (SCL_1024.c:574-578)
            const IN *src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
            for (int i = lane; i < N; i += 64) {
                double v = (double)src[i];
                if (P.sigma > 0) v = llr_from_y(v, P.sigma);
                chg[i] = (R)v;
            }
        }
        wave_sync();

        R PM = R(0);
        R ra[PER1], rb[PER2];   // register levels (RL > 0); a row belongs to the slot whose lanes hold it, like a row of hiA
        (void)ra; (void)rb;
        uint64_t ptrA = 0, ptrB = 0;
        uint32_t crc = 0, bl0 = 0, cur0 = 0, fl = 0, fw = 0, ctv = 0;
        int act = 1;

        // ---- level t > TL from level t+1, all 64 lanes on consecutive elements ----
        // Paths that share the source buffer (and, for a g step, the partial-sum buffer) would compute the same
        // row: it is computed once, by the lowest such path ("leader"), and the others point at the leader's row.
        // Far above the leaves most of the list still shares its ancestors, so this removes most of the work and
        // of the L2 / HBM traffic of the upper levels without changing a single value.
        auto bulk = [&](int t, bool gstep) {
            const int h = 1 << t;
            const int my_src = (t + 1 == n) ? 0 : ptr_get<LOGL>(ptrA, t + 1);
            const int my_bits = ptr_get<LOGL>(ptrB, t);
            // g below level 5 takes its bits from the register word, not from a shared buffer
            const int key = !gstep ? my_src : (t >= 5) ? (my_src | (my_bits << LOGL)) : (my_src | (int)(((bl0 >> h) & ((1u << h) - 1u)) << LOGL));
            int leader = p;
            for (int k = act - 1; k >= 0; --k)
                if (__builtin_amdgcn_readlane(key, k * S) == key) leader = k;
            const uint64_t m_lead = __ballot(pos == 0 && p < act && leader == p);
            const int nlead = __popcll(m_lead);
            if (pos == 0 && p < act && leader == p) tbl[__popcll(m_lead & below)] = p;
            __asm__ volatile("" ::: "memory");
            if (t >= 6) {
                // work item = (leader q, pass k): elements e = 64k + lane of path q.  U items are loaded before the
                // first is used, so that U round trips to L2 overlap instead of queueing behind each other.
#ifndef POLAR_BIG_U
#define POLAR_BIG_U 8
#endif
                constexpr int U = CH ? POLAR_BIG_U : 4;   // the three-wavefront kernel has the registers for more loads in flight
                                                          // (config 5: 8 / 8 against 4 / 4: 97.1 -> 94.5 ms; 12 or 16 spill 25+ VGPRs)
                const int lp = t - 6, per = 1 << lp, total = nlead << lp;
                for (int it = 0; it < total; it += U) {
                    R a[U], b[U];
                    uint32_t wv[U];
                    int qs[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = min(it + u, total - 1);   // clamp: the tail repeats the last item
                        const int q = __builtin_amdgcn_readfirstlane(tbl[idx >> lp]), e = ((idx & (per - 1)) << 6) + lane;
                        qs[u] = q;
                        const int ss = __builtin_amdgcn_readlane(my_src, q * S);
                        const R *src = (t + 1 == n) ? chg : hiA + (size_t)ss * N + 2 * h;
                        a[u] = ld_bypass(src + e);
                        b[u] = ld_bypass(src + e + h);
                        if (gstep) {
                            const int bs = __builtin_amdgcn_readlane(my_bits, q * S);
                            wv[u] = (t > TB) ? ld_bypass(gbl + (size_t)bs * NW + ((h + e) >> 5)) : blw[bs * WL + ((h + e) >> 5)];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int idx = min(it + u, total - 1);
                        const int e = ((idx & (per - 1)) << 6) + lane;
                        R *out = hiA + (size_t)qs[u] * N + h;
                        out[e] = gstep ? g_bit<R>(a[u], b[u], wv[u], e & 31) : chk_wide<R>(a[u], b[u], lut);
                    }
                }
            } else {
                // t = 4 or 5 (only when TL < 5): 64 / 2^t leaders per pass, source slots fetched by lane.  The loads of
                // UL passes go out before the first result is needed (one round trip per UL passes instead of one per pass).
                const int total = nlead << t;   // elements over all leaders
#ifndef POLAR_BIG_UL
#define POLAR_BIG_UL 8
#endif
                constexpr int UL = CH ? POLAR_BIG_UL : 4;
                for (int it = 0; it < total; it += 64 * UL) {
                    R a[UL], b[UL];
                    uint32_t wv[UL];
                    int qq[UL];
#pragma unroll
                    for (int u = 0; u < UL; ++u) {
                        const int idx = it + 64 * u + lane;
                        const bool on = idx < total;
                        const int q = on ? tbl[idx >> t] : 0, e = idx & (h - 1);
                        qq[u] = q;
                        const int ss = __shfl(my_src, q * S);
                        const R *src = hiA + (size_t)ss * N + 2 * h;
                        a[u] = ld_bypass(src + e);
                        b[u] = ld_bypass(src + e + h);
                        wv[u] = 0;
                        if (gstep) {
                            const uint32_t w0 = __shfl(bl0, q * S);                 // t = 4: bits 16 + e of the register word
                            const int bs = __shfl(my_bits, q * S);
                            wv[u] = (t == 5) ? blw[bs * WL + 1] : (w0 >> h);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UL; ++u) {
                        const int idx = it + 64 * u + lane;
                        const int e = idx & (h - 1);
                        const R r = gstep ? g_bit<R>(a[u], b[u], wv[u], e) : chk_wide<R>(a[u], b[u], lut);
                        if (idx < total) hiA[(size_t)qq[u] * N + h + e] = r;
                    }
                }
            }
            if (p < act) ptrA = ptr_set<LOGL>(ptrA, t, leader);
            wave_sync();
            BIG_STAMP((t >= 6) ? 1 : 2);
        };
        // ---- levels d, d-1, ..., 6 in ONE pass (d >= 7): the step at level d (g, or f from the channel row), then the f steps
        // below it without reading a row back.  A lane owns the elements lane + 64 k of every row, and element k of level
        // t-1 is CHK of elements k and k + 2^(t-7) of level t -- both in the same lane -- so the whole f chain of a node is
        // a reduction tree inside each lane.  The 2^(d-6) elements of level d are visited in bit-reversed order of k, which
        // makes every pair, every pair of pairs, ... consecutive: one pending value per level, no row read back.  Every row
        // is still WRITTEN (the g step of its right child reads it later).  Same operations on the same operands as one
        // bulk() per level; a third of the fabric traffic of the upper levels (the read-back of every f step) and all but
        // one of the full drains per chain are gone.  The leaders of the step at level d are the leaders of every level
        // below it in the chain (f steps do not look at partial sums).
        auto chain = [&](int d, bool gstep) {
            const int h = 1 << d;
            const int my_src = (d + 1 == n) ? 0 : ptr_get<LOGL>(ptrA, d + 1);
            const int my_bits = ptr_get<LOGL>(ptrB, d);
            const int key = !gstep ? my_src : (my_src | (my_bits << LOGL));
            int leader = p;
            for (int k = act - 1; k >= 0; --k)
                if (__builtin_amdgcn_readlane(key, k * S) == key) leader = k;
            const uint64_t m_lead = __ballot(pos == 0 && p < act && leader == p);
            const int nlead = __popcll(m_lead);
            if (pos == 0 && p < act && leader == p) tbl[__popcll(m_lead & below)] = p;
            __asm__ volatile("" ::: "memory");
            const int LV = d - 6, K = 1 << LV;
#ifndef POLAR_BIG_CHAIN_NO_GROUP
            // Short chains (d = 7: two elements of level d per lane and leader; d = 8: four) would be one memory round trip per
            // LEADER: LG leaders share a round instead, sixteen loads in flight per lane as in the long chains.
            auto grouped = [&](auto PLc) {
                constexpr int PL = decltype(PLc)::value;   // elements of level d per lane and leader: 2 (d = 7) or 4 (d = 8)
                constexpr int LG = 8 / PL;                 // leaders per round: 4 or 2
                for (int li = 0; li < nlead; li += LG) {
                    R a[8], b[8];
                    uint32_t wv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    R *rows_u[LG];
#pragma unroll
                    for (int u = 0; u < LG; ++u) {
                        const int q = __builtin_amdgcn_readfirstlane(tbl[min(li + u, nlead - 1)]);   // past the end: the last leader again, not stored
                        const int ss = __builtin_amdgcn_readlane(my_src, q * S);
                        const int bs = __builtin_amdgcn_readlane(my_bits, q * S);
                        const R *src = (d + 1 == n) ? chg : hiA + (size_t)ss * N + 2 * h;
                        rows_u[u] = hiA + (size_t)q * N;
                        const uint32_t *gw = gbl + (size_t)bs * NW;
                        const uint32_t *lw = blw + bs * WL;
#pragma unroll
                        for (int x = 0; x < PL; ++x) {
                            // visiting order of the elements k: 0, 1 (K = 2) or 0, 2, 1, 3 (K = 4): pairs are (k, k + K/2)
                            const int k = (PL == 2) ? x : ((x & 1) * 2 + (x >> 1));
                            a[u * PL + x] = ld_bypass(src + (unsigned)(64 * k + lane));
                            b[u * PL + x] = ld_bypass(src + (unsigned)(64 * k + h + lane));
                            if (gstep) {
                                const int wi = (h + 64 * k + lane) >> 5;
                                wv[u * PL + x] = (d > TB) ? ld_bypass(gw + wi) : lw[wi];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < LG; ++u) {
                        if (li + u < nlead) {
                            R *rows = rows_u[u];
                            R v[PL];
#pragma unroll
                            for (int x = 0; x < PL; ++x) {
                                const int k = (PL == 2) ? x : ((x & 1) * 2 + (x >> 1));
                                v[x] = gstep ? g_bit<R>(a[u * PL + x], b[u * PL + x], wv[u * PL + x], lane & 31)
                                             : chk_wide<R>(a[u * PL + x], b[u * PL + x], lut);
                                rows[(unsigned)(h + 64 * k + lane)] = v[x];
                            }
                            if constexpr (PL == 2) {
                                const R w = chk_wide<R>(v[0], v[1], lut);                 // level 6, element 0
                                rows[(unsigned)(64 + lane)] = w;
                            } else {
                                const R w0 = chk_wide<R>(v[0], v[1], lut), w1 = chk_wide<R>(v[2], v[3], lut);   // level 7: elements 0, 1
                                rows[(unsigned)(128 + lane)] = w0;
                                rows[(unsigned)(128 + 64 + lane)] = w1;
                                const R x6 = chk_wide<R>(w0, w1, lut);                    // level 6, element 0
                                rows[(unsigned)(64 + lane)] = x6;
                            }
                        }
                    }
                }
            };
            if (LV == 1) {
                grouped(std::integral_constant<int, 2>{});
            } else if (LV == 2) {
                grouped(std::integral_constant<int, 4>{});
            } else
#endif
            for (int li = 0; li < nlead; ++li) {
                const int q = __builtin_amdgcn_readfirstlane(tbl[li]);
                const int ss = __builtin_amdgcn_readlane(my_src, q * S);
                const int bs = __builtin_amdgcn_readlane(my_bits, q * S);
                // wave-uniform bases, 32-bit lane offsets (scalar-base addressing: no 64-bit address registers per lane)
                const R *src = (d + 1 == n) ? chg : hiA + (size_t)ss * N + 2 * h;
                R *rows = hiA + (size_t)q * N;                   // level t of leader q: rows[2^t + 64 k + lane]
                const uint32_t *gw = gbl + (size_t)bs * NW;
                const uint32_t *lw = blw + bs * WL;
                // element k of level d (g or f), stored
                auto top_ld = [&](int k, R &a, R &b, uint32_t &wv) {
                    a = ld_bypass(src + (unsigned)(64 * k + lane));
                    b = ld_bypass(src + (unsigned)(64 * k + h + lane));
                    if (gstep) {
                        const int wi = (h + 64 * k + lane) >> 5;
                        wv = (d > TB) ? ld_bypass(gw + wi) : lw[wi];
                    }
                };
                auto top_ev = [&](int k, R a, R b, uint32_t wv) -> R {
                    const R v = gstep ? g_bit<R>(a, b, wv, lane & 31) : chk_wide<R>(a, b, lut);
                    rows[(unsigned)(h + 64 * k + lane)] = v;
                    return v;
                };
                auto fnode = [&](int t, int k, R x, R y) -> R {   // element k of level t from its two parents, stored
                    const R v = chk_wide<R>(x, y, lut);
                    rows[(unsigned)((1 << t) + 64 * k + lane)] = v;
                    return v;
                };
                if (LV >= 3) {   // eight elements of level d per round: sixteen loads in flight per lane
                    R pend3 = R(0), pend4 = R(0);
                    for (int i = 0; i < K; i += 8) {
                        const int k0 = (int)(__brev((unsigned)i) >> (32 - LV));
                        const int e2 = K >> 1, e4 = K >> 2, e8 = K >> 3;
                        const int kk[8] = {k0, k0 + e2, k0 + e4, k0 + e4 + e2, k0 + e8, k0 + e8 + e2, k0 + e8 + e4, k0 + e8 + e4 + e2};
                        R a[8], b[8];
                        uint32_t wv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int u = 0; u < 8; ++u) top_ld(kk[u], a[u], b[u], wv[u]);
                        R w[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const R v0 = top_ev(kk[2 * u], a[2 * u], b[2 * u], wv[2 * u]);
                            const R v1 = top_ev(kk[2 * u + 1], a[2 * u + 1], b[2 * u + 1], wv[2 * u + 1]);
                            w[u] = fnode(d - 1, kk[2 * u], v0, v1);
                        }
                        const R x0 = fnode(d - 2, kk[0], w[0], w[1]), x1 = fnode(d - 2, kk[4], w[2], w[3]);
                        R x = fnode(d - 3, k0, x0, x1);
                        if (LV > 3) {
                            if (!((i >> 3) & 1)) {
                                pend3 = x;
                            } else {
                                x = fnode(d - 4, (int)(__brev((unsigned)(i & ~15)) >> (32 - LV)), pend3, x);
                                if (LV > 4) {
                                    if (!((i >> 4) & 1)) pend4 = x;
                                    else (void)fnode(d - 5, 0, pend4, x);
                                }
                            }
                        }
                    }
                }
            }
            if (p < act)
                for (int t = 6; t <= d; ++t) ptrA = ptr_set<LOGL>(ptrA, t, leader);
            wave_sync();
            BIG_STAMP(1);
        };
        // ---- level t in (TL, TL+RL]: the path's own S lanes, rows in registers (ra: level TL+1, rb: level TL+2) ----
        // No leader sharing here (every path evaluates its own row: VALU work instead of a scratch round trip per step).
        auto reg_c = [&](auto TC, auto GC) {
            constexpr int t = decltype(TC)::value, h = 1 << t;
            constexpr bool gstep = decltype(GC)::value;
            constexpr int PER = h / S;                       // elements per lane: e = pos + S*k
            static_assert(t > TL && t <= TR && PER >= 1, "register level");
            auto put = [&](int k, R v) {
                if constexpr (t == TL + 1) ra[k] = v;
                else rb[k] = v;
            };
            if (p < act) {
                const int ss = (t + 1 == n) ? 0 : ptr_get<LOGL>(ptrA, t + 1);
                uint32_t w0 = 0, w1 = 0;                     // partial-sum bits h + e of the path (g step)
                if (gstep) {
                    if constexpr (t < 5) w0 = bl0 >> h;
                    else {
                        const int bs = ptr_get<LOGL>(ptrB, t);
                        w0 = blw[bs * WL + (h >> 5)];
                        if constexpr (t == 6) w1 = blw[bs * WL + (h >> 5) + 1];
                    }
                }
                // element e = pos + S k: bit (S k) & 31 of the word shifted down by pos (a compile-time shift per k: no per-element
                // mask constants for the compiler to hoist out of the leaf loop and spill)
                const uint32_t ws0 = w0 >> pos, ws1 = w1 >> pos;
                auto gk = [&](R a, R b, int k) -> R { return g_bit<R>(a, b, (t == 6 && k * S >= 32) ? ws1 : ws0, (k * S) & 31); };
                if constexpr (t < TR) {
                    // source level t+1 is rb, in the lanes of slot ss
                    const int sl = ss * S + pos;
#pragma unroll
                    for (int k = 0; k < PER; ++k) {
                        R a = rb[k], b = rb[k + PER];
                        if (gstep) {
                            a = __shfl(a, sl);
                            b = __shfl(b, sl);
                        }
                        put(k, gstep ? gk(a, b, k) : chk_wide<R>(a, b, lut));
                    }
                } else {
                    // source level t+1 is a scratch row (written by bulk, drained by its wave_sync), or the channel row
                    const R *src = (t + 1 == n) ? chg : hiA + (size_t)ss * N + 2 * h;
#ifndef POLAR_BIG_REG_U
#define POLAR_BIG_REG_U 8
#endif
                    constexpr int U = PER < POLAR_BIG_REG_U ? PER : POLAR_BIG_REG_U;     // pairs of loads in flight per lane
#pragma unroll
                    for (int k0 = 0; k0 < PER; k0 += U) {
                        R a[U], b[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            a[u] = ld_bypass(src + pos + (k0 + u) * S);
                            b[u] = ld_bypass(src + pos + (k0 + u) * S + h);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            put(k0 + u, gstep ? gk(a[u], b[u], k0 + u) : chk_wide<R>(a[u], b[u], lut));
                    }
                }
                ptrA = ptr_set<LOGL>(ptrA, t, p);
            }
        };
        auto reg = [&](int t, bool gstep) {
            using std::integral_constant;
            if constexpr (RL >= 1) {
                if (t == TL + 1) {
                    if (gstep) reg_c(integral_constant<int, TL + 1>{}, integral_constant<bool, true>{});
                    else reg_c(integral_constant<int, TL + 1>{}, integral_constant<bool, false>{});
                }
            }
            if constexpr (RL >= 2) {
                if (t == TL + 2) {
                    if (gstep) reg_c(integral_constant<int, TL + 2>{}, integral_constant<bool, true>{});
                    else reg_c(integral_constant<int, TL + 2>{}, integral_constant<bool, false>{});
                }
            }
            BIG_STAMP(2);
        };
        // ---- level t <= TL from level t+1 by the path's own lanes ----
        auto low_c = [&](auto TC, auto GC) {
            constexpr int t = decltype(TC)::value, h = 1 << t;
            constexpr bool gstep = decltype(GC)::value;
            if (p < act) {
                const int ss = ptr_get<LOGL>(ptrA, t + 1);
                R *out = lowA + p * LOW + h;
                uint32_t wv = 0;
                if constexpr (t < 5) wv = bl0 >> h;  // bit 2^t + e of the register word
                else if (gstep) wv = blw[ptr_get<LOGL>(ptrB, 5) * WL + 1];
                const uint32_t wvp = wv >> pos;      // element e = pos + S k <-> bit S k of wvp (a compile-time shift per k)
                if constexpr (t == TL && RL >= 1) {
                    // level TL+1 is in the registers of slot ss: an f step follows the step that wrote the path's own row
                    // (ss == p, no exchange), a g step fetches the owner's lanes (SCL_1024.c:404-421 through the pointer)
                    constexpr int PER = h / S;
                    static_assert(PER >= 1 && 2 * PER == PER1, "register level above the LDS levels");
                    const int sl = ss * S + pos;
#pragma unroll
                    for (int k = 0; k < PER; ++k) {
                        const int e = pos + k * S;
                        R a = ra[k], b = ra[k + PER];
                        if (gstep) {
                            a = __shfl(a, sl);
                            b = __shfl(b, sl);
                        }
                        out[e] = gstep ? g_bit<R>(a, b, wvp, k * S) : chk_wide<R>(a, b, lut);
                    }
                } else if (t == TL) {
                    const R *src = hiA + (size_t)ss * N + 2 * h;
                    constexpr int PER = h / S;                             // elements per lane
                    constexpr int U = PER < 1 ? 1 : (PER < 8 ? PER : 8);   // loads in flight per lane
                    for (int e0 = pos; e0 < h; e0 += S * U) {
                        R a[U], b[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            a[u] = ld_bypass(src + e0 + u * S);
                            b[u] = ld_bypass(src + e0 + u * S + h);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int e = e0 + u * S;
                            out[e] = gstep ? g_bit<R>(a[u], b[u], wv, e) : chk_wide<R>(a[u], b[u], lut);
                        }
                    }
                } else {
                    const R *src = lowA + ss * LOW + 2 * h;
                    constexpr int PER = (h + S - 1) / S;
                    if (h >= S || pos < h) {
#pragma unroll
                        for (int k = 0; k < PER; ++k) {
                            const int e = pos + k * S;
                            const R a = src[e], b = src[e + h];
                            out[e] = gstep ? g_bit<R>(a, b, wvp, k * S) : chk_lut1<R>(a, b, lut);
                        }
                    }
                }
                ptrA = ptr_set<LOGL>(ptrA, t, p);
            }
            __asm__ volatile("" ::: "memory");  // LDS executes one wave's operations in order
        };
        auto low = [&](int t, bool gstep) {
            using std::integral_constant;
#define POLAR_LOW_CASE(T) \
    case T: if constexpr (T <= TL) { if (gstep) low_c(integral_constant<int, T>{}, integral_constant<bool, true>{}); \
                                     else low_c(integral_constant<int, T>{}, integral_constant<bool, false>{}); } break;
            switch (t) {
                POLAR_LOW_CASE(0) POLAR_LOW_CASE(1) POLAR_LOW_CASE(2) POLAR_LOW_CASE(3) POLAR_LOW_CASE(4) POLAR_LOW_CASE(5)
            }
#undef POLAR_LOW_CASE
        };

        // The register levels are dead when a chain starts: every path is about to rewrite its own rows on the way down (f
        // steps at every level below the chain) before anybody reads them again.  Saying so frees their registers for the
        // chain's loads in flight (their writes are predicated on p < act, which hides that from the compiler).
        auto kill_regs = [&]() {
            if constexpr (RL >= 1) {
#pragma unroll
                for (int k = 0; k < PER1; ++k) ra[k] = R(0);
            }
            if constexpr (RL >= 2) {
#pragma unroll
                for (int k = 0; k < PER2; ++k) rb[k] = R(0);
            }
        };
        for (int j = 0; j < N; ++j) {
            // per-leaf constants, one coalesced vector load per 64 leaves (a scalar load per leaf would put its whole
            // latency in front of the next LDS wait: both count on lgkmcnt)
            if ((j & 63) == 0) {
                ctv = P.crc_tab ? P.crc_tab[j + lane] : 0u;      // CRC remainder of leaf j + lane
                fw = P.frozen[(j >> 5) + (lane & 1)];            // frozen masks of leaves j..j+31 / j+32..j+63
            }
            // ================= LLR of leaf j for every active path =================
            BIG_MARK("big_llr");
            int tf = n - 1;
            if (CH && TR <= 5 && j == 0 && n - 1 >= 7) {
                kill_regs();
                chain(n - 1, false);
                tf = 5;
            } else if (j > 0) {
                const int d = __builtin_ctz((unsigned)j);
                if (CH && TR <= 5 && d >= 7) {
                    kill_regs();
                    chain(d, true);
                    tf = 5;
                } else {
                if (d > TR) bulk(d, true);
                else if (d > TL) reg(d, true);
                else low(d, true);
                tf = d - 1;
                }
            }
            for (int t = tf; t >= 0; --t) {
                if (t > TR) bulk(t, false);
                else if (t > TL) reg(t, false);
                else low(t, false);
            }
            const R lam = (p < act) ? lowA[p * LOW + 1] : R(0);
            BIG_STAMP(3);
            BIG_MARK("big_decide");

            // ================= decision =================
            const bool frozen = (__builtin_amdgcn_readlane(fw, (j >> 5) & 1) >> (j & 31)) & 1;
            int bit = 0;
            if (frozen) {
                if (p < act) PM += lut.tabv(lam) + negmax(lam);  // PHI(.,0), SCL_1024.c:601-604, :662-665
            } else if (act < L) {
                // phase 1: every path forks, clone k -> k + act (SCL_1024.c:586-600)
                const bool is_new = (p >= act) && (p < 2 * act);
                const int sg = is_new ? p - act : p;
                const int sl = sg * S + pos;
                const R lam_s = __shfl(lam, sl);
                const R pm_s = __shfl(PM, sl);
                ptrA = __shfl(ptrA, sl);
                ptrB = __shfl(ptrB, sl);
                crc = __shfl(crc, sl);
                bl0 = __shfl(bl0, sl);
                if (is_new) {
                    bit = 1;
                    PM = pm_s + (lut.tabv(lam_s) + posmax(lam_s));
                } else if (p < act) {
                    PM = PM + (lut.tabv(lam) + negmax(lam));
                }
                act *= 2;
            } else {
                // phase 2: keep the L best of 2L candidates (SCL_1024.c:610-661)
                // PHI of the branch lambda favours is T(|lambda|), of the other one T(|lambda|) + |lambda| (:481-502; T + 0
                // is T), so cb / cw ARE c0 / c1 in the order the sign of lambda says (lambda = +-0: cb == cw, never trivial)
                const R tt = lut.tabv(lam);
                const R cb = PM + tt, cw = PM + (tt + absr(lam));
                const uint32_t lneg = big_sign(lam);
                if (big_trivial_prune<S>(metric_key_i(cb), metric_key_i(cw))) {
                    bit = (int)lneg;
                    PM = cb;
                } else {
                const R c0 = lneg ? cw : cb;
                const R c1 = lneg ? cb : cw;
                const R mine = (pos == 0) ? c0 : c1;
                // First on 32-bit keys (the bits of a float, the high word of a double: monotone for metrics >= 0):
                // t = key_m - key_own - 1 is negative iff key_m <= key_own, and one v_alignbit shifts that sign bit
                // into a 32-candidate accumulator.  If exactly L candidates come out with count <= L they are the
                // L smallest for the full values too; otherwise two keys tie at the boundary (or two metrics are
                // equal) and the count is redone on the full values.
                const int kown1 = metric_key_i(mine) + 1;
                if (pos < 2) tbl[2 * p + pos] = kown1 - 1;
                __asm__ volatile("" ::: "memory");
                uint32_t acc0 = 0, acc1 = 0;
#pragma unroll
                for (int m = 0; m < 2 * L; m += 4) {
                    const i4 k = *reinterpret_cast<const i4 *>(tbl + m);
                    uint32_t &acc = (m < 32) ? acc0 : acc1;
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(k.x - kown1), 31);
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(k.y - kown1), 31);
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(k.z - kown1), 31);
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(k.w - kown1), 31);
                }
                int cnt = __popc(acc0) + __popc(acc1);
                uint64_t m_s0 = __ballot(pos == 0 && cnt <= L);
                uint64_t m_s1 = __ballot(pos == 1 && cnt <= L) >> 1;   // aligned to the lead lanes
                if (__popcll(m_s0) + __popcll(m_s1) < L) {
                    // strict "< med" with med = (L+1)-th smallest  <=>  #{m : c_m <= c} <= L
                    if (sizeof(R) == 8) fl |= 0x4u;   // POLAR_FLAG_RERANK (a float's key is the float: only a tie gets here)
                    if (pos < 2) cand[2 * p + pos] = mine;
                    __asm__ volatile("" ::: "memory");
                    cnt = 0;
#pragma unroll 8
                    for (int m = 0; m < 2 * L; m += 2) {
                        const R v0 = cand[m], v1 = cand[m + 1];
                        cnt += (v0 <= mine);
                        cnt += (v1 <= mine);
                    }
                    m_s0 = __ballot(pos == 0 && cnt <= L);
                    m_s1 = __ballot(pos == 1 && cnt <= L) >> 1;
                }
                const uint64_t m_both = m_s0 & m_s1;
                const uint64_t m_dead = __ballot(pos == 0) & ~(m_s0 | m_s1);
                if (__popcll(m_s0) + __popcll(m_s1) < L) fl |= 0x1u;  // median tie ("Oops!", :621-622)
                const bool s0 = (m_s0 >> (p * S)) & 1, s1 = (m_s1 >> (p * S)) & 1;
                // m-th both-survivor (ascending slot) forks into the m-th dead slot (:636-661)
                const int nboth = __popcll(m_both);
                bool refilled = false;
                if (nboth) {
                    if (s0 && s1 && pos == 0) {   // the m-th both-survivor leaves its state in row m
                        State me;
                        me.ptrA = ptrA; me.ptrB = ptrB; me.c1 = c1; me.crc = crc; me.bl0 = bl0;
                        st[__popcll(m_both & below)] = me;
                    }
                    __asm__ volatile("" ::: "memory");
                    const int rank_dead = __popcll(m_dead & below);
                    refilled = !s0 && !s1 && rank_dead < nboth;
                    if (refilled) {
                        const State src = st[rank_dead];
                        ptrA = src.ptrA; ptrB = src.ptrB; crc = src.crc; bl0 = src.bl0;
                        PM = src.c1;
                        bit = 1;
                    }
                }
                if (!refilled) {
                    if (s0) {
                        bit = 0;  // class 0 or the staying half of class 2
                        PM = c0;
                    } else if (s1) {
                        bit = 1;
                        PM = c1;
                    } else {
                        bit = 0;  // tie rule: an un-refilled dead slot continues as its 0-branch
                        PM = c0;
                    }
                }
                }
                __asm__ volatile("" ::: "memory");
            }

            BIG_STAMP(frozen ? 4 : 5);
            BIG_MARK("big_sums");
            // ================= partial sums (updateBit, SCL_1024.c:424-448) =================
            if (bit) crc ^= __builtin_amdgcn_readlane(ctv, j & 63);
            cur0 = (uint32_t)bit;
            int t = 0;
            while (t < n && ((j >> t) & 1)) {
                if (t < 5) {
                    const int h = 1 << t;
                    const uint32_t mask = (1u << h) - 1u;
                    const uint32_t l = (bl0 >> h) & mask;
                    const uint32_t c = cur0 & mask;
                    cur0 = (l ^ c) | (c << h);
                } else {
                    const int nw = 1 << (t - 5);
                    if (t == 5) {
                        if (pos == 0 && p < act) curw[p * WL] = cur0;
                        __asm__ volatile("" ::: "memory");
                    }
                    if (p < act) {
                        const int bs = ptr_get<LOGL>(ptrB, t);
                        if (t <= TB) {        // cur_t, bl_t, cur_{t+1} (<= WLU words) all in LDS
                            for (int w = pos; w < nw; w += S) {
                                const uint32_t c = curw[p * WL + w];
                                const uint32_t l = blw[bs * WL + nw + w];
                                curw[p * WL + w] = l ^ c;
                                curw[p * WL + w + nw] = c;
                            }
                        } else {              // result goes to the scratch slice
                            for (int w = pos; w < nw; w += S) {
                                const uint32_t c = (t == TB + 1) ? curw[p * WL + w] : ld_bypass(gcur + (size_t)p * NW + w);
                                const uint32_t l = ld_bypass(gbl + (size_t)bs * NW + nw + w);
                                gcur[(size_t)p * NW + w] = l ^ c;
                                gcur[(size_t)p * NW + w + nw] = c;
                            }
                        }
                    }
                    if (t <= TB) __asm__ volatile("" ::: "memory");
                    else wave_sync();
                }
                ++t;
            }
            if (t < n) {
                if (t < 5) {
                    const int h = 1 << t;
                    const uint32_t mask = (1u << h) - 1u;
                    bl0 = (bl0 & ~(mask << h)) | ((cur0 & mask) << h);
                } else {
                    const int nw = 1 << (t - 5);
                    if (p < act) {
                        if (t == 5) {
                            if (pos == 0) blw[p * WL + 1] = cur0;
                        } else if (t <= TB) {
                            for (int w = pos; w < nw; w += S) blw[p * WL + nw + w] = curw[p * WL + w];
                        } else {
                            for (int w = pos; w < nw; w += S)
                                gbl[(size_t)p * NW + nw + w] = (t == TB + 1) ? curw[p * WL + w] : ld_bypass(gcur + (size_t)p * NW + w);
                        }
                        ptrB = ptr_set<LOGL>(ptrB, t, p);
                    }
                    if (t <= TB) __asm__ volatile("" ::: "memory");
                    else wave_sync();
                }
            }
            BIG_STAMP(6);
            BIG_MARK("big_leaf_end");
        }

        // ================= choose the path (SCL_1024.c:667-674; CASCL_1024_L8.c:725-755) =================
        const bool pass = (P.crc_tab != nullptr) && (crc == 0);
        const bool any = __ballot(pass && p < act) != 0ull;
        int best = -1;
        R best_pm = R(0);
        for (int q = 0; q < act; ++q) {
            const R pq = __shfl(PM, q * S);
            const int okq = __shfl((int)(any ? pass : true), q * S);
            if (okq && (best < 0 || pq < best_pm)) {
                best = q;
                best_pm = pq;
            }
        }
        if (any) fl |= 0x2u;
        // x_hat of the chosen path: root partial sums (scratch slice); u_hat = x_hat * F^{(x)n}, in place
        uint32_t *xw = gcur + (size_t)best * NW;
        for (int w = lane; w < NW; w += 64) {
            uint32_t x = ld_bypass(xw + w);
            x ^= (x >> 1) & 0x55555555u;
            x ^= (x >> 2) & 0x33333333u;
            x ^= (x >> 4) & 0x0F0F0F0Fu;
            x ^= (x >> 8) & 0x00FF00FFu;
            x ^= (x >> 16) & 0x0000FFFFu;
            xw[w] = x;
        }
        wave_sync();
        for (int s = 5; s < n; ++s) {
            const int hw = 1 << (s - 5);
            for (int w = lane; w < NW; w += 64)
                if (!(w & hw)) xw[w] = ld_bypass(xw + w) ^ ld_bypass(xw + w + hw);
            wave_sync();
        }
        for (int w = lane; w < NW; w += 64) P.out_bits[(size_t)frame * NW + w] = ld_bypass(xw + w);
        if (lane == 0) {
            if (P.pm) P.pm[frame] = (double)best_pm;
            if (P.flags) P.flags[frame] = fl;
        }
        wave_sync();
        BIG_STAMP(7);
    }
#ifdef POLAR_STAMPS
    if (lane == 0 && P.dbg)
        for (int i = 0; i < 8; ++i) atomicAdd(&P.dbg[i], tsec[i]);
#endif
}

}  // namespace polar
