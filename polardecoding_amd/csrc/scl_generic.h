// scl_generic.h -- generic (any N, any L = 2^LOGL <= 32) SC / SCL / CA-SCL decode kernel.
//
// One codeword per wavefront (block = one 64-lane wave).  Lane = (path p, position pos):
// p = lane / S, pos = lane % S, S = 64 / L.  All LLR levels live in LDS:
//
//     ch[N]                    channel LLRs (level n), shared by all paths
//     alpha[L][N]              level t (2^t values) at offset 2^t of each slot's row
//     blw[L][N/32], curw[..]   saved left-child partial sums / working partial sums, bit-packed;
//                              level t, element e  <->  bit 2^t + e
//
// The reference clones whole factor graphs on every fork (copyPath / simpleCopy,
// SCL_1024.c:451-478: 81 % of its run time).  Here a fork copies a per-level pointer table
// (ptrA) instead: every slot owns one buffer per level, writes always go to the slot's own
// buffer, and in the lock-step schedule every live path rewrites level t before anybody reads
// it again.  The decided bits u_hat are never stored per path: the partial sums of the chosen
// path at the root are its codeword x_hat, and u_hat = x_hat * F^{(x)n} (F is an involution).
//
// Schedule = the reference's lazy recursion getLLR/updateBit (SCL_1024.c:404-448) in its
// natural-order form.  This kernel is the correctness baseline and
// the fallback for shapes without a tuned instantiation; scl_fast.h holds the tuned ones.
#pragma once
#include "polar_math.h"
#include "polar_lut.h"
#include "polar_params.h"

namespace polar {


template <int LOGL>
__device__ __forceinline__ int ptr_get(uint64_t tbl, int t)
{
    if (LOGL == 0) return 0;
    return (int)((tbl >> (t * LOGL)) & ((1u << LOGL) - 1));
}
template <int LOGL>
__device__ __forceinline__ uint64_t ptr_set(uint64_t tbl, int t, int v)
{
    if (LOGL == 0) return 0;
    const uint64_t m = (uint64_t)((1u << LOGL) - 1) << (t * LOGL);
    return (tbl & ~m) | ((uint64_t)v << (t * LOGL));
}

// scratch loads that must not be served from a stale per-CU L1 line (relaxed agent-scope atomic = sc1 load)
__device__ __forceinline__ double ld_bypass(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_bypass(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// GA = false: every LLR level of every path in LDS.  GA = true ("LLRs spill HBM", BASELINE config 5): the
// level arrays alpha[L][N] live in a per-workgroup slice of a global scratch buffer and the channel LLRs are
// read from the input; only the bit-packed partial sums stay in LDS.
template <typename R, typename IN, int LOGL, bool GA>
__global__ __launch_bounds__(64) void k_scl_generic(SclParams P)
{
    constexpr int L = 1 << LOGL;
    constexpr int S = 64 / L;
    const int N = P.N, n = P.n, NW = N >> 5;
    const int lane = threadIdx.x;
    const int p = lane / S, pos = lane % S;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *ch, *alpha;
    uint32_t *blw;
    if constexpr (GA) {
        ch = reinterpret_cast<R *>(P.scratch) + (size_t)blockIdx.x * (size_t)(L + 1) * N;
        alpha = ch + N;
        blw = reinterpret_cast<uint32_t *>(smem);
    } else {
        ch = reinterpret_cast<R *>(smem);
        alpha = ch + N;
        blw = reinterpret_cast<uint32_t *>(alpha + (size_t)L * N);
    }
    uint32_t *curw = blw + (size_t)L * NW;
    R *cand = reinterpret_cast<R *>(curw + (size_t)L * NW);
    // table-driven staircase (polar_lut.h): same bits as polar_math.h's chk / phi, a third of the instructions
    unsigned char *lut_mem = reinterpret_cast<unsigned char *>(cand + 2 * L);
    lut_mem += (16 - (reinterpret_cast<uintptr_t>(lut_mem) & 15)) & 15;
    Lut<R>::build(lut_mem, lane, 64);
    Lut<R> lut;
    lut.bind(lut_mem);
    __syncthreads();
    auto ld = [](const R *q) -> R {
        if constexpr (GA) return ld_bypass(q);
        else return *q;
    };

    for (int frame = blockIdx.x; frame < P.B; frame = next_job_wave(P.queue, frame, (int)gridDim.x, P.B)) {   // one wavefront per workgroup
        // ---- channel LLRs (SCL_1024.c:574-578) ----
        {
            const IN *src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
            for (int i = lane; i < N; i += 64) {
                double v = (double)src[i];
                if (P.sigma > 0) v = llr_from_y(v, P.sigma);
                ch[i] = (R)v;
            }
        }
        __syncthreads();

        R PM = R(0);
        uint64_t ptrA = 0;
        uint32_t crc = 0, bl0 = 0, cur0 = 0;
        uint32_t fl = 0;
        int act = 1;

        for (int j = 0; j < N; ++j) {
            // ================= LLR of leaf j for every active path =================
            int tf;
            if (j > 0) {
                const int d = __builtin_ctz((unsigned)j);
                const int h = 1 << d;
                if (p < act) {
                    const R *src = (d + 1 == n) ? ch : alpha + (size_t)ptr_get<LOGL>(ptrA, d + 1) * N + (2 << d);
                    R *out = alpha + (size_t)p * N + h;
                    for (int e = pos; e < h; e += S) {
                        const int bi = h + e;
                        const uint32_t wv = (bi < 32) ? bl0 : blw[p * NW + (bi >> 5)];
                        out[e] = gfun<R>(ld(src + e), ld(src + e + h), (wv >> (bi & 31)) & 1);
                    }
                    ptrA = ptr_set<LOGL>(ptrA, d, p);
                }
                __syncthreads();
                tf = d - 1;
            } else {
                tf = n - 1;
            }
            for (int t = tf; t >= 0; --t) {
                const int h = 1 << t;
                if (p < act) {
                    const R *src = (t + 1 == n) ? ch : alpha + (size_t)ptr_get<LOGL>(ptrA, t + 1) * N + (2 << t);
                    R *out = alpha + (size_t)p * N + h;
                    for (int e = pos; e < h; e += S) out[e] = chk_lut<R>(ld(src + e), ld(src + e + h), lut);
                    ptrA = ptr_set<LOGL>(ptrA, t, p);
                }
                __syncthreads();
            }
            const R lam = (p < act) ? ld(alpha + (size_t)p * N + 1) : R(0);

            // ================= decision =================
            const bool frozen = (P.frozen[j >> 5] >> (j & 31)) & 1;
            int bit = 0;
            if (P.sc_mode) {
                bit = (!frozen && lam < R(0)) ? 1 : 0;  // SC_128.c:426-431
            } else if (frozen) {
                if (p < act) PM += lut.tabv(lam) + negmax(lam);  // PHI(.,0), SCL_1024.c:601-604, :662-665
            } else if (act < L) {
                // phase 1: every path forks, clone k -> k + act (SCL_1024.c:586-600)
                const bool is_new = (p >= act) && (p < 2 * act);
                const int sg = is_new ? p - act : p;
                const int sl = sg * S + pos;
                const R lam_s = __shfl(lam, sl);
                const R pm_s = __shfl(PM, sl);
                ptrA = __shfl(ptrA, sl);
                crc = __shfl(crc, sl);
                bl0 = __shfl(bl0, sl);
                if (is_new) {
                    for (int w = 1 + pos; w < NW; w += S) blw[p * NW + w] = blw[sg * NW + w];
                    bit = 1;
                    PM = pm_s + (lut.tabv(lam_s) + posmax(lam_s));
                } else if (p < act) {
                    PM = PM + (lut.tabv(lam) + negmax(lam));
                }
                act *= 2;
                __syncthreads();
            } else {
                // phase 2: keep the L best of 2L candidates (SCL_1024.c:610-661)
                const R tt = lut.tabv(lam);
                const R c0 = PM + (tt + negmax(lam));
                const R c1 = PM + (tt + posmax(lam));
                if (pos == 0) {
                    cand[p] = c0;
                    cand[p + L] = c1;
                }
                __syncthreads();
                // strict "< med" with med = (L+1)-th smallest  <=>  #{m : c_m <= c} <= L
                int n0 = 0, n1 = 0;
                for (int m = 0; m < 2 * L; ++m) {
                    const R v = cand[m];
                    n0 += (v <= c0);
                    n1 += (v <= c1);
                }
                const bool s0 = n0 <= L, s1 = n1 <= L;
                const bool lead = pos == 0;
                const uint64_t m_s0 = __ballot(lead && s0);
                const uint64_t m_s1 = __ballot(lead && s1);
                const uint64_t m_both = m_s0 & m_s1;
                const uint64_t m_dead = __ballot(lead) & ~(m_s0 | m_s1);
                if (__popcll(m_s0) + __popcll(m_s1) < L) fl |= 0x1u;  // median tie ("Oops!", :621-622)
                const bool dead = !s0 && !s1;
                // m-th both-survivor (ascending slot) forks into the m-th dead slot (:636-661)
                const int myrank = __popcll(m_dead & ((1ull << (p * S)) - 1ull));
                int sg = p;
                bool refilled = false;
                {
                    uint64_t bm = m_both;
                    int cnt = 0;
                    while (bm) {
                        const int b = __builtin_ctzll(bm);
                        if (dead && cnt == myrank) {
                            sg = b / S;
                            refilled = true;
                        }
                        bm &= bm - 1;
                        ++cnt;
                    }
                }
                const int sl = sg * S + pos;
                const R c1_s = __shfl(c1, sl);
                ptrA = __shfl(ptrA, sl);
                crc = __shfl(crc, sl);
                bl0 = __shfl(bl0, sl);
                if (refilled) {
                    for (int w = 1 + pos; w < NW; w += S) blw[p * NW + w] = blw[sg * NW + w];
                    bit = 1;
                    PM = c1_s;
                } else if (s0) {
                    bit = 0;  // class 0 or the staying half of class 2
                    PM = c0;
                } else if (s1) {
                    bit = 1;
                    PM = c1;
                } else {
                    bit = 0;  // tie rule: an un-refilled dead slot continues as its 0-branch
                    PM = c0;
                }
                __syncthreads();
            }

            // ================= partial sums (updateBit, SCL_1024.c:424-448) =================
            if (P.crc_tab && bit) crc ^= P.crc_tab[j];
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
                        if (pos == 0 && p < act) curw[p * NW] = cur0;
                        __syncthreads();
                    }
                    if (p < act) {
                        for (int w = pos; w < nw; w += S) {
                            const uint32_t c = curw[p * NW + w];
                            const uint32_t l = blw[p * NW + nw + w];
                            curw[p * NW + w] = l ^ c;
                            curw[p * NW + w + nw] = c;
                        }
                    }
                    __syncthreads();
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
                    if (t == 5) {
                        if (pos == 0 && p < act) blw[p * NW + 1] = cur0;
                    } else if (p < act) {
                        for (int w = pos; w < nw; w += S) blw[p * NW + nw + w] = curw[p * NW + w];
                    }
                    __syncthreads();
                }
            }
        }

        // ================= choose the path (SCL_1024.c:667-674; CASCL_1024_L8.c:725-755) =================
        int best = 0;
        R best_pm = PM;
        if (!P.sc_mode) {
            const bool pass = (P.crc_tab != nullptr) && (crc == 0);
            const bool any = __ballot(pass && p < act) != 0ull;
            best = -1;
            best_pm = R(0);
            for (int q = 0; q < act; ++q) {
                const R pq = __shfl(PM, q * S);
                const int okq = __shfl((int)(any ? pass : true), q * S);
                if (okq && (best < 0 || pq < best_pm)) {
                    best = q;
                    best_pm = pq;
                }
            }
            if (any) fl |= 0x2u;
        }
        // x_hat of the chosen path: root partial sums; u_hat = x_hat * F^{(x)n}
        if (n <= 5) {
            uint32_t x = __shfl(cur0, best * S);
            for (int s = 0; s < n; ++s) {
                const uint32_t msk = (s == 0) ? 0x55555555u : (s == 1) ? 0x33333333u : (s == 2) ? 0x0F0F0F0Fu
                                   : (s == 3) ? 0x00FF00FFu : 0x0000FFFFu;
                x ^= (x >> (1 << s)) & msk;
            }
            if (lane == 0) P.out_bits[(size_t)frame * NW] = x;
        } else {
            uint32_t *xw = curw + (size_t)best * NW;
            for (int w = lane; w < NW; w += 64) {
                uint32_t x = xw[w];
                x ^= (x >> 1) & 0x55555555u;
                x ^= (x >> 2) & 0x33333333u;
                x ^= (x >> 4) & 0x0F0F0F0Fu;
                x ^= (x >> 8) & 0x00FF00FFu;
                x ^= (x >> 16) & 0x0000FFFFu;
                xw[w] = x;
            }
            __syncthreads();
            for (int s = 5; s < n; ++s) {
                const int hw = 1 << (s - 5);
                for (int w = lane; w < NW; w += 64)
                    if (!(w & hw)) xw[w] ^= xw[w + hw];
                __syncthreads();
            }
            for (int w = lane; w < NW; w += 64) P.out_bits[(size_t)frame * NW + w] = xw[w];
        }
        if (lane == 0) {
            if (P.pm) P.pm[frame] = P.sc_mode ? 0.0 : (double)best_pm;
            if (P.flags) P.flags[frame] = P.sc_mode ? 0u : fl;
        }
        __syncthreads();
    }
}

template <typename R, int LOGL>
constexpr size_t scl_generic_lds_bytes(int N, bool ga)
{
    return (ga ? 0 : sizeof(R) * (size_t)N * (1 + (1 << LOGL))) + 2 * sizeof(uint32_t) * (size_t)(N / 32) * (1 << LOGL) +
           sizeof(R) * 2 * (1 << LOGL) + 16 + Lut<R>::bytes;
}

}  // namespace polar
