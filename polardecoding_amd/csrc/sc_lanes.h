// sc_lanes.h -- SC decoder, ONE CODEWORD PER LANE (reference: SCdecode, SC_128.c:395-460, SC_1024.c:434-498).
//
// SC has no list to manage, so nothing in it needs two lanes to talk: a wavefront decodes 64 codewords in
// lock step, lane l working on frame 64*batch + l with the schedule (the frozen pattern) common to all lanes.
// There is no cross-lane instruction, no divergence and no idle lane anywhere in the decoder.
//
//   LLR levels 0..4 (the 31 lowest values): registers, fully unrolled 32-leaf block (template recursion).
//   LLR levels 5..n-1: the wavefront's slice of a global scratch buffer, element e of level t at
//     (2^t + e)*64 + lane -- every load and store of a level step is one coalesced 512-byte row.
//     Plain stores, sc1 loads, s_waitcnt vmcnt(0) between a step and its consumer.
//   Input rows [frame][N] are never copied: the two steps that read the channel level (f at the first block, g at the
//     middle one) take their operands straight from the caller's rows, lane l from frame l's row, sixteen consecutive
//     elements (one or half a cache line) per lane and burst, so that every line is fetched once per step.
//   Partial sums: bits < 32 in a register, levels >= 5 as words [w][lane] in LDS.
//   Subtrees without an information leaf are skipped altogether: their decisions are 0, their partial sums 0,
//   and SC (unlike SCL) needs nothing else from them, so every decision and every LLR that IS computed is the
//   reference's, operation for operation.
#pragma once
#include "polar_math.h"
#include "polar_lut.h"
#include "scl_generic.h"

namespace polar {

template <typename R>
struct ScLanesCfg {
    static constexpr int WAVES = 4;
#ifndef POLAR_SC_WAVES_PER_SIMD
#define POLAR_SC_WAVES_PER_SIMD 2
#endif
#ifndef POLAR_SC_WAVES_PER_SIMD_F32
#define POLAR_SC_WAVES_PER_SIMD_F32 2
#endif
    static constexpr size_t bits_bytes(int N) { return 4 * 64 * (size_t)(N / 32 + N / 64); }   // blw[NW][64] + curw[NW/2][64]
    static constexpr size_t wave_bytes(int N) { return (bits_bytes(N) + 15) & ~(size_t)15; }
    static constexpr size_t lds_bytes(int N) { return wave_bytes(N) * WAVES + Lut<R>::bytes; }
    static constexpr size_t scratch_bytes(int N) { return sizeof(R) * (size_t)N * 64; }   // levels 5 .. n-1: element indices < N
};

template <typename R>
struct ScLanes {
    const Lut<R> &lut;
    uint32_t fz;    // frozen mask of the current 32-leaf block (uniform)
    uint32_t dec;   // decisions of the block, bit k = leaf k

    // node of 2^T leaves starting at leaf K0 of the block, LLRs a[0..2^T); returns its partial sums
    template <int T, int K0>
    __device__ __forceinline__ uint32_t rec(const R *a)
    {
        constexpr uint32_t span = (T == 5) ? 0xFFFFFFFFu : ((1u << (1 << T)) - 1u);
        if (((fz >> K0) & span) == span) return 0u;   // no information leaf below: decisions 0, partial sums 0
        if constexpr (T == 0) {
            const uint32_t bit = (a[0] < R(0)) ? 1u : 0u;   // SC_128.c:426-431 (l >= 0 -> 0)
            dec |= bit << K0;
            return bit;
        } else {
            constexpr int h = 1 << (T - 1);
            constexpr uint32_t half = (1u << h) - 1u;
            uint32_t bl = 0, br = 0;
            if (((fz >> K0) & half) != half) {
                R l[h];
#pragma unroll
                for (int e = 0; e < h; ++e) l[e] = chk_lut<R>(a[e], a[e + h], lut);
                bl = rec<T - 1, K0>(l);
            }
            if (((fz >> (K0 + h)) & half) != half) {
                R r[h];
#pragma unroll
                for (int e = 0; e < h; ++e) r[e] = gfun<R>(a[e], a[e + h], (bl >> e) & 1u);
                br = rec<T - 1, K0 + h>(r);
            }
            return (bl ^ br) | (br << h);
        }
    }
};

template <typename R, typename IN>
__global__ __launch_bounds__(256, (sizeof(R) == 4 ? POLAR_SC_WAVES_PER_SIMD_F32 : POLAR_SC_WAVES_PER_SIMD)) void k_sc_lanes(SclParams P)
{
    using Cfg = ScLanesCfg<R>;
    const int N = P.N, n = P.n, NW = N >> 5;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *mine = smem + (size_t)wave * Cfg::wave_bytes(N);
    uint32_t *blw = reinterpret_cast<uint32_t *>(mine);          // [NW][64]: saved left partial sums
    uint32_t *curw = blw + (size_t)NW * 64;                      // [NW/2][64] working partial sums
    unsigned char *lut_mem = smem + (size_t)Cfg::WAVES * Cfg::wave_bytes(N);
    Lut<R>::build(lut_mem, threadIdx.x, blockDim.x);
    Lut<R> lut;
    lut.bind(lut_mem);
    __syncthreads();   // the waves of a workgroup share the tables and nothing else

    const int slot = blockIdx.x * Cfg::WAVES + wave, nslots = gridDim.x * Cfg::WAVES;
    R *lev = reinterpret_cast<R *>(reinterpret_cast<unsigned char *>(P.scratch) + (size_t)slot * Cfg::scratch_bytes(N)) + lane;
    // element idx = 2^t + e of this lane's codeword.  levb is lev, laundered once per 32-leaf block: otherwise every one of the
    // ~200 statically indexed row addresses is hoisted out of the batch loop as a loop invariant and most of them are spilled
    R *levb = lev;
    auto at = [&](int idx) -> R * { return levb + (size_t)idx * 64; };
    auto sync = [] { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); };
    const int nbatch = (P.B + 63) >> 6;

    for (int batch = slot; batch < nbatch; batch = next_job_wave(P.queue, batch, nslots, nbatch)) {
        const int frame0 = batch << 6;
        // ---- channel LLRs (SC_128.c:416-420): read in place, by the two steps at level n-1 ----
        const bool have = frame0 + lane < P.B;   // the ragged last batch: idle lanes compute on zeros and store nothing
        const IN *row = reinterpret_cast<const IN *>(P.in) + (size_t)(have ? frame0 + lane : 0) * N;
        const bool al16 = ((reinterpret_cast<uintptr_t>(P.in) | ((size_t)N * sizeof(IN))) & 15u) == 0;   // uniform
        auto chan16 = [&](int e0, R *dst) {   // elements e0 .. e0+15 of the lane's row (e0 a multiple of 16)
            IN raw[16];
            if (al16) {
                const IN *r = reinterpret_cast<const IN *>(__builtin_assume_aligned(row + e0, 16));
#pragma unroll
                for (int u = 0; u < 16; ++u) raw[u] = r[u];
            } else {
#pragma unroll
                for (int u = 0; u < 16; ++u) raw[u] = row[e0 + u];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                double v = have ? (double)raw[u] : 0.0;
                if (P.sigma > 0) v = llr_from_y(v, P.sigma);
                dst[u] = (R)v;
            }
        };
        sync();
        uint32_t fwv = 0;   // lane l: frozen word of block (b & ~63) + l
        for (int b = 0; b < NW; ++b) {
            levb = lev;
            __asm__ volatile("" : "+v"(levb));
            if ((b & 63) == 0) fwv = (b + lane < NW) ? P.frozen[b + lane] : 0xFFFFFFFFu;
            auto frozen_span = [&](int b0, int nwords) -> bool {   // all leaves of blocks b0 .. b0+nwords-1 frozen?
                uint32_t all = 0xFFFFFFFFu;
                for (int w = 0; w < nwords; ++w) all &= (uint32_t)__builtin_amdgcn_readlane((int)fwv, (b0 + w) & 63);
                return all == 0xFFFFFFFFu;
            };
            // ---- levels n-1 .. 5 above this block: g at the level where the path turns right, f below it ----
            // One pass computes level t (g or f of level t+1, or of the channel rows at t = n-1) and, when the subtree below
            // goes on to the left, level t-1 = f(level t) as well: the two halves of level t a level-(t-1) element needs are
            // produced together, so level t is written once and not read back by the f step that would follow.
            auto step = [&](int t, bool gstep, bool fuse) {
                const int h = 1 << t, hh = h >> 1;
                const uint32_t *bw = blw + (size_t)(h >> 5) * 64 + lane;   // left partial sums of level t (g step)
                if (t == n - 1) {
                    for (int e0 = 0; e0 < (fuse ? hh : h); e0 += 16) {
                        R v[2][16];
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            if (half == 1 && !fuse) break;
                            const int e = e0 + half * hh;
                            R a[16], c[16];
                            chan16(e, a);
                            chan16(e + h, c);
                            const uint32_t wv = gstep ? bw[(size_t)(e >> 5) * 64] : 0u;
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                v[half][u] = gstep ? gfun<R>(a[u], c[u], (wv >> ((e + u) & 31)) & 1u) : chk_lut<R>(a[u], c[u], lut);
                                *at(h + e + u) = v[half][u];
                            }
                        }
                        if (fuse) {
#pragma unroll
                            for (int u = 0; u < 16; ++u) *at(hh + e0 + u) = chk_lut<R>(v[0][u], v[1][u], lut);
                        }
                    }
                } else {
                    for (int e0 = 0; e0 < (fuse ? hh : h); e0 += 8) {
                        R v[2][8];
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            if (half == 1 && !fuse) break;
                            const int e = e0 + half * hh;
                            R a[8], c[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                a[u] = ld_bypass(at(2 * h + e + u));
                                c[u] = ld_bypass(at(2 * h + e + u + h));
                            }
                            const uint32_t wv = gstep ? bw[(size_t)(e >> 5) * 64] : 0u;
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                v[half][u] = gstep ? gfun<R>(a[u], c[u], (wv >> ((e + u) & 31)) & 1u) : chk_lut<R>(a[u], c[u], lut);
                                *at(h + e + u) = v[half][u];
                            }
                        }
                        if (fuse) {
#pragma unroll
                            for (int u = 0; u < 8; ++u) *at(hh + e0 + u) = chk_lut<R>(v[0][u], v[1][u], lut);
                        }
                    }
                }
                sync();
            };
            bool live = true;   // does the subtree we are descending into hold an information leaf?
            int td = n - 1;
            bool gstep = false;
            if (b > 0) {
                td = __builtin_ctz((unsigned)b) + 5;
                gstep = true;
            }
            // Long codes (n > 7): the steps that produce level 5 run after this loop and leave it in registers
            while (td >= 5) {
                live = !frozen_span(b, 1 << (td - 5));
                if (!live) break;
                if (n > 7 && td <= 6) break;
                const bool fuse = (td >= 6) && !frozen_span(b, 1 << (td - 6));
                step(td, gstep, fuse);
                td -= fuse ? 2 : 1;
                gstep = false;
            }
            R x5[32];   // level 5 of this block: read by the two halves of the block and by nobody else, so never stored
            if (live && n > 7 && td == 6) {   // level 6 from level 7 (stored: a later g step reads it), level 5 = f(level 6)
                const bool below = !frozen_span(b, 1);
                const uint32_t *bw = blw + (size_t)2 * 64 + lane;   // left partial sums of level 6: words 2, 3
#pragma unroll
                for (int e0 = 0; e0 < 32; e0 += 8) {
                    R v[2][8];
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int e = e0 + half * 32;
                        R a[8], c[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            a[u] = ld_bypass(at(128 + e + u));
                            c[u] = ld_bypass(at(192 + e + u));
                        }
                        const uint32_t wv = gstep ? bw[(size_t)half * 64] : 0u;
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            v[half][u] = gstep ? gfun<R>(a[u], c[u], (wv >> (e0 + u)) & 1u) : chk_lut<R>(a[u], c[u], lut);
                            *at(64 + e + u) = v[half][u];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) x5[e0 + u] = chk_lut<R>(v[0][u], v[1][u], lut);
                }
                sync();
                live = below;
            } else if (live && n > 7 && td == 5) {   // level 5 from level 6
                const uint32_t wv = gstep ? blw[(size_t)1 * 64 + lane] : 0u;   // left partial sums of level 5: word 1
#pragma unroll
                for (int e0 = 0; e0 < 32; e0 += 8) {
                    R a[8], c[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        a[u] = ld_bypass(at(64 + e0 + u));
                        c[u] = ld_bypass(at(96 + e0 + u));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        x5[e0 + u] = gstep ? gfun<R>(a[u], c[u], (wv >> (e0 + u)) & 1u) : chk_lut<R>(a[u], c[u], lut);
                }
            }
            // ---- the 32-leaf block: level 5 in x5, levels 4..0 in registers ----
            uint32_t beta = 0, dec = 0;
            const uint32_t fz = (uint32_t)__builtin_amdgcn_readlane((int)fwv, b & 63);
            if (live && fz != 0xFFFFFFFFu) {
                ScLanes<R> S{lut, fz, 0u};
                uint32_t bl = 0, br = 0;
                if (n == 5) {          // N = 32: level 5 IS the channel level, read in place like the level n-1 steps above
                    chan16(0, x5);
                    chan16(16, x5 + 16);
                } else if (n <= 7) {   // N = 64, 128: the generic steps above left level 5 in the scratch
#pragma unroll
                    for (int u = 0; u < 32; ++u) x5[u] = ld_bypass(at(32 + u));
                }
                if ((fz & 0xFFFFu) != 0xFFFFu) {
                    R l[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) l[e] = chk_lut<R>(x5[e], x5[16 + e], lut);
                    bl = S.template rec<4, 0>(l);
                }
                if ((fz >> 16) != 0xFFFFu) {
                    R r[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) r[e] = gfun<R>(x5[e], x5[16 + e], (bl >> e) & 1u);
                    br = S.template rec<4, 16>(r);
                }
                beta = (bl ^ br) | (br << 16);
                dec = S.dec;
            }
            if (frame0 + lane < P.B) P.out_bits[(size_t)(frame0 + lane) * NW + b] = dec;   // frozen leaves stay 0
            // ---- partial sums upwards (updateBit, SC_128.c:368-392): words [w][lane] in LDS ----
            int t = 5;
            curw[lane] = beta;
            while (t < n - 1 && ((b >> (t - 5)) & 1)) {
                const int nw = 1 << (t - 5);
                for (int w = 0; w < nw; ++w) {
                    const uint32_t c = curw[w * 64 + lane];
                    const uint32_t l = blw[(nw + w) * 64 + lane];
                    curw[w * 64 + lane] = l ^ c;
                    curw[(w + nw) * 64 + lane] = c;
                }
                ++t;
            }
            if (t < n && !((b >> (t - 5)) & 1)) {
                const int nw = 1 << (t - 5);
                for (int w = 0; w < nw; ++w) blw[(nw + w) * 64 + lane] = curw[w * 64 + lane];
            }
        }
        if (lane == 0 || true) {
            if (P.pm && frame0 + lane < P.B) P.pm[frame0 + lane] = 0.0;
            if (P.flags && frame0 + lane < P.B) P.flags[frame0 + lane] = 0u;
        }
    }
}

}  // namespace polar
