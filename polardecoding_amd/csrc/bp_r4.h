// bp_r4.h -- flooding BP for N = 1024 (reference: BP, BP_1024.c:372-427), register-blocked.
//
// k_bp (bp_kernel.h) keeps every message row in LDS: 152 KB in f64, one codeword per CU, eight waves of dependent LDS
// round trips (VALU 47 % / LDS pipe 46 % busy).  Here the ten stages are taken two at a time ("radix 4"): a thread owns the
// four elements e + {0, s, 2s, 3s}, s = 4^g, that are closed under stages 2g and 2g+1, so the row between the two stages
// of a group -- R[2g+1] in the right-going sweep, L[2g+1] in the left-going one -- is produced and consumed by the same
// thread and lives in its registers for the whole decode.  Only the rows that cross a group boundary (R[2], R[4], R[6],
// R[8], L[2], L[4], L[6], L[8]) go through LDS: 64 KB per codeword in f64 instead of 152 KB -> two codewords per CU, each
// thread with two independent butterflies in flight per stage, 2.7 message accesses per butterfly instead of 6, and no
// per-stage index arithmetic (the element sets are compile-time functions of the group).  The four L rows are read from
// LDS once per iteration and kept in the reader's registers for the next right-going sweep, so they share ONE transfer
// row: 40 KB per codeword, three f64 codewords per CU (LCACHE below).
//
// Threads: 256 per codeword, t = 64 w + x.  Group g < 4 removes element bits 2g, 2g+1, which are below bit 8: wave w works
// on elements [256 w, 256 w + 256) in all of them, so groups 0..3 hand rows over inside one wave (LDS executes a wave's
// operations in order; a wave-level fence is enough).  Group 4 (stages 8, 9) takes elements x' + 256 k: the only
// workgroup barriers are around it, two per iteration.
//
// Same operations on the same operands in the same order as the reference, stage by stage (operand order inside CHK and
// inside the sums kept); r[n] and, except in the last iteration, l[0] are not computed (nobody reads them).
#pragma once
#include "polar_lut.h"
#include "bp_kernel.h"

namespace polar {

// CHK form: 0 = chk_idx (prefix popcount + one conflict-free threshold read per look-up, 24 bytes of LDS per CHK),
// 1 = chk_lut (50-cell table, 40 bytes per CHK from scattered cells), 2 = chk_lut1 (one round trip, 56 bytes),
// 3 = chk_cnt (staircase counted with 14 subtractions, 8 bytes), 4 = chk_tab (the prefix count of form 0 read from a 26-byte
// table: three VALU instructions less per look-up, one more LDS read in the dependent chain -- 0.92 M against 1.19 M: slower).  With the messages out of LDS the table reads of form 1
// kept the LDS pipe 68 % busy at 42 % VALU (0.90 M frames/s f64); form 3 is VALU-bound by its f64 subtractions (0.85 M).
// Measured (2^16 frames, 50 iterations): f64 form 0 1.04 M frames/s, form 1 0.89 M; f32 form 0 1.57 M, form 1 1.81 M
// (a float entry of the 50-cell table is half the bytes) -> -1 picks form 0 for double and form 1 for float (round 2's choice).
// Round 3, after the L rows moved into registers (LCACHE: three codewords per CU, 39 % LDS busy): form 2, the one-round-trip
// table form, is ahead in both types -- f64 1.206 -> 1.303 M frames/s, f32 1.81 -> 2.09 M (same box, 2^16 frames, 50
// iterations, decisions hashed equal; profiles/r03_ab_experiments.txt run 22): five VALU instructions less per CHK than
// form 0, and the LDS pipe now has the room for its 48 bytes per CHK.  Default: 2.
#ifndef POLAR_BPR4_CHK
#define POLAR_BPR4_CHK 2
#endif
#ifndef POLAR_BPR4_ABS_LDS
#define POLAR_BPR4_ABS_LDS 0
#endif

template <typename R>
struct BpR4Cfg {
    static constexpr int N = 1024, n = 10, NW = 32, THREADS = 256;
#ifndef POLAR_BPR4_LCACHE
#define POLAR_BPR4_LCACHE 1
#endif
    // LCACHE: the rows L[2], L[4], L[6], L[8] are read from LDS once per iteration (by the left-going sweep), kept in the
    // reader's registers for the right-going sweep of the next iteration, and so only need ONE transfer row in LDS:
    // 40 KB per codeword instead of 64 KB -> three f64 codewords per CU, no LDS initialisation per frame.
    static constexpr bool LCACHE = POLAR_BPR4_LCACHE != 0;
    static constexpr int LROWS = LCACHE ? 1 : 4;
    static constexpr int MIN_BLOCKS = LCACHE ? 3 : 2;
    static constexpr size_t rows_bytes = sizeof(R) * (size_t)N * (4 + LROWS);   // R2 R4 R6 R8, then L2 L4 L6 L8 (or the one transfer row)
    static constexpr size_t off_lut = rows_bytes;
    static constexpr size_t off_dn = off_lut + ((Lut<R>::bytes + 15) / 16) * 16;
    static constexpr size_t off_st = off_dn + sizeof(R) * 64;
    static constexpr size_t lds_bytes = off_st + Stair<R>::bytes;
};

template <typename R, typename IN>
struct BpR4 {
    using C = BpR4Cfg<R>;
    static constexpr int N = C::N;
    R Ri[5][4], Li[5][4];   // interior rows R[2g+1], L[2g+1] at this thread's elements of group g
    R Lb[4][4];             // LCACHE: L[2g+2] at this thread's elements of group g, as read by the last left-going sweep
    R ch[4];                // channel LLRs at the group-4 elements
    R *rowR, *rowL;         // LDS: rowR + (g-1) N = R[2g], rowL + (g-1) N = L[2g], g = 1..4
    Lut<R> lut;
    const R *dn;            // LDS: 8x8 differences of the staircase levels, by (count, count)
    Stair<R> st;
    int t;                  // thread index 0..255
    uint32_t fz;            // frozen flags of elements 4t .. 4t+3 (group 0)

    __device__ __forceinline__ R chk2(R a, R b) const
    {
#if POLAR_BPR4_CHK == -1
        if constexpr (sizeof(R) == 8) return chk_idx<R>(a, b, st);
        else return chk_lut<R>(a, b, lut);
#elif POLAR_BPR4_CHK == 2
        return chk_lut1<R>(a, b, lut);
#elif POLAR_BPR4_CHK == 1
        return chk_lut<R>(a, b, lut);
#elif POLAR_BPR4_CHK == 3
        return chk_cnt<R>(a, b, dn);
#elif POLAR_BPR4_CHK == 4
        return chk_tab<R>(a, b, st);
#else
        return chk_idx<R>(a, b, st);
#endif
    }
    // first element of this thread in group G: t with two zero bits inserted at position 2G
    template <int G>
    __device__ __forceinline__ int e0() const
    {
        constexpr int sh = 2 * G;
        return ((t >> sh) << (sh + 2)) | (t & ((1 << sh) - 1));
    }
    __device__ __forceinline__ R prior(int k) const { return ((fz >> k) & 1u) ? R(999) : R(0); }   // BP_1024.c:386-391

    // One butterfly of the right-going sweep (BP_1024.c:395-404): inputs r[i] and l[i+1] at (upper, lower) -> r[i+1]
    __device__ __forceinline__ void bfR(R r0, R r1, R l0, R l1, R &o0, R &o1) const
    {
        o0 = chk2(r0, l1 + r1);
        o1 = r1 + chk2(r0, l0);
    }
    // One butterfly of the left-going sweep (BP_1024.c:406-415): inputs l[i+1] and r[i] -> l[i]
    __device__ __forceinline__ void bfL(R l0, R l1, R r0, R r1, R &o0, R &o1) const
    {
        o0 = chk2(l0, l1 + r1);
        o1 = l1 + chk2(r0, l0);
    }

    // ---- right-going sweep, group G: stages 2G (pairs k, k+1) and 2G+1 (pairs k, k+2) ----
    template <int G>
    __device__ __forceinline__ void sweepR()
    {
        constexpr int s = 1 << (2 * G);
        const int e = e0<G>();
        R rin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rin[k] = (G == 0) ? prior(k) : rowR[(G - 1) * N + e + k * s];
        // stage 2G: r[2G], l[2G+1] -> r[2G+1]
        bfR(rin[0], rin[1], Li[G][0], Li[G][1], Ri[G][0], Ri[G][1]);
        bfR(rin[2], rin[3], Li[G][2], Li[G][3], Ri[G][2], Ri[G][3]);
        if constexpr (G < 4) {
            // stage 2G+1: r[2G+1], l[2G+2] -> r[2G+2]
            R lin[4], o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) lin[k] = C::LCACHE ? Lb[G][k] : rowL[G * N + e + k * s];
            bfR(Ri[G][0], Ri[G][2], lin[0], lin[2], o[0], o[2]);
            bfR(Ri[G][1], Ri[G][3], lin[1], lin[3], o[1], o[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) rowR[G * N + e + k * s] = o[k];
        }
        // G == 4: stage 9 would only produce r[10], which nobody reads
    }

    // ---- left-going sweep, group G: stages 2G+1 then 2G ----
    template <int G, bool LAST>
    __device__ __forceinline__ void sweepL(uint32_t &bits)
    {
        constexpr int s = 1 << (2 * G);
        const int e = e0<G>();
        R lin[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) lin[k] = (G == 4) ? ch[k] : rowL[(C::LCACHE ? 0 : G * N) + e + k * s];
        if constexpr (C::LCACHE && G < 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) Lb[G][k] = lin[k];
        }
        // stage 2G+1: l[2G+2], r[2G+1] -> l[2G+1]
        bfL(lin[0], lin[2], Ri[G][0], Ri[G][2], Li[G][0], Li[G][2]);
        bfL(lin[1], lin[3], Ri[G][1], Ri[G][3], Li[G][1], Li[G][3]);
        if constexpr (G > 0) {
            // stage 2G: l[2G+1], r[2G] -> l[2G]
            R rin[4], o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) rin[k] = rowR[(G - 1) * N + e + k * s];
            bfL(Li[G][0], Li[G][1], rin[0], rin[1], o[0], o[1]);
            bfL(Li[G][2], Li[G][3], rin[2], rin[3], o[2], o[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) rowL[(C::LCACHE ? 0 : (G - 1) * N) + e + k * s] = o[k];
        } else if constexpr (LAST) {
            // stage 0, needed only for the decision (BP_1024.c:417-425): frozen -> 0, else (l[0] + r[0] >= 0) -> 0
            R o[4];
            bfL(Li[0][0], Li[0][1], prior(0), prior(1), o[0], o[1]);
            bfL(Li[0][2], Li[0][3], prior(2), prior(3), o[2], o[3]);
            bits = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!((fz >> k) & 1u) && !(o[k] + prior(k) >= R(0))) bits |= 1u << k;
        }
    }
};

__device__ __forceinline__ void bp_wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

template <typename R, typename IN>
__global__ __launch_bounds__(256, (BpR4Cfg<R>::MIN_BLOCKS)) void k_bp_r4(BpParams P)
{
    using C = BpR4Cfg<R>;
    constexpr int N = C::N, NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    BpR4<R, IN> s;
    s.t = threadIdx.x;
    s.rowR = reinterpret_cast<R *>(smem);
    s.rowL = s.rowR + 4 * N;
    Lut<R>::build(smem + C::off_lut, threadIdx.x, blockDim.x);
    s.lut.bind(smem + C::off_lut);
    build_delta_by_count<R>(reinterpret_cast<R *>(smem + C::off_dn), threadIdx.x, blockDim.x);
    s.dn = reinterpret_cast<const R *>(smem + C::off_dn);
    Stair<R>::build(smem + C::off_st, threadIdx.x, blockDim.x);
#if POLAR_BPR4_ABS_LDS
    {   // The kernel has no static LDS, so its dynamic LDS starts at address 0 (checked by the launcher): binding the staircase
        // tables by absolute address lets every table read take its offset as an immediate instead of an address add.
        typedef __attribute__((address_space(3))) unsigned char lds_u8;
        s.st.bind((const unsigned char *)(lds_u8 *)(uintptr_t)C::off_st);
    }
#else
    s.st.bind(smem + C::off_st);
#endif
    {
        const int e = 4 * s.t;
        s.fz = (P.frozen[e >> 5] >> (e & 31)) & 0xFu;
    }
    __syncthreads();

    __shared__ int job_slot;
    for (int frame = blockIdx.x; frame < P.B; frame = next_job_block(P.queue, frame, (int)gridDim.x, P.B, &job_slot)) {
        const IN *src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // group-4 elements t + 256 k: coalesced
            double v = (double)src[s.t + 256 * k];
            if (P.sigma > 0) v = llr_from_y(v, P.sigma);
            s.ch[k] = (R)v;   // BP_1024.c:381-382
        }
#pragma unroll
        for (int g = 0; g < 5; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s.Ri[g][k] = R(0);   // BP_1024.c:384-386
                s.Li[g][k] = R(0);   // :378-380
                if (g < 4) s.Lb[g][k] = R(0);
            }
        if constexpr (!C::LCACHE) {   // the L rows in LDS are read by the first right-going sweep before anything wrote them
            for (int i = s.t; i < 8 * N; i += 256) s.rowR[i] = R(0);
            __syncthreads();
        }

        uint32_t bits = 0;
        for (int it = 0; it < P.iters; ++it) {
            s.template sweepR<0>(); bp_wave_fence();
            s.template sweepR<1>(); bp_wave_fence();
            s.template sweepR<2>(); bp_wave_fence();
            s.template sweepR<3>();
            __syncthreads();
            s.template sweepR<4>();
            s.template sweepL<4, false>(bits);
            __syncthreads();
            s.template sweepL<3, false>(bits); bp_wave_fence();
            s.template sweepL<2, false>(bits); bp_wave_fence();
            s.template sweepL<1, false>(bits); bp_wave_fence();
            if (it + 1 == P.iters) s.template sweepL<0, true>(bits);
            else s.template sweepL<0, false>(bits);
            bp_wave_fence();
        }
        // decisions of elements 4t .. 4t+3: eight lanes make one output word
        uint32_t w = bits << (4 * (s.t & 7));
        w |= (uint32_t)__shfl_xor((int)w, 1);
        w |= (uint32_t)__shfl_xor((int)w, 2);
        w |= (uint32_t)__shfl_xor((int)w, 4);
        if ((s.t & 7) == 0) P.out_bits[(size_t)frame * NW + (s.t >> 3)] = w;
        __syncthreads();   // the LDS rows are cleared for the next frame
    }
}

}  // namespace polar
