// scl_fast.h -- tuned SCL / CA-SCL kernel for list size 8 (S = 8 lanes per path), N = 2^NLOG.
//
// One codeword per wavefront, 4 independent wavefronts per workgroup (no inter-wave barrier in
// the decode loop).  Lane = (path p = lane/8, position pos = lane%8).  What the measurements in
// profiles/r01_ubench_gfx950.txt dictate ("VALU budget"):
//   * f64 VALU ops cost 2 issue slots, and 4x that when <= 8 lanes are enabled  -> nothing in the
//     decode loop is EXEC-masked: every lane always executes, idle lanes carry don't-care data;
//   * every v_cmp / v_cndmask / v_addc costs 2 slots, so the reference's 8-level staircase
//     (7 compares per look-up, two look-ups per CHK) is replaced by ONE exact compare per look-up:
//     the operand's exponent + 3 mantissa bits select a cell of a 50-entry LDS table holding the
//     only threshold that can lie in that cell and the number of thresholds below it;
//     T(s) - T(d) comes from an 8x8 LDS table of the IEEE differences (bit-exact by construction);
//   * LDS capacity, not bandwidth, limits waves/CU -> all per-path LLR levels live in REGISTERS:
//       level t >= 4 : 2^t/8 registers per lane, element e = pos + 8 r        (A[2^t/8 + r])
//       level t <= 3 : one register per lane (a3, a2, a1), element e = pos
//     only the channel vector, the single-path left child of the root (computed before the first
//     fork) and the bit-packed partial sums of levels >= 5 are in LDS (~14 KB per codeword);
//   * the reference's copyPath/simpleCopy (SCL_1024.c:451-478) is replaced by lazy pointers for
//     levels >= 4 (a g-step reads its source level from the owning path's lanes with
//     ds_bpermute; f-steps always read the lane's own registers) and by an eager 3-register
//     shuffle for levels <= 3;
//   * while fewer than L paths exist (phase 1, SCL_1024.c:581-605) the idle lane groups run as
//     replicas (group q mirrors slot q mod act), so a fork needs no copy at all;
//   * u_hat is not stored per path: x_hat = root partial sums of the winner, u_hat = x_hat F^{(x)n};
//   * every step body exists once in the code (uniform branches on the leaf index), so the kernel
//     stays inside the instruction cache.
// Arithmetic order is the reference's (polar_math.h); results are bit-identical to k_scl_generic.
#pragma once
#include "polar_math.h"
#include "polar_lut.h"
#include "scl_generic.h"

namespace polar {

// ---- cross-lane helpers -----------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int K>  // lane i <- lane i+K within its row of 16 (row_shl), every lane valid
__device__ __forceinline__ double shl_true(double x)
{
    long long b = __double_as_longlong(x);
    int lo = dpp_i<0x100 + K>((int)b), hi = dpp_i<0x100 + K>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int K>
__device__ __forceinline__ float shl_true(float x)
{
    return __int_as_float(dpp_i<0x100 + K>(__float_as_int(x)));
}
#ifdef POLAR_SWIZZLE_PARTNER
// lane i <- lane i^K through the LDS crossbar (no VALU slot); equals lane i+K for the low lanes that are used
template <int K>
__device__ __forceinline__ double shl_lanes(double x)
{
    long long b = __double_as_longlong(x);
    int lo = __builtin_amdgcn_ds_swizzle((int)b, (K << 10) | 0x1F), hi = __builtin_amdgcn_ds_swizzle((int)(b >> 32), (K << 10) | 0x1F);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int K>
__device__ __forceinline__ float shl_lanes(float x)
{
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (K << 10) | 0x1F));
}
#define POLAR_SHL_DEFINED
#endif
#ifndef POLAR_SHL_DEFINED
template <int K>  // lane i <- lane i+K within its row of 16 (row_shl)
__device__ __forceinline__ double shl_lanes(double x)
{
    long long b = __double_as_longlong(x);
    int lo = dpp_i<0x100 + K>((int)b), hi = dpp_i<0x100 + K>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int K>
__device__ __forceinline__ float shl_lanes(float x)
{
    return __int_as_float(dpp_i<0x100 + K>(__float_as_int(x)));
}
#endif
template <int K>  // lane i <- lane i-K within its row (row_shr)
__device__ __forceinline__ double shr_lanes(double x)
{
    long long b = __double_as_longlong(x);
    int lo = dpp_i<0x110 + K>((int)b), hi = dpp_i<0x110 + K>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int K>
__device__ __forceinline__ float shr_lanes(float x)
{
    return __int_as_float(dpp_i<0x110 + K>(__float_as_int(x)));
}

// lane i <- lane i^4 (inside an 8-lane path group): row_shl:4 into the low quads, row_shr:4 into the high quads
__device__ __forceinline__ int xor4_i(int v)
{
    int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
}
template <int QP>  // quad_perm
__device__ __forceinline__ int quad_i(int v) { return __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, true); }
__device__ __forceinline__ double xor_lanes4(double x)
{
    long long b = __double_as_longlong(x);
    int lo = xor4_i((int)b), hi = xor4_i((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ float xor_lanes4(float x) { return __int_as_float(xor4_i(__float_as_int(x))); }
template <int QP>
__device__ __forceinline__ double quad_lanes(double x)
{
    long long b = __double_as_longlong(x);
    int lo = quad_i<QP>((int)b), hi = quad_i<QP>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int QP>
__device__ __forceinline__ float quad_lanes(float x) { return __int_as_float(quad_i<QP>(__float_as_int(x))); }

#ifdef POLAR_MARKS  // static instruction accounting (tools/count_marks.py): comments in the ISA
#define POLAR_MARK(name) __asm__ volatile("; MARK " name)
#else
#define POLAR_MARK(name) do { } while (0)
#endif

__device__ __forceinline__ void lds_fence() { __asm__ volatile("" ::: "memory"); }

// negate x when bit 31 of `m` is set (m & 0x80000000 pre-masked): g = cL + (bit ? -cU : cU)
__device__ __forceinline__ double flip(double x, uint32_t m) { return __hiloint2double(__double2hiint(x) ^ (int)m, __double2loint(x)); }
__device__ __forceinline__ float flip(float x, uint32_t m) { return __int_as_float(__float_as_int(x) ^ (int)m); }
// lower-node update with the partner bit at bit position `sh` of w (SCL_1024.c:412-416): cL +- cU
// flip_bit / g_bit: polar_math.h

template <typename R, int NLOG>
struct FastCfg {
    static constexpr int N = 1 << NLOG;
    static constexpr int NW = N / 32;
    static constexpr int TOP = NLOG - 1;      // virtual level (never stored per path)
    static constexpr int HI = NLOG - 2;       // highest stored level
    // BIG (N = 1024): levels 7 and 8, the top-left level and the channel vector stay OUT of registers and LDS
    // (per-wave scratch in global memory, L2/MALL resident; the channel LLRs are re-read from the input), so
    // that four waves per SIMD fit.  Registers then hold levels 4..6 only.
    static constexpr bool BIG = NLOG >= 10;
    static constexpr int RHI = BIG ? 6 : HI;  // highest REGISTER level
    static constexpr int NA = (1 << RHI) / 4; // registers for levels 4..RHI (level t at offset 2^t/8)
    static constexpr int WAVES = 4;
    static constexpr int MIN_WAVES_PER_SIMD = BIG ? 4 : 4;
    // per-wave global scratch (elements of R): l8 [8][256], l7 [8][128], l6 staging [8][64], top-left [512]
    static constexpr size_t sc_l8 = 0;
    static constexpr size_t sc_l7 = sc_l8 + 8 * 256;
    static constexpr size_t sc_l6 = sc_l7 + 8 * 128;
    static constexpr size_t sc_tl = sc_l6 + 8 * 64;
    static constexpr size_t scratch_elems = BIG ? sc_tl + 512 : 0;
    // block-shared LDS
    static constexpr size_t off_lut = 0;
    static constexpr size_t off_frz = off_lut + ((Lut<R>::bytes + 15) / 16) * 16;   // frozen words [NW]
    static constexpr size_t off_crc = off_frz + 4 * NW;                             // crc table [N]
    static constexpr size_t shared_bytes = off_crc + 4 * N;
    // per-wave LDS
    static constexpr size_t off_ch = 0;
    static constexpr size_t off_tl = off_ch + (BIG ? 0 : sizeof(R) * N);        // (small N) channel LLRs
    static constexpr size_t off_bl = off_tl + (BIG ? 0 : sizeof(R) * (N / 2));  // (small N) top-left level
    static constexpr size_t off_cw = off_bl + 4 * 8 * NW;           // saved partial sums [8][NW], then working [8][NW]
    static constexpr size_t off_cd = off_cw + 4 * 8 * NW;           // candidates [16] (exact fall-back)
    static constexpr size_t off_ky = off_cd + sizeof(R) * 16;       // candidate keys [16] u32
    static constexpr size_t per_wave = off_ky + 64;
    static constexpr size_t total = shared_bytes + WAVES * per_wave;
};

// scratch traffic: plain stores (write-through to L2), loads that bypass the per-CU L1 (relaxed agent-scope
// atomic load = global_load ... sc1) so that a line cached from an earlier frame can never be returned
__device__ __forceinline__ double ld_sc(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void vm_drain() { __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 32-bit ordering key of a non-negative metric: f32 bits are exact, the high word of an f64 is exact
// unless two candidates share it across the survivor boundary (then the exact fall-back runs)
__device__ __forceinline__ uint32_t metric_key(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t metric_key(float x) { return (uint32_t)__float_as_int(x); }

template <typename R, typename IN, int NLOG, bool CRC_ON>
struct FastDec {
    using C = FastCfg<R, NLOG>;
    static constexpr int N = C::N, NW = C::NW, TOP = C::TOP, HI = C::HI, RHI = C::RHI, L = 8;
    static constexpr bool BIG = C::BIG;
    static constexpr int NFA = HI - 3;  // pointer fields for LLR levels 4..HI; partial-sum levels 5..TOP follow

    // ---- per-lane state ----
    R A[C::NA];      // levels 4..RHI
    R a3, a2, a1;    // levels 3..1 (element pos)
    R PM;            // valid at pos 0
    uint32_t ptr;    // 3 bits per field: LLR level t -> field t-4; partial sums of level t -> field NFA + t-5
    uint32_t crc, bl0;
    uint32_t fl;
    int logact;      // log2(number of distinct paths); groups are replicas while < 3
    int p, pos, lane;
    int own_addr, oth_addr;  // byte addresses into keys[] for the rank network
    Lut<R> lut;
    // LDS
    R *ch, *tl, *cand;
    uint32_t *blw, *curw, *keys;
    const uint32_t *crct;
    // global
    R *scr;          // per-wave scratch (BIG)
    const IN *src;   // this frame's input row
    double sigma;

    // channel LLR straight from the input (SCL_1024.c:574-578)
    __device__ __forceinline__ R chv(int e) const
    {
        double v = (double)src[e];
        if (sigma > 0) v = llr_from_y(v, sigma);
        return (R)v;
    }

    __device__ __forceinline__ int pa(int t) const { return (ptr >> (3 * (t - 4))) & 7; }
    __device__ __forceinline__ void set_pa(int t, int v) { ptr = (ptr & ~(7u << (3 * (t - 4)))) | ((uint32_t)v << (3 * (t - 4))); }
    __device__ __forceinline__ int pb(int t) const { return (ptr >> (3 * (NFA + t - 5))) & 7; }
    __device__ __forceinline__ void set_pb(int t, int v) { ptr = (ptr & ~(7u << (3 * (NFA + t - 5)))) | ((uint32_t)v << (3 * (NFA + t - 5))); }

    __device__ __forceinline__ R chk(R a, R b) const { return chk_lut<R>(a, b, lut); }

    // ---- f steps on own registers ----
    template <int T>  // T in [4, HI-1]: level T from level T+1
    __device__ __forceinline__ void f_up()
    {
        constexpr int RO = (1 << T) / 8;
#pragma unroll
        for (int r = 0; r < RO; ++r) A[RO + r] = chk(A[2 * RO + r], A[3 * RO + r]);
        set_pa(T, p);
    }

    // ---- g steps: level T from level T+1 of the owning path (bpermute), partial sums beta_T ----
    template <int T>  // T in [4, HI-1]
    __device__ __forceinline__ void g_up()
    {
        constexpr int RO = (1 << T) / 8;
        const int sl = pa(T + 1) * 8 + pos;
        if constexpr (T >= 5) {
            const uint32_t *bw = blw + pb(T) * NW + (1 << (T - 5));
#pragma unroll
            for (int r4 = 0; r4 < RO / 4; ++r4) {
                const uint32_t w = bw[r4] >> pos;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = r4 * 4 + k;
                    const R x = __shfl(A[2 * RO + r], sl), y = __shfl(A[3 * RO + r], sl);
                    A[RO + r] = g_bit<R>(x, y, w, 8 * k);
                }
            }
        } else {  // T == 4: bits 16 + e in bl0
            const uint32_t w = bl0 >> (16 + pos);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const R x = __shfl(A[4 + r], sl), y = __shfl(A[6 + r], sl);
                A[2 + r] = g_bit<R>(x, y, w, 8 * r);
            }
        }
        set_pa(T, p);
    }

    // ======================= small N: levels up to HI in registers, ch / top-left in LDS =======================
    // Level HI from the top level.  right: j >= N/2 (virtual top level from ch and beta_TOP), else the stored
    // top-left array.  gstep: g (with beta_HI) instead of f.  The level-HI registers are rotated so that the
    // body exists once in the code.
    __device__ __forceinline__ void top_to_hi(bool right, bool gstep)
    {
        constexpr int RO = (1 << HI) / 8;  // registers of level HI
        constexpr int H = 1 << HI;         // pair distance in elements
        const uint32_t *bt = blw + pb(TOP) * NW + (1 << (TOP - 5));
        const uint32_t *bh = blw + pb(HI) * NW + (1 << (HI - 5));
        constexpr int CH = RO < 8 ? RO : 8;  // registers produced per pass
        for (int q = 0; q < RO / CH; ++q) {
            R out[CH];
#pragma unroll
            for (int h4 = 0; h4 < CH / 4; ++h4) {
                const int r4 = q * (CH / 4) + h4;
                const uint32_t wt0 = bt[r4] >> pos, wt1 = bt[H / 32 + r4] >> pos, wh = bh[r4] >> pos;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = pos + 8 * (r4 * 4 + k);
                    R x, y;
                    if (right) {
                        x = g_bit<R>(ch[e], ch[e + N / 2], wt0, 8 * k);
                        y = g_bit<R>(ch[e + H], ch[e + H + N / 2], wt1, 8 * k);
                    } else {
                        x = tl[e];
                        y = tl[e + H];
                    }
                    out[h4 * 4 + k] = gstep ? g_bit<R>(x, y, wh, 8 * k) : chk(x, y);
                }
            }
#pragma unroll
            for (int r = 0; r < RO - CH; ++r) A[RO + r] = A[RO + r + CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) A[2 * RO - CH + k] = out[k];
        }
        set_pa(HI, p);
    }

    // ======================= N = 1024: levels 8, 7 and the top-left level in the per-wave scratch =======================
    __device__ __forceinline__ R *l8(int slot) const { return scr + C::sc_l8 + slot * 256; }
    __device__ __forceinline__ R *l7(int slot) const { return scr + C::sc_l7 + slot * 128; }
    __device__ __forceinline__ R *l6s(int slot) const { return scr + C::sc_l6 + slot * 64; }
    __device__ __forceinline__ R *tls() const { return scr + C::sc_tl; }

    __device__ __forceinline__ void load_l6()
    {
        vm_drain();
        const R *q = l6s(p) + pos;
#pragma unroll
        for (int r = 0; r < 8; ++r) A[8 + r] = ld_sc(q + 8 * r);
        set_pa(6, p);
    }
    // Octet heads with d >= 8 (j = 0, 256, 512, 768): level 8 from the top level (f or g), then f to 7 and 6.
    // Per pass rr the lane produces its level-8 elements e = pos + 8 rr + 64 k (k < 4), the two level-7 and the
    // one level-6 element below them, so no value is re-read before it is complete.
    __device__ __forceinline__ void from_top(bool right, bool gstep)
    {
        vm_drain();
        const uint32_t *bt = blw + pb(TOP) * NW + (1 << (TOP - 5));  // beta_9: 16 words
        const uint32_t *bh = blw + pb(HI) * NW + (1 << (HI - 5));    // beta_8: 8 words
        const R *t = tls();
        R *o8 = l8(p), *o7 = l7(p), *o6 = l6s(p);
        for (int rr = 0; rr < 8; ++rr) {
            const int e0 = pos + 8 * rr;
            const int sh = 8 * (rr & 3);
            const int wq = rr >> 2;
            R v8[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + 64 * k;
                R x, y;
                if (right) {
                    const uint32_t w0 = bt[2 * k + wq] >> pos, w1 = bt[8 + 2 * k + wq] >> pos;
                    x = g_bit<R>(chv(e), chv(e + 512), w0, sh);
                    y = g_bit<R>(chv(e + 256), chv(e + 768), w1, sh);
                } else {
                    x = ld_sc(t + e);
                    y = ld_sc(t + e + 256);
                }
                if (gstep) v8[k] = g_bit<R>(x, y, bh[2 * k + wq] >> pos, sh);
                else v8[k] = chk(x, y);
                o8[e] = v8[k];
            }
            const R v70 = chk(v8[0], v8[2]), v71 = chk(v8[1], v8[3]);
            o7[e0] = v70;
            o7[e0 + 64] = v71;
            o6[e0] = chk(v70, v71);
        }
        set_pa(8, p);
        set_pa(7, p);
        load_l6();
    }
    // d == 7 (j = 128 * odd): g to level 7 from the owner's level 8, then f to level 6
    __device__ __forceinline__ void from_l8()
    {
        vm_drain();
        const R *s8 = l8(pa(8));
        const uint32_t *b7 = blw + pb(7) * NW + (1 << (7 - 5));  // beta_7: 4 words
        R *o7 = l7(p), *o6 = l6s(p);
        for (int rr = 0; rr < 8; ++rr) {
            const int e0 = pos + 8 * rr;
            const int sh = 8 * (rr & 3);
            const int wq = rr >> 2;
            const R a0 = ld_sc(s8 + e0), a1_ = ld_sc(s8 + e0 + 64), a2_ = ld_sc(s8 + e0 + 128), a3_ = ld_sc(s8 + e0 + 192);
            const R v70 = g_bit<R>(a0, a2_, b7[wq] >> pos, sh);
            const R v71 = g_bit<R>(a1_, a3_, b7[2 + wq] >> pos, sh);
            o7[e0] = v70;
            o7[e0 + 64] = v71;
            o6[e0] = chk(v70, v71);
        }
        set_pa(7, p);
        load_l6();
    }
    // d == 6 (j = 64 * odd): g to level 6 (registers) from the owner's level 7
    __device__ __forceinline__ void from_l7()
    {
        vm_drain();
        const R *s7 = l7(pa(7)) + pos;
        const uint32_t *b6 = blw + pb(6) * NW + (1 << (6 - 5));  // beta_6: 2 words
        const uint32_t w0 = b6[0] >> pos, w1 = b6[1] >> pos;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const R x = ld_sc(s7 + 8 * r), y = ld_sc(s7 + 8 * r + 64);
            A[8 + r] = g_bit<R>(x, y, r < 4 ? w0 : w1, 8 * (r & 3));
        }
        set_pa(6, p);
    }

    template <int T>
    __device__ __forceinline__ void f_chain_up(int d)
    {
        if constexpr (T >= 4) {
            if (d > T) f_up<T>();
            f_chain_up<T - 1>(d);
        }
    }
    template <int T>
    __device__ __forceinline__ void g_select(int d)
    {
        if constexpr (T >= 4) {
            if (d == T) g_up<T>();
            else g_select<T - 1>(d);
        }
    }
    // Head of octet o (leaves 8o .. 8o+7): g at level d = ctz(8o) (root f for o = 0), f chain down to level 3.
    // Each step body appears once; the branches are uniform.
    __device__ __forceinline__ void octet_head(int o)
    {
        const int d = (o == 0) ? NLOG : 3 + __builtin_ctz((unsigned)o);
        if constexpr (BIG) {
            if (d >= 8) from_top(/*right=*/o >= N / 16, /*gstep=*/d == 8);
            else if (d == 7) from_l8();
            else if (d == 6) from_l7();
            else if (d >= 4) g_select<5>(d);
            else {
                const int sl = pa(4) * 8 + pos;
                a3 = g_bit<R>(__shfl(A[2], sl), __shfl(A[3], sl), bl0, 8 + pos);
            }
            f_chain_up<5>(d);
        } else {
            if (d >= HI) {
                top_to_hi(/*right=*/o >= N / 16, /*gstep=*/d == HI);
            } else if (d >= 4) {
                g_select<HI - 1>(d);
            } else {
                const int sl = pa(4) * 8 + pos;
                a3 = g_bit<R>(__shfl(A[2], sl), __shfl(A[3], sl), bl0, 8 + pos);
            }
            f_chain_up<HI - 1>(d);
        }
        if (d > 3) a3 = chk(A[2], A[3]);
    }

    // ---- partial sums (updateBit, SCL_1024.c:424-448), bit positions static inside an octet ----
    template <int K>
    __device__ __forceinline__ void set_bit_k(int o, uint32_t bit)
    {
        if constexpr ((K & 1) == 0) {
            bl0 = (bl0 & ~2u) | (bit << 1);  // left child at level 0
        } else if constexpr ((K & 3) == 1) {
            const uint32_t c1 = (((bl0 >> 1) & 1u) ^ bit) | (bit << 1);
            bl0 = (bl0 & ~0xCu) | (c1 << 2);
        } else if constexpr (K == 3) {
            const uint32_t c1 = (((bl0 >> 1) & 1u) ^ bit) | (bit << 1);
            const uint32_t c2 = (((bl0 >> 2) & 3u) ^ c1) | (c1 << 2);
            bl0 = (bl0 & ~0xF0u) | (c2 << 4);
        } else {
            set_bit_tail(8 * o + 7, bit);
        }
    }
    __device__ __forceinline__ void set_bit_tail(int j, uint32_t bit)
    {
        uint32_t cur = bit;
        const int z = __builtin_ctz(~(unsigned)j);  // trailing ones of j (uniform), >= 3 here
        const int zl = z < 5 ? z : 5;
        for (int t = 0; t < zl; ++t) {
            const int h = 1 << t;
            const uint32_t mask = (1u << h) - 1u;
            const uint32_t l = (bl0 >> h) & mask;
            cur = (l ^ (cur & mask)) | ((cur & mask) << h);
        }
        if (z < 5) {
            const int h = 1 << z;
            const uint32_t mask = (1u << h) - 1u;
            bl0 = (bl0 & ~(mask << h)) | ((cur & mask) << h);
            return;
        }
        // rare path (every 32 leaves): levels >= 5 are words in LDS
        lds_fence();
        if (pos == 0) curw[p * NW] = cur;
        lds_fence();
        int t = 5;
        while (t < NLOG && ((j >> t) & 1)) {
            const int nw = 1 << (t - 5);
            const int sb = pb(t);
            for (int w = pos; w < nw; w += 8) {
                const uint32_t c = curw[p * NW + w];
                const uint32_t l = blw[sb * NW + nw + w];
                curw[p * NW + w] = l ^ c;
                curw[p * NW + w + nw] = c;
            }
            lds_fence();
            ++t;
        }
        if (t < NLOG) {
            const int nw = 1 << (t - 5);
            for (int w = pos; w < nw; w += 8) blw[p * NW + nw + w] = curw[p * NW + w];
            set_pb(t, p);
            lds_fence();
        }
    }

    // ---- survivors of the 2L candidates (SCL_1024.c:612-633) ----
    // Returns the 16-bit mask (bit p: 0-branch of slot p survives, bit 8+p: 1-branch).  Strict "< med" with
    // med the (L+1)-th smallest  <=>  #{m : c_m <= c} <= L.  Rank network: the 16 keys sit in every row of 16
    // lanes; row r compares its keys with the row rotated by 4r .. 4r+3, the four rows add up.
    __device__ __forceinline__ uint32_t survivors(R c0, R c1)
    {
        lds_fence();
        if (pos == 0) {
            keys[2 * p] = metric_key(c0);
            keys[2 * p + 1] = metric_key(c1);
        }
        lds_fence();
        const uint32_t own = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(keys) + own_addr);
        const uint32_t oth = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(keys) + oth_addr);
        uint32_t cnt = (oth <= own) ? 1u : 0u;
        cnt += ((uint32_t)dpp_i<0x121>((int)oth) <= own) ? 1u : 0u;  // row_ror:1
        cnt += ((uint32_t)dpp_i<0x122>((int)oth) <= own) ? 1u : 0u;
        cnt += ((uint32_t)dpp_i<0x123>((int)oth) <= own) ? 1u : 0u;
        {
            auto r = __builtin_amdgcn_permlane16_swap(cnt, cnt, false, false);
            cnt = r[0] + r[1];
            auto q = __builtin_amdgcn_permlane32_swap(cnt, cnt, false, false);
            cnt = q[0] + q[1];
        }
        uint32_t mask = (uint32_t)__ballot(cnt <= (uint32_t)L) & 0xFFFFu;
        if (sizeof(R) == 8 && __popc(mask) != L) {
            // keys tie across the boundary (or a true median tie): decide on the full metrics
            fl |= 0x4u;   // POLAR_FLAG_RERANK
            lds_fence();
            if (pos == 0) {
                cand[p] = c0;
                cand[8 + p] = c1;
            }
            lds_fence();
            const R mine = cand[lane & 15];
            int n = 0;
#pragma unroll
            for (int m = 0; m < 16; ++m) n += (cand[m] <= mine) ? 1 : 0;
            mask = (uint32_t)__ballot(n <= L) & 0xFFFFu;
        }
        return mask;
    }

    // true if every path keeps the branch its lambda favours (cb <= cw: its metric with that / the other branch, at pos 0)
    __device__ __forceinline__ bool trivial_prune(R cb, R cw) const
    {
        // the largest favoured key, then one compare per path: "max favoured < min other" <=> every other key is above it
        const uint32_t m0 = pos == 0 ? 0xFFFFFFFFu : 0u;
        uint32_t mx = metric_key(cb) & m0;
        mx = max(mx, (uint32_t)dpp_i<0x128>((int)mx));   // row_ror:8: the other path of this row of 16 lanes
        {
            auto a = __builtin_amdgcn_permlane16_swap(mx, mx, false, false);
            mx = max(a[0], a[1]);
        }
        {
            auto a = __builtin_amdgcn_permlane32_swap(mx, mx, false, false);
            mx = max(a[0], a[1]);
        }
        return __ballot(mx >= (metric_key(cw) | ~m0)) == 0ull;
    }

    // ---- decision at leaf j = 8o + K given lambda (valid at pos 0) ----
    template <int K>
    __device__ __forceinline__ void decide(int o, bool frozen, R lam)
    {
        const int j = 8 * o + K;
        POLAR_MARK("decide_begin");
        uint32_t crcw = 0;
        if (CRC_ON && !frozen) crcw = crct[j];
        uint32_t bit = 0;
        const R tt = lut.tabv(lam);
        if (frozen) {
            // PHI(.,0) = T + (lam < 0 ? |lam| : 0)  (SCL_1024.c:481-502); T + 0 is exact
            PM += tt + negmax(lam);  // SCL_1024.c:601-604, :662-665
        } else {
            if (logact < 3) {
                // phase 1 (SCL_1024.c:586-600): group q is slot q mod 2^logact; the fork sends the replicas
                // with bit `logact` of q set down the 1-branch
                const R ph0 = tt + negmax(lam), ph1 = tt + posmax(lam);  // PHI(.,1) = T + (lam > 0 ? |lam| : 0)
                bit = (p >> logact) & 1;
                PM += bit ? ph1 : ph0;
                ++logact;
            } else {
                // phase 2 (SCL_1024.c:610-661)
                POLAR_MARK("phase2_begin");
                // the branch lambda favours costs T(|lambda|), the other one T(|lambda|) + |lambda|: these ARE c0 / c1 in
                // the order the sign of lambda says.  If the largest favoured key is below the smallest other key, the
                // eight favoured candidates are the eight smallest of the sixteen (85 % of the information leaves at
                // 1-3 dB): every path keeps its favoured branch, no ranking, no fork (scl_fast2.h has the long version)
                const R cb = PM + tt, cw = PM + (tt + absr(lam));
                const uint32_t lneg = hi_word(lam) >> 31;
                // (measured: +4 % at N = 1024; at N = 128 it lost 1-5 % with the two-reduction form of the test and gains 9 %
                // with the one-reduction form below -- CA-SCL N = 128: 60.3 -> 65.6 M frames/s f64 -- so it is on for both)
                if (trivial_prune(cb, cw)) {
                    bit = (uint32_t)__shfl((int)lneg, p * 8);   // pos 0 holds lambda
                    PM = cb;
                } else {
                const R c0 = lneg ? cw : cb, c1 = lneg ? cb : cw;
                const uint32_t mask = survivors(c0, c1);
                POLAR_MARK("rank_end");
                const uint32_t m0 = mask & 0xFFu, m1 = mask >> 8;
                const uint32_t m_both = m0 & m1, m_dead = ~(m0 | m1) & 0xFFu;
                if (__popc(mask) < L) fl |= 0x1u;  // median tie ("Oops!", :621-622)
                const bool s0 = (m0 >> p) & 1, s1 = (m1 >> p) & 1;
                if (m_dead == 0u) {
                    bit = (!s0 && s1) ? 1u : 0u;  // every slot keeps exactly one branch: no copy
                    PM = bit ? c1 : c0;
                } else {
                    POLAR_MARK("fork_begin");
                    // m-th both-survivor (ascending slot) forks into the m-th dead slot (:636-661)
                    const bool dead = !s0 && !s1;
                    const int myrank = __popc(m_dead & ((1u << p) - 1u));
                    int sg = p;
                    bool refilled = false;
                    {
                        uint32_t bm = m_both;
                        int k = 0;
                        while (bm) {
                            const int b = __builtin_ctz(bm);
                            if (dead && k == myrank) {
                                sg = b;
                                refilled = true;
                            }
                            bm &= bm - 1;
                            ++k;
                        }
                    }
                    const int sl = sg * 8 + pos;
                    const R c1s = __shfl(c1, sl);
                    ptr = __shfl(ptr, sl);
                    crc = __shfl(crc, sl);
                    bl0 = __shfl(bl0, sl);
                    if constexpr ((K & 4) == 0) a3 = __shfl(a3, sl);  // still to be read by g2
                    if constexpr ((K & 2) == 0) a2 = __shfl(a2, sl);  // ... by g1
                    if constexpr ((K & 1) == 0) a1 = __shfl(a1, sl);  // ... by g0
                    if (refilled) { bit = 1; PM = c1s; }
                    else if (s0) { bit = 0; PM = c0; }
                    else if (s1) { bit = 1; PM = c1; }
                    else { bit = 0; PM = c0; }  // tie rule: un-refilled dead slot continues as its 0-branch
                }
                }
            }
            POLAR_MARK("fork_end");
            if (CRC_ON) crc ^= bit ? crcw : 0u;
        }
        POLAR_MARK("setbit_begin");
        set_bit_k<K>(o, bit);
        POLAR_MARK("decide_end");
    }

    // ---- octets whose first seven leaves are frozen (patterns 0xFF, 0x7F) ----
    // All partner bits inside the octet are 0, so every g is cL + cU and the three levels can be evaluated
    // breadth-first: one CHK pass per level produces the f results in the low half and the g results in the
    // high half of each node -- the same operations on the same operands as the leaf-by-leaf schedule.
    // lambda_k ends in lane pos = k.  The frozen-leaf metric updates PM += PHI(lambda_k, 0) keep leaf order.
    __device__ __forceinline__ void octet_frozen_prefix(int o, bool last_frozen)
    {
        R x = a3, y = xor_lanes4(x);
        R f = chk(x, y), g = x + y;
        x = (pos & 4) ? g : f;
        y = quad_lanes<0x4E>(x);  // quad_perm [2,3,0,1]: lane ^ 2
        f = chk(x, y); g = x + y;
        x = (pos & 2) ? g : f;
        y = quad_lanes<0xB1>(x);  // quad_perm [1,0,3,2]: lane ^ 1
        f = chk(x, y); g = x + y;
        const R lam = (pos & 1) ? g : f;
        const R tt = lut.tabv(lam);
        R ph = tt + negmax(lam);  // PHI(lambda_k, 0) in lane k
        PM += ph;
        ph = shl_true<1>(ph); PM += ph;
        ph = shl_true<1>(ph); PM += ph;
        ph = shl_true<1>(ph); PM += ph;
        ph = shl_true<1>(ph); PM += ph;
        ph = shl_true<1>(ph); PM += ph;
        ph = shl_true<1>(ph); PM += ph;
        bl0 &= ~0xFEu;  // partial sums of levels 0..2 inside this octet: all zero
        if (last_frozen) {
            ph = shl_true<1>(ph); PM += ph;
            set_bit_tail(8 * o + 7, 0u);
        } else {
            decide<7>(o, false, shl_true<7>(lam));
        }
    }

    // ---- the 8 leaves of octet o; a3 holds the level-3 LLRs ----
    __device__ __forceinline__ void octet(int o, uint32_t fm)
    {
        // leaf 0: f2 f1 f0
        a2 = chk(a3, shl_lanes<4>(a3));
        a1 = chk(a2, shl_lanes<2>(a2));
        decide<0>(o, fm & 1, chk(a1, shl_lanes<1>(a1)));
        // leaf 1: g0
        decide<1>(o, (fm >> 1) & 1, g_bit<R>(a1, shl_lanes<1>(a1), bl0, 1));
        // leaf 2: g1 f0
        a1 = g_bit<R>(a2, shl_lanes<2>(a2), bl0, 2 + pos);
        decide<2>(o, (fm >> 2) & 1, chk(a1, shl_lanes<1>(a1)));
        // leaf 3: g0
        decide<3>(o, (fm >> 3) & 1, g_bit<R>(a1, shl_lanes<1>(a1), bl0, 1));
        // leaf 4: g2 f1 f0
        a2 = g_bit<R>(a3, shl_lanes<4>(a3), bl0, 4 + pos);
        a1 = chk(a2, shl_lanes<2>(a2));
        decide<4>(o, (fm >> 4) & 1, chk(a1, shl_lanes<1>(a1)));
        // leaf 5: g0
        decide<5>(o, (fm >> 5) & 1, g_bit<R>(a1, shl_lanes<1>(a1), bl0, 1));
        // leaf 6: g1 f0
        a1 = g_bit<R>(a2, shl_lanes<2>(a2), bl0, 2 + pos);
        decide<6>(o, (fm >> 6) & 1, chk(a1, shl_lanes<1>(a1)));
        // leaf 7: g0
        decide<7>(o, (fm >> 7) & 1, g_bit<R>(a1, shl_lanes<1>(a1), bl0, 1));
    }
};

#ifdef POLAR_STAMPS  // diagnostic build: s_memtime per section, summed per wave, added to P.dbg[]
#define STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsec[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <typename R, typename IN, int NLOG, bool CRC_ON>
__global__ __launch_bounds__(256, (FastCfg<R, NLOG>::MIN_WAVES_PER_SIMD)) void k_scl_fast(SclParams P)
{
#ifdef POLAR_STAMPS
    unsigned long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    using D = FastDec<R, IN, NLOG, CRC_ON>;
    using C = FastCfg<R, NLOG>;
    constexpr int N = C::N, NW = C::NW, L = 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6;
    unsigned char *base = smem + C::shared_bytes + (size_t)wave * C::per_wave;
    uint32_t *frz = reinterpret_cast<uint32_t *>(smem + C::off_frz);
    uint32_t *crct = reinterpret_cast<uint32_t *>(smem + C::off_crc);

    // block-shared tables
    Lut<R>::build(smem + C::off_lut, threadIdx.x, blockDim.x);
    for (int i = threadIdx.x; i < NW; i += blockDim.x) frz[i] = P.frozen[i];
    if (CRC_ON)
        for (int i = threadIdx.x; i < N; i += blockDim.x) crct[i] = P.crc_tab[i];
    __syncthreads();

    D s;
    s.lane = threadIdx.x & 63;
    s.p = s.lane >> 3;
    s.pos = s.lane & 7;
    s.lut.bind(smem + C::off_lut);
    s.crct = crct;
    s.ch = reinterpret_cast<R *>(base + C::off_ch);
    s.tl = reinterpret_cast<R *>(base + C::off_tl);
    s.sigma = P.sigma;
    s.blw = reinterpret_cast<uint32_t *>(base + C::off_bl);
    s.curw = reinterpret_cast<uint32_t *>(base + C::off_cw);
    s.cand = reinterpret_cast<R *>(base + C::off_cd);
    s.keys = reinterpret_cast<uint32_t *>(base + C::off_ky);
    {   // rank network addressing: candidate i = lane & 15 is (slot i & 7, branch i >> 3), stored at keys[2*slot + branch]
        const int i = s.lane & 15, r = s.lane >> 4;
        const int jj = (i + 4 * r) & 15;
        s.own_addr = 4 * (2 * (i & 7) + (i >> 3));
        s.oth_addr = 4 * (2 * (jj & 7) + (jj >> 3));
    }
    const int lane = s.lane, p = s.p, pos = s.pos;
    const int wave_global = blockIdx.x * C::WAVES + wave;
    const int waves_total = gridDim.x * C::WAVES;

    STAMP(0);
    for (int frame = wave_global; frame < P.B; frame = next_job_wave(P.queue, frame, waves_total, P.B)) {
        s.src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
        if constexpr (C::BIG) {
            // root f, single path: the left child of the root is shared by every path (computed before any fork)
            s.scr = reinterpret_cast<R *>(P.scratch) + (size_t)wave_global * C::scratch_elems;
            R *t = s.tls();
            STAMP(1);
#pragma unroll 2
            for (int e = lane; e < N / 2; e += 64) t[e] = s.chk(s.chv(e), s.chv(e + N / 2));
        } else {
            // ---- channel LLRs (SCL_1024.c:574-578) ----
#pragma unroll 4
            for (int i = lane; i < N; i += 64) s.ch[i] = s.chv(i);
            lds_fence();
            STAMP(1);
#pragma unroll 2
            for (int e = lane; e < N / 2; e += 64) s.tl[e] = s.chk(s.ch[e], s.ch[e + N / 2]);
        }
        // the top-level steps read beta words that the first half of the tree has not written yet: keep them defined
        for (int w = lane; w < 8 * NW; w += 64) s.blw[w] = 0;
        lds_fence();

        s.PM = R(0);
        s.ptr = 0;
        for (int t = 4; t <= C::HI; ++t) s.set_pa(t, p);
        for (int t = 5; t <= C::TOP; ++t) s.set_pb(t, p);
        s.crc = 0;
        s.bl0 = 0;
        s.a1 = s.a2 = s.a3 = R(0);
        s.fl = 0;
        s.logact = 0;
        uint32_t fword = 0;
        STAMP(2);

        for (int o = 0; o < N / 8; ++o) {
            if ((o & 3) == 0) fword = frz[o >> 2];
            s.octet_head(o);
            STAMP(3);
            const uint32_t fm = (fword >> (8 * (o & 3))) & 0xFFu;
            if ((fm & 0x7Fu) == 0x7Fu) { s.octet_frozen_prefix(o, fm == 0xFFu); STAMP(4); }
            else { s.octet(o, fm); STAMP(5); }
        }

        // ================= choose the path (SCL_1024.c:667-674; CASCL_1024_L8.c:725-755) =================
        const bool pass = CRC_ON && (s.crc == 0);
        const bool any = __ballot(pass && pos == 0) != 0ull;
        int best = -1;
        R best_pm = R(0);
        for (int q = 0; q < L; ++q) {
            const R pq = __shfl(s.PM, q * 8);
            const int okq = __shfl((int)(any ? pass : true), q * 8);
            if (okq && (best < 0 || pq < best_pm)) {
                best = q;
                best_pm = pq;
            }
        }
        uint32_t fl = s.fl;
        if (any) fl |= 0x2u;
        // x_hat = root partial sums of the winner (curw after the last leaf); u_hat = x_hat F^{(x)n}
        lds_fence();
        uint32_t *xw = s.curw + (size_t)best * NW;
        {
            uint32_t x = xw[lane & (NW - 1)];
            x ^= (x >> 1) & 0x55555555u;
            x ^= (x >> 2) & 0x33333333u;
            x ^= (x >> 4) & 0x0F0F0F0Fu;
            x ^= (x >> 8) & 0x00FF00FFu;
            x ^= (x >> 16) & 0x0000FFFFu;
#pragma unroll
            for (int hw = 1; hw < NW; hw <<= 1) {
                const uint32_t o = __shfl_down(x, hw);
                if (!(lane & hw)) x ^= o;
            }
            if (lane < NW) P.out_bits[(size_t)frame * NW + lane] = x;
        }
        if (lane == 0) {
            if (P.pm) P.pm[frame] = (double)best_pm;
            if (P.flags) P.flags[frame] = fl;
        }
        lds_fence();
        STAMP(6);
    }
#ifdef POLAR_STAMPS
    if (P.dbg && lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&P.dbg[i], tsec[i]);
#endif
}

}  // namespace polar
