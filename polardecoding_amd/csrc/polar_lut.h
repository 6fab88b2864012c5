// polar_lut.h -- table-driven form of the reference's 8-level staircase T (SCL_1024.c:352-359) and of the check
// node CHK (SCL_1024.c:343-374), shared by the tuned SCL kernel and the BP kernel.  Bit-exact by construction:
// the look-up ends in ONE exact compare against the only threshold that can lie in the operand's cell, and
// T(s) - T(d) is read from an 8x8 table of the IEEE differences of the table constants.
#pragma once
#include "polar_math.h"

namespace polar {

// ---- table-driven staircase -----------------------------------------------------------------------
// cell(x) for x >= 0: 0 for x < 0.125, 1..48 = 8 cells per binade over [0.125, 8), 49 for x >= 8.
// The seven thresholds 0.196 .. 4.5 (SCL_1024.c:352-358) fall into seven different cells, so ONE exact
// compare against the cell's threshold finishes the 8-level look-up.
// LDS entry (48 bytes): [thr | s_lo s_hi] [thr | d_lo d_hi] [T_lo T_hi], where s_*/d_* are the byte offsets
// of the row / column of the 8x8 table of differences for "below thr" / "at or above thr".
template <typename R>
struct Cell;
template <>
struct Cell<double> {
    static constexpr int BIAS = 0x3FC00000 >> 17;  // cell number of 0.125
    static __device__ __forceinline__ int raw(double x) { return (int)__builtin_amdgcn_ubfe((unsigned)__double2hiint(x), 17, 14); }
};
template <>
struct Cell<float> {
    static constexpr int BIAS = 0x3E000000 >> 20;
    static __device__ __forceinline__ int raw(float x) { return (int)__builtin_amdgcn_ubfe((unsigned)__float_as_int(x), 20, 11); }
};

template <typename R>
struct Lut {
    struct __attribute__((aligned(16))) Q { R thr; int pad_[(8 - sizeof(R)) / 4 + 0]; int lo, hi; };  // {thr @0, lo @8, hi @12}
    struct __attribute__((aligned(16))) TP { R lo, hi; };
    static constexpr int NCELL = 50, STRIDE = 48;
    static constexpr size_t cell_bytes = (size_t)NCELL * STRIDE;
    static constexpr size_t copy_bytes = cell_bytes + sizeof(R) * 64;   // one copy: cells + 8x8 differences
#ifndef POLAR_LUT_COPIES
#define POLAR_LUT_COPIES 1
#endif
    // COPIES > 1: lane l uses copy l % COPIES, which spreads the random cell reads over more LDS banks
    static constexpr int COPIES = POLAR_LUT_COPIES;
    static constexpr size_t bytes = copy_bytes * COPIES;
    unsigned base;  // LDS byte address of the table, pre-biased: entry(x) = base + clamp(raw(x)) * 48
    const unsigned char *lds0;  // LDS address 0 as a pointer (keeps the address space known)
    const R *dlt;   // dlt[i*8+j] = T_i - T_j (one IEEE subtraction, like `delta = T(s); delta -= T(d)`)

    static __device__ __forceinline__ int index_of(R x)
    {
        const int t = Cell<R>::raw(x);
        return min(max(t, Cell<R>::BIAS - 1), Cell<R>::BIAS + 48) - (Cell<R>::BIAS - 1);
    }
    __device__ __forceinline__ void bind(unsigned char *tab)
    {
        if (COPIES > 1) tab += (size_t)(threadIdx.x % COPIES) * copy_bytes;
        lds0 = tab;
        base = (unsigned)(0 - (Cell<R>::BIAS - 1) * STRIDE);
        if (COPIES == 1) __asm__ volatile("" : "+s"(base));  // opaque: keeps the bias inside the multiply-add
        else __asm__ volatile("" : "+v"(base));
        dlt = reinterpret_cast<const R *>(tab + cell_bytes);
    }
    typedef int i2 __attribute__((ext_vector_type(2)));
    // offset selected by the exact compare |x| >= thr of the 16-byte sub-entry {thr @0, (lo, hi) @8} at `sub`
    __device__ __forceinline__ int pick(R x, unsigned sub) const
    {
        const unsigned e = entry(x) + sub;
        const R thr = *reinterpret_cast<const R *>(lds0 + e);
        const i2 lh = *reinterpret_cast<const i2 *>(lds0 + e + 8);
        return (absr(x) >= thr) ? lh.y : lh.x;
    }
    // executed by a whole workgroup before its first barrier
    static __device__ void build(unsigned char *tab0, int tid, int nthreads)
    {
      for (int cp = 0; cp < COPIES; ++cp) {
        unsigned char *tab = tab0 + (size_t)cp * copy_bytes;
        R *d = reinterpret_cast<R *>(tab + cell_bytes);
        const R thr[7] = {R(0.196), R(0.433), R(0.71), R(1.05), R(1.508), R(2.252), R(4.5)};
        const R tv[8] = {R(0.65), R(0.55), R(0.45), R(0.35), R(0.25), R(0.15), R(0.05), R(0)};
        for (int i = tid; i < NCELL; i += nthreads) {
            int b = 0;
            R t = R(__builtin_huge_val());
            for (int k = 0; k < 7; ++k) {
                const int ck = index_of(thr[k]);
                if (ck < i) ++b;
                if (ck == i) t = thr[k];
            }
            const int b1 = b < 7 ? b + 1 : 7;
            Q *qs = reinterpret_cast<Q *>(tab + i * STRIDE);
            Q *qd = reinterpret_cast<Q *>(tab + i * STRIDE + 16);
            TP *tp = reinterpret_cast<TP *>(tab + i * STRIDE + 32);
            qs->thr = t; qs->lo = b * 8 * (int)sizeof(R); qs->hi = b1 * 8 * (int)sizeof(R);
            qd->thr = t; qd->lo = b * (int)sizeof(R); qd->hi = b1 * (int)sizeof(R);
            tp->lo = tv[b]; tp->hi = tv[b1];
        }
        for (int i = tid; i < 64; i += nthreads) d[i] = tv[i >> 3] - tv[i & 7];
      }
    }
    // byte offset (from lds0) of the entry of |x|
    __device__ __forceinline__ unsigned entry(R x) const
    {
        const int t = Cell<R>::raw(x);
        const int c = min(max(t, Cell<R>::BIAS - 1), Cell<R>::BIAS + 48);
        return __umul24((unsigned)c, (unsigned)STRIDE) + base;
    }
    // select by a sign mask (all ones: a, zero: b) -- two full-rate v_bfi_b32 instead of two v_cndmask_b32 for a double
    // (inline asm: written in C the compiler turns mask-and-merge back into v_cmp + v_cndmask)
    static __device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b)
    {
        uint32_t r;
        __asm__("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double sel_mask(uint32_t m, double a, double b)
    {
        const uint32_t lo = bfi(m, (uint32_t)__double2loint(a), (uint32_t)__double2loint(b));
        const uint32_t hi = bfi(m, (uint32_t)__double2hiint(a), (uint32_t)__double2hiint(b));
        return __hiloint2double((int)hi, (int)lo);
    }
    static __device__ __forceinline__ float sel_mask(uint32_t m, float a, float b)
    {
        return __int_as_float((int)bfi(m, (uint32_t)__float_as_int(a), (uint32_t)__float_as_int(b)));
    }
    static __device__ __forceinline__ uint32_t neg_mask(double t) { return (uint32_t)(__double2hiint(t) >> 31); }
    static __device__ __forceinline__ uint32_t neg_mask(float t) { return (uint32_t)(__float_as_int(t) >> 31); }
    // T(|x|) in one LDS round trip
    __device__ __forceinline__ R tabv(R x) const
    {
#ifdef POLAR_TABV_MASK   // measured round 3: 4.5 % SLOWER than compare + v_cndmask in the pair kernel (profiles/r03_ab_experiments.txt run 24)
        {   // |x| - thr is negative exactly when |x| < thr (equal: +0; thr = +inf in a cell without threshold: -inf, and
            // both T are the same there): its sign bit, spread over the word, selects T_lo; no compare, no VCC, no v_cndmask
            const unsigned e1 = entry(x);
            const R thr1 = *reinterpret_cast<const R *>(lds0 + e1);
            const TP tp1 = *reinterpret_cast<const TP *>(lds0 + e1 + 32);
            return sel_mask(neg_mask(absr(x) - thr1), tp1.lo, tp1.hi);
        }
#endif
#ifdef POLAR_SENS_NOLUT   // timing sensitivity experiment only (WRONG values): no LDS traffic, same compare + select
        const unsigned e0 = entry(x);
        const R thr0 = R(1.05) + R(e0 & 1u);
        return (absr(x) >= thr0) ? R(0.05) : R(0.45);
#endif
        const unsigned e = entry(x);
        const R thr = *reinterpret_cast<const R *>(lds0 + e);
        const TP tp = *reinterpret_cast<const TP *>(lds0 + e + 32);
        return (absr(x) >= thr) ? tp.hi : tp.lo;
    }
};

// max(x, 0) (x = +-lambda): the metric penalty |lambda| or 0, one instruction
__device__ __forceinline__ double posmax(double x)
{
    double m;
    __asm__("v_max_f64 %0, %1, 0" : "=v"(m) : "v"(x));
    return m;
}
__device__ __forceinline__ float posmax(float x)
{
    float m;
    __asm__("v_max_f32_e64 %0, %1, 0" : "=v"(m) : "v"(x));
    return m;
}

// max(-x, 0)
__device__ __forceinline__ double negmax(double x)
{
    double m;
    __asm__("v_max_f64 %0, -%1, 0" : "=v"(m) : "v"(x));
    return m;
}
__device__ __forceinline__ float negmax(float x)
{
    float m;
    __asm__("v_max_f32_e64 %0, -%1, 0" : "=v"(m) : "v"(x));
    return m;
}

// min(|a|, |b|) in one instruction (the generic fmin lowering canonicalises both operands first)
__device__ __forceinline__ double minabs(double a, double b)
{
    double m;
    __asm__("v_min_f64 %0, |%1|, |%2|" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ float minabs(float a, float b)
{
    float m;
    __asm__("v_min_f32_e64 %0, |%1|, |%2|" : "=v"(m) : "v"(a), "v"(b));
    return m;
}
// m >= 0 with the sign of a*b: (m & 0x7fffffff) | ((a ^ b) & 0x80000000) as one v_bfi
// (the sign-bit constant comes from an SGPR: VOP3 takes no literal on gfx9, and without it the compiler splits the
// operation into v_and + v_or)
__device__ __forceinline__ unsigned and_or_sign(unsigned x, unsigned mhi)
{
#ifdef POLAR_SIGN_OLD
    return (x & 0x80000000u) | mhi;
#else
    unsigned r;
    __asm__("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(0x80000000u), "v"(mhi));
    return r;
#endif
}
__device__ __forceinline__ double xor_sign(double m, double a, double b)
{
    const unsigned x = (unsigned)(__double2hiint(a) ^ __double2hiint(b));
    const unsigned hi = and_or_sign(x, (unsigned)__double2hiint(m));  // m >= 0
    return __hiloint2double((int)hi, __double2loint(m));
}
__device__ __forceinline__ float xor_sign(float m, float a, float b)
{
    const unsigned x = (unsigned)(__float_as_int(a) ^ __float_as_int(b));
    return __int_as_float((int)and_or_sign(x, (unsigned)__float_as_int(m)));
}

// One-round-trip form: T(|s|) and T(|d|) are selected from the cell entries and subtracted here (the reference's
// own `delta = T(s); delta -= T(d)`), instead of a second, dependent read of the 8x8 difference table.
// Four more issue slots, one LDS latency less: for the serial narrow-level chains.
template <typename R>
__device__ __forceinline__ R chk_lut1(R a, R b, const Lut<R> &L)
{
    const R s = a + b, d = a - b;
    const R delta = L.tabv(s) - L.tabv(d);
    return xor_sign(minabs(a, b), a, b) + delta;
}

// CHK (SCL_1024.c:343-374) with the staircase taken from the tables.  sign(a)sign(b) is applied by
// xor of the sign bits: for a = -0.0 the reference uses +1, but then min = 0 and delta = +0, and
// (+-0) + (+0) = +0 either way, so the result is identical.
template <typename R>
__device__ __forceinline__ R chk_lut(R a, R b, const Lut<R> &L)
{
#ifdef POLAR_CHK_LUT_IS_LUT1   // experiment switch (tools/variant.py): the one-round-trip form wherever the compact one is used
    return chk_lut1<R>(a, b, L);
#endif
    const R s = a + b, d = a - b;
    const int os = L.pick(s, 0), od = L.pick(d, 16);
    const R delta = *reinterpret_cast<const R *>(reinterpret_cast<const unsigned char *>(L.dlt) + (os + od));
    return xor_sign(minabs(a, b), a, b) + delta;
}


// ---- the staircase on the VALU (for BP, where the table reads of the forms above saturate the LDS pipe) ----------
// n(x) = #{k : |x| < thr_k}, 0..7, exact: |x| - thr_k is negative exactly when |x| < thr_k (an IEEE difference has the
// sign of the exact difference, and +0 when they are equal), and v_alignbit shifts that sign bit into an accumulator:
// one subtraction and one full-rate instruction per threshold, no compare, no select, no memory.  T(|x|) is then
// tv[7 - n] (SCL_1024.c:352-359), and T(|s|) - T(|d|) one read of an 8x8 table of the IEEE differences.
__device__ __forceinline__ uint32_t hi_word(double x) { return (uint32_t)__double2hiint(x); }
__device__ __forceinline__ uint32_t hi_word(float x) { return (uint32_t)__float_as_int(x); }

template <typename R>
__device__ __forceinline__ uint32_t below_count(R x)
{
    const R a = absr(x);
    uint32_t acc = 0;
    acc = __builtin_amdgcn_alignbit(acc, hi_word(a - R(0.196)), 31);
    acc = __builtin_amdgcn_alignbit(acc, hi_word(a - R(0.433)), 31);
    acc = __builtin_amdgcn_alignbit(acc, hi_word(a - R(0.71)), 31);
    acc = __builtin_amdgcn_alignbit(acc, hi_word(a - R(1.05)), 31);
    acc = __builtin_amdgcn_alignbit(acc, hi_word(a - R(1.508)), 31);
    acc = __builtin_amdgcn_alignbit(acc, hi_word(a - R(2.252)), 31);
    acc = __builtin_amdgcn_alignbit(acc, hi_word(a - R(4.5)), 31);
    return (uint32_t)__popc(acc);
}

// dn[ns * 8 + nd] = T(level 7 - ns) - T(level 7 - nd): built once per workgroup into LDS (512 B in f64)
template <typename R>
__device__ void build_delta_by_count(R *dn, int tid, int nthreads)
{
    const R tv[8] = {R(0.65), R(0.55), R(0.45), R(0.35), R(0.25), R(0.15), R(0.05), R(0)};
    for (int i = tid; i < 64; i += nthreads) dn[i] = tv[7 - (i >> 3)] - tv[7 - (i & 7)];
}

// CHK (SCL_1024.c:343-374) with the staircase counted on the VALU; same value as chk / chk_lut / chk_lut1
template <typename R>
__device__ __forceinline__ R chk_cnt(R a, R b, const R *dn)
{
    const uint32_t ns = below_count<R>(a + b), nd = below_count<R>(a - b);
    const R delta = dn[ns * 8 + nd];
    return xor_sign(minabs(a, b), a, b) + delta;
}

// ---- the staircase with ONE small table read (for BP) -----------------------------------------------------------
// Four cells per binade over [0.125, 8) (26 cells with the two ends) still separate the seven thresholds, so the number
// b of thresholds in the cells below x's own is a prefix popcount of a 26-bit mask -- three full-rate instructions --
// and the only threshold that can still lie at or below |x| is thr[b] (the next one up; +inf after the last): the
// level index is b + (|x| >= thr[b]).  The read of thr[b] touches eight addresses in sixteen different LDS banks: no
// conflicts, 8 bytes per look-up instead of 16 from 50 scattered cells; T(|s|) - T(|d|) is one more 8-byte read.
template <typename R>
struct Cell4;
template <>
struct Cell4<double> {
    static constexpr int BIAS = 0x3FC00000 >> 18;   // cell number of 0.125
    static __device__ __forceinline__ int raw(double x) { return (int)__builtin_amdgcn_ubfe((unsigned)__double2hiint(x), 18, 13); }
};
template <>
struct Cell4<float> {
    static constexpr int BIAS = 0x3E000000 >> 21;
    static __device__ __forceinline__ int raw(float x) { return (int)__builtin_amdgcn_ubfe((unsigned)__float_as_int(x), 21, 10); }
};

template <typename R>
struct Stair {
    static constexpr size_t bytes = sizeof(R) * (8 + 64) + 32;
    const R *thr;    // LDS [8]: the seven thresholds, then +inf
    const R *dlt;    // LDS [64]: dlt[i * 8 + j] = T_i - T_j
    uint32_t mask;   // bit c set: cell c holds a threshold
    const unsigned char *pfx;   // LDS [26] bytes, by cell: the number of thresholds in the cells below

    static __device__ __forceinline__ int cell_of(R x)
    {
        const int t = Cell4<R>::raw(x);
        return min(max(t, Cell4<R>::BIAS - 1), Cell4<R>::BIAS + 24) - (Cell4<R>::BIAS - 1);
    }
    static __device__ void build(unsigned char *mem, int tid, int nthreads)
    {
        R *t = reinterpret_cast<R *>(mem);
        const R thr7[8] = {R(0.196), R(0.433), R(0.71), R(1.05), R(1.508), R(2.252), R(4.5), R(__builtin_huge_val())};
        const R tv[8] = {R(0.65), R(0.55), R(0.45), R(0.35), R(0.25), R(0.15), R(0.05), R(0)};
        for (int i = tid; i < 8; i += nthreads) t[i] = thr7[i];
        for (int i = tid; i < 64; i += nthreads) t[8 + i] = tv[i >> 3] - tv[i & 7];
        unsigned char *pf = reinterpret_cast<unsigned char *>(t + 72);
        for (int c = tid; c < 32; c += nthreads) {
            int nb = 0;
            for (int k = 0; k < 7; ++k) nb += (cell_of(thr7[k]) < c) ? 1 : 0;
            pf[c] = (unsigned char)nb;
        }
    }
    __device__ __forceinline__ void bind(const unsigned char *mem)
    {
        thr = reinterpret_cast<const R *>(mem);
        dlt = thr + 8;
        pfx = reinterpret_cast<const unsigned char *>(dlt + 64);
        const R thr7[7] = {R(0.196), R(0.433), R(0.71), R(1.05), R(1.508), R(2.252), R(4.5)};
        mask = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) mask |= 1u << cell_of(thr7[k]);
    }
    // level index of |x|: #{k : thr_k <= |x|}
    __device__ __forceinline__ uint32_t level(R x) const
    {
        const uint32_t c = (uint32_t)cell_of(x);
        const uint32_t b = (uint32_t)__popc(mask & ((1u << c) - 1u));
        return b + ((absr(x) >= thr[b]) ? 1u : 0u);
    }
    // the same with the prefix count read from a 26-byte table instead of mask / shift / popcount: three VALU instructions
    // less per look-up for one more (conflict-free: seven consecutive words) LDS read in the dependent chain
    __device__ __forceinline__ uint32_t level_tab(R x) const
    {
        const int t = min(max(Cell4<R>::raw(x), Cell4<R>::BIAS - 1), Cell4<R>::BIAS + 24);
        const uint32_t b = pfx[(uint32_t)t - (uint32_t)(Cell4<R>::BIAS - 1)];
        return b + ((absr(x) >= thr[b]) ? 1u : 0u);
    }
};

template <typename R>
__device__ __forceinline__ R chk_idx(R a, R b, const Stair<R> &S)
{
    const uint32_t is = S.level(a + b), id = S.level(a - b);
    const R delta = S.dlt[is * 8 + id];
    return xor_sign(minabs(a, b), a, b) + delta;
}

template <typename R>
__device__ __forceinline__ R chk_tab(R a, R b, const Stair<R> &S)
{
    const uint32_t is = S.level_tab(a + b), id = S.level_tab(a - b);
    const R delta = S.dlt[is * 8 + id];
    return xor_sign(minabs(a, b), a, b) + delta;
}

}  // namespace polar
