// polar_math.h -- device arithmetic shared by every decoder kernel (gfx950).
//
// The reference's check node is NOT min-sum: it is sign*min plus a two-sided 8-level table
// correction (CHK, SCL_1024.c:343-374, identical in all reference programs), and the SCL path
// metric uses the same table (PHI, SCL_1024.c:481-502).  Everything here keeps the reference's
// operation order so that the f64 instantiation is bit-identical: one rounding in a+b, a-b,
// T(s)-T(d) and the final add.  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace polar {

// T(a), a >= 0: staircase approximation of ln(1+e^-a) (SCL_1024.c:352-359).
template <typename R>
__device__ __forceinline__ R tab(R a)
{
    R t = R(0.65);
    t = (a >= R(0.196)) ? R(0.55) : t;
    t = (a >= R(0.433)) ? R(0.45) : t;
    t = (a >= R(0.71)) ? R(0.35) : t;
    t = (a >= R(1.05)) ? R(0.25) : t;
    t = (a >= R(1.508)) ? R(0.15) : t;
    t = (a >= R(2.252)) ? R(0.05) : t;
    t = (a >= R(4.5)) ? R(0) : t;
    return t;
}

__device__ __forceinline__ double absr(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float absr(float x) { return __builtin_fabsf(x); }

// sign(a)*sign(b) applied to m >= 0, with the reference's sgn(0) = +1 (SCL_1024.c:366-367).
// (a >= 0) is false only for a < 0 (and NaN); -0.0 counts as +.
template <typename R>
__device__ __forceinline__ R apply_sign(R m, R a, R b)
{
    bool neg = (a < R(0)) != (b < R(0));
    return neg ? -m : m;
}

// CHK(L1, L2)  (SCL_1024.c:343-374)
template <typename R>
__device__ __forceinline__ R chk(R a, R b)
{
    R s = absr(a + b);
    R d = absr(a - b);
    R delta = tab<R>(s);
    delta -= tab<R>(d);
    R A = absr(a), Bm = absr(b);
    R m = (A > Bm) ? Bm : A;
    return apply_sign<R>(m, a, b) + delta;
}

// PHI(k, j, u) with lambda = V[0][j]->l[k]  (SCL_1024.c:481-502)
template <typename R>
__device__ __forceinline__ R phi(R lam, int u)
{
    R a = absr(lam);
    R r = tab<R>(a);
    bool pen = (u == 0) ? (lam < R(0)) : (lam > R(0));
    return pen ? r + a : r;
}

// lower-node update of getLLR (SCL_1024.c:412-416): cL + cU if the partner bit is 0, cL - cU otherwise
template <typename R>
__device__ __forceinline__ R gfun(R cU, R cL, int bit)
{
    return bit ? cL - cU : cL + cU;
}

// The same with the partner bit taken from bit `sh` of a word: cL - cU = cL + (-cU) exactly, so the bit only has to reach the
// sign of cU.  (w >> sh) << 31 keeps exactly bit sh of w, and ADDING 2^31 to the high word flips the sign bit (the carry
// leaves the word): one shift and one v_lshl_add_u32 instead of mask, compare and two selects.
__device__ __forceinline__ double flip_bit(double x, uint32_t t) { return __hiloint2double((int)((t << 31) + (uint32_t)__double2hiint(x)), __double2loint(x)); }
__device__ __forceinline__ float flip_bit(float x, uint32_t t) { return __int_as_float((int)((t << 31) + (uint32_t)__float_as_int(x))); }
template <typename R>
__device__ __forceinline__ R g_bit(R cU, R cL, uint32_t w, int sh)
{
    return cL + flip_bit(cU, w >> sh);
}

// channel LLR from an observation: 2*y/std/std, that order (SCL_1024.c:576), always in double
__device__ __forceinline__ double llr_from_y(double y, double sigma) { return 2 * y / sigma / sigma; }

}  // namespace polar
