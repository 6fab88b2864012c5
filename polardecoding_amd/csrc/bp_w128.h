// bp_w128.h -- flooding BP for N = 128 (reference: BP, BP_128.c:334-388): ONE codeword per wavefront, every
// message in registers.
//
// k_bp (bp_kernel.h) keeps the 2 x 6 message rows of a codeword in LDS and walks them with one butterfly per thread and
// stage: six message accesses and two table look-ups per butterfly, all of them dependent LDS round trips, 13.5 KB of LDS
// per codeword (eleven wavefronts per CU).  At n = 7 the whole state is 12 rows x 128 values: lane l owns elements l and
// l + 64 of every row, 24 values per lane.  Stage 6 (stride 64) pairs a lane's own two elements; stages 0..5 pair lane l
// with lane l ^ 2^i, fetched with DPP (strides 1, 2, 4, 8) or through the LDS crossbar (ds_swizzle / ds_bpermute, strides
// 16, 32: no memory is touched).  Both elements of a lane are in the same half of their butterflies, so the two butterflies
// a lane works on per stage are independent instruction streams.
//
// A butterfly at stage i couples rows j ("upper", bit i of j clear) and j + 2^i ("lower").  With own = the lane's element and
// p = the partner's (BP_128.c:357-376; the operand order inside CHK and inside the sums is the reference's):
//   right-going sweep, r[i+1] from r[i] and l[i+1]:   upper: CHK(r_own, l_p + r_p)      lower: r_own + CHK(r_p, l_p)
//   left-going sweep,  l[i]   from r[i] and l[i+1]:   upper: CHK(l_own, l_p + r_p)      lower: l_own + CHK(r_p, l_p)
// i.e. one CHK and at most one addition per element and stage, the operands selected by the lane's role.
// r[n] and, except in the last iteration, l[0] are not computed (nobody reads them), as in k_bp / k_bp_r4.
#pragma once
#include "polar_lut.h"
#include "polar_params.h"

namespace polar {

// CHK form of this kernel: 0 = chk_idx (24 bytes of LDS per CHK, six more VALU instructions), 2 = chk_lut1 (48 bytes, one
// round trip).  Measured, 2^18 frames, 100 iterations: f64 form 2 8.93 M frames/s, form 0 7.50 M; f32 form 2 12.6 M, form 0
// 9.3 M (k_bp, the LDS kernel this one replaces at N = 128: 2.60 / 4.34 M) -- with the messages in registers the LDS pipe has
// room for the table reads and the VALU is the bound.
#ifndef POLAR_BPW_CHK
#define POLAR_BPW_CHK 2
#endif

template <typename R>
struct BpW128Cfg {
    static constexpr int N = 128, n = 7, NW = 4, WAVES = 4;
    static constexpr size_t off_lut = 0;
    static constexpr size_t off_st = ((Lut<R>::bytes + 15) / 16) * 16;
    static constexpr size_t lds_bytes = off_st + ((Stair<R>::bytes + 15) / 16) * 16;
    static constexpr int MIN_WAVES_PER_SIMD = sizeof(R) == 8 ? 4 : 6;
};

// value of lane (l ^ S) for S = 1 .. 32
template <int S>
__device__ __forceinline__ int bpw_xor_i(int v)
{
    if constexpr (S == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    else if constexpr (S == 4) {                                                                  // row_shl:4 / row_shr:4 by bank
        const int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
    } else if constexpr (S == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);  // row_ror:8
    else if constexpr (S == 16) return __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1F);          // bit mode: lane ^ 16
    else return __builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ 32) << 2), v);
}
template <int S>
__device__ __forceinline__ double bpw_xor(double x)
{
    const long long b = __double_as_longlong(x);
    const int lo = bpw_xor_i<S>((int)b), hi = bpw_xor_i<S>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <int S>
__device__ __forceinline__ float bpw_xor(float x) { return __int_as_float(bpw_xor_i<S>(__float_as_int(x))); }

template <typename R, typename IN>
struct BpW128 {
    using C = BpW128Cfg<R>;
    static constexpr int CHKF = POLAR_BPW_CHK;
    Lut<R> lut;
    Stair<R> st;
    R rm[6][2];   // r[1..6], elements lane and lane + 64
    R lm[6][2];   // l[1..6]
    R ch[2];      // l[7]: channel LLRs
    R r0[2];      // r[0]: 999 for a frozen position, else 0 (BP_128.c:346-353)

    __device__ __forceinline__ R chk(R a, R b) const
    {
        if constexpr (CHKF == 0) return chk_idx<R>(a, b, st);
        else return chk_lut1<R>(a, b, lut);
    }
    // one element of one cross-lane stage: own / partner values of r[i] and l[i+1]; `lead` = r_own (right-going) or l_own
    template <int I>
    __device__ __forceinline__ R cross(bool upper, R lead, R r_own_for_fetch, R l_own_for_fetch) const
    {
        constexpr int S = 1 << I;
        const R rp = bpw_xor<S>(r_own_for_fetch), lp = bpw_xor<S>(l_own_for_fetch);
        const R t = lp + rp;                    // l[i+1][j+s] + r[i][j+s] of the upper half
        const R x = upper ? lead : rp;
        const R y = upper ? t : lp;
        const R c = chk(x, y);
        return upper ? c : lead + c;
    }
    template <int I>   // right-going stage I in 0..5: r[I+1] from r[I], l[I+1]
    __device__ __forceinline__ void stage_r(bool upper)
    {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const R r_own = (I == 0) ? r0[e] : rm[I == 0 ? 0 : I - 1][e];
            const R l_own = (I + 1 == C::n) ? ch[e] : lm[I][e];
            rm[I][e] = cross<I>(upper, r_own, r_own, l_own);
        }
    }
    template <int I>   // left-going stage I in 0..5: l[I] from r[I], l[I+1]; returns l[0] when I == 0
    __device__ __forceinline__ void stage_l(bool upper, R (&out)[2])
    {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const R r_own = (I == 0) ? r0[e] : rm[I == 0 ? 0 : I - 1][e];
            const R l_own = lm[I][e];           // I + 1 <= 6
            out[e] = cross<I>(upper, l_own, r_own, l_own);
        }
    }
    // left-going stage 6 (stride 64): the lane's own two elements, element 0 upper, element 1 lower
    __device__ __forceinline__ void stage_l6()
    {
        const R r0_ = rm[5][0], r1_ = rm[5][1], l0 = ch[0], l1 = ch[1];
        const R a = chk(l0, l1 + r1_);
        const R c = l1 + chk(r0_, l0);
        lm[5][0] = a;
        lm[5][1] = c;
    }
};

template <typename R, typename IN>
__global__ __launch_bounds__(256, (BpW128Cfg<R>::MIN_WAVES_PER_SIMD)) void k_bp_w128(BpParams P)
{
    using C = BpW128Cfg<R>;
    using D = BpW128<R, IN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Lut<R>::build(smem + C::off_lut, threadIdx.x, blockDim.x);
    Stair<R>::build(smem + C::off_st, threadIdx.x, blockDim.x);
    __syncthreads();
    D s;
    s.lut.bind(smem + C::off_lut);
    s.st.bind(smem + C::off_st);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = blockIdx.x * C::WAVES + wave, nw = gridDim.x * C::WAVES;
    // frozen priors of the lane's two elements: words 0,1 hold elements 0..63, words 2,3 elements 64..127
    const uint32_t f0 = (P.frozen[lane >> 5] >> (lane & 31)) & 1u, f1 = (P.frozen[2 + (lane >> 5)] >> (lane & 31)) & 1u;
    s.r0[0] = f0 ? R(999) : R(0);
    s.r0[1] = f1 ? R(999) : R(0);

    for (int frame = wg; frame < P.B; frame = next_job_wave(P.queue, frame, nw, P.B)) {
        const IN *src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * C::N;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            double v = (double)src[lane + 64 * e];
            if (P.sigma > 0) v = llr_from_y(v, P.sigma);
            s.ch[e] = (R)v;                      // BP_128.c:343-344
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                s.rm[i][e] = R(0);               // BP_128.c:346-348
                s.lm[i][e] = R(0);               // BP_128.c:340-342
            }
        R l0v[2] = {R(0), R(0)};
        for (int it = 0; it < P.iters; ++it) {
            // right-going sweep (BP_128.c:357-366); stage 6 would only produce r[7], which nobody reads
            s.template stage_r<0>(!(lane & 1));
            s.template stage_r<1>(!(lane & 2));
            s.template stage_r<2>(!(lane & 4));
            s.template stage_r<3>(!(lane & 8));
            s.template stage_r<4>(!(lane & 16));
            s.template stage_r<5>(!(lane & 32));
            // left-going sweep (BP_128.c:368-377)
            s.stage_l6();
            R t[2];
            s.template stage_l<5>(!(lane & 32), t); s.lm[4][0] = t[0]; s.lm[4][1] = t[1];
            s.template stage_l<4>(!(lane & 16), t); s.lm[3][0] = t[0]; s.lm[3][1] = t[1];
            s.template stage_l<3>(!(lane & 8), t);  s.lm[2][0] = t[0]; s.lm[2][1] = t[1];
            s.template stage_l<2>(!(lane & 4), t);  s.lm[1][0] = t[0]; s.lm[1][1] = t[1];
            s.template stage_l<1>(!(lane & 2), t);  s.lm[0][0] = t[0]; s.lm[0][1] = t[1];
            if (it + 1 == P.iters) s.template stage_l<0>(!(lane & 1), l0v);   // l[0] is needed only for the decision
        }
        // BP_128.c:379-387: frozen -> 0, else (l[0] + r[0] >= 0) -> 0
        const bool b0 = !f0 && !(l0v[0] + s.r0[0] >= R(0));
        const bool b1 = !f1 && !(l0v[1] + s.r0[1] >= R(0));
        const uint64_t m0 = __ballot(b0), m1 = __ballot(b1);
        if (lane < 4) {
            const uint64_t m = (lane < 2) ? m0 : m1;
            P.out_bits[(size_t)frame * C::NW + lane] = (uint32_t)(m >> (32 * (lane & 1)));
        }
    }
}

}  // namespace polar
