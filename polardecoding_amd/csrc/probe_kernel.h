// probe_kernel.h -- TEST-ONLY kernel behind polar_testing_math (include/polar_hip_testing.h): puts caller-chosen
// operands through the device functions the decoders inline, so that the table form of the check node can be
// compared with the reference's compare chain on the operands where they could differ (thresholds +-1 ulp, equal
// magnitudes, signed zeros, denormals, values beyond the table).  Not launched by any product entry point.
#pragma once
#include "polar_lut.h"

namespace polar {

enum { PROBE_CHK = 0, PROBE_CHK_LUT = 1, PROBE_CHK_LUT1 = 2, PROBE_TABV = 3, PROBE_PHI = 4, PROBE_PHI_LUT = 5, PROBE_CHK_CNT = 6, PROBE_CHK_IDX = 7, PROBE_CHK_TAB = 8 };

// the metric increment exactly as the list kernels write it (scl_fast2.h decide, scl_big.h, scl_generic.h)
template <typename R>
__device__ __forceinline__ R phi_lut(R lam, int u, const Lut<R> &L)
{
    const R tt = L.tabv(lam);
    return u ? tt + posmax(lam) : tt + negmax(lam);
}

template <typename R>
__global__ void __launch_bounds__(256) k_probe_math(int op, const R *a, const R *b, R *out, size_t n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Lut<R>::build(smem, (int)threadIdx.x, (int)blockDim.x);
    R *dn = reinterpret_cast<R *>(smem + ((Lut<R>::bytes + 15) / 16) * 16);
    build_delta_by_count<R>(dn, (int)threadIdx.x, (int)blockDim.x);
    unsigned char *stm = reinterpret_cast<unsigned char *>(dn + 64);
    Stair<R>::build(stm, (int)threadIdx.x, (int)blockDim.x);
    Stair<R> st;
    st.bind(stm);
    __syncthreads();
    Lut<R> lut;
    lut.bind(smem);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const R x = a[i], y = b[i];
        R r;
        switch (op) {
        case PROBE_CHK: r = chk<R>(x, y); break;
        case PROBE_CHK_LUT: r = chk_lut<R>(x, y, lut); break;
        case PROBE_CHK_LUT1: r = chk_lut1<R>(x, y, lut); break;
        case PROBE_TABV: r = lut.tabv(x); break;
        case PROBE_PHI: r = phi<R>(x, y != R(0) ? 1 : 0); break;
        case PROBE_CHK_CNT: r = chk_cnt<R>(x, y, dn); break;
        case PROBE_CHK_IDX: r = chk_idx<R>(x, y, st); break;
        case PROBE_CHK_TAB: r = chk_tab<R>(x, y, st); break;
        default: r = phi_lut<R>(x, y != R(0) ? 1 : 0, lut); break;
        }
        out[i] = r;
    }
}

}  // namespace polar
