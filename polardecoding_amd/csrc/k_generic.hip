// k_generic.hip -- k_scl_generic (correctness baseline, short codes, single-frame SC) and its launch code
#include "polar_host.h"
#include "scl_generic.h"

namespace {

template <typename R, typename IN, int LOGL, bool GA>
int launch_scl_v(polar_ctx *c, const polar::SclParams &P)
{
    auto kern = polar::k_scl_generic<R, IN, LOGL, GA>;
    const size_t lds = polar::scl_generic_lds_bytes<R, LOGL>(P.N, GA);
    if (lds > 160 * 1024) return POLAR_ENOKERNEL;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64, lds));
    if (occ < 1) occ = 1;
    if (GA && occ > 8) occ = 8;
    int grid = std::min<long long>((long long)P.B, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::SclParams Q = P;
    if (GA) {
        const size_t bytes = sizeof(R) * (size_t)((1 << LOGL) + 1) * P.N * (size_t)grid;
        int rc = ensure(c, c->scratch, bytes);
        if (rc) return rc;
        Q.scratch = c->scratch.p;
    }
    if ((long long)P.B > (long long)grid) {
        int rc = work_queue(c, c->scratch, &Q.queue);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

template <typename R, typename IN, int LOGL>
int launch_scl(polar_ctx *c, const polar::SclParams &P)
{
    if (polar::scl_generic_lds_bytes<R, LOGL>(P.N, false) <= 160 * 1024 && !c->force_spill)
        return launch_scl_v<R, IN, LOGL, false>(c, P);
    return launch_scl_v<R, IN, LOGL, true>(c, P);
}

template <typename R, typename IN>
int launch_scl_l(polar_ctx *c, const polar::SclParams &P)
{
    switch (c->logL) {
    case 0: return launch_scl<R, IN, 0>(c, P);
    case 1: return launch_scl<R, IN, 1>(c, P);
    case 2: return launch_scl<R, IN, 2>(c, P);
    case 3: return launch_scl<R, IN, 3>(c, P);
    case 4: return launch_scl<R, IN, 4>(c, P);
    case 5: return launch_scl<R, IN, 5>(c, P);
    }
    return POLAR_ENOKERNEL;
}

}  // namespace

int polar_tu::scl_generic(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32)
{
    if (r32) return in32 ? launch_scl_l<float, float>(c, P) : launch_scl_l<float, double>(c, P);
    return in32 ? launch_scl_l<double, float>(c, P) : launch_scl_l<double, double>(c, P);
}
