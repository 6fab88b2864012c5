// k_fast2.hip -- k_scl_fast2 (L = 8, N = 1024, two codewords per wavefront: the headline kernel) and its launch code
#include "polar_host.h"
#include "scl_fast2.h"

namespace {

// two codewords per wavefront (scl_fast2.h), N = 1024, L = 8
template <typename R, typename IN, bool CRC_ON>
int launch_fast2(polar_ctx *c, const polar::SclParams &P)
{
    using Cfg = polar::Fast2Cfg<R>;
    auto kern = polar::k_scl_fast2<R, IN, CRC_ON>;
    constexpr int WAVES = Cfg::WAVES;
    const size_t lds = Cfg::total;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WAVES, lds));
    if (occ < 1) occ = 1;
    const long long pairs = ((long long)P.B + 1) / 2;
    long long blocks_needed = (pairs + WAVES - 1) / WAVES;
#ifndef POLAR_F2_GRID_MULT
#define POLAR_F2_GRID_MULT 1
#endif
    int grid = (int)std::min<long long>(blocks_needed, (long long)occ * c->num_cu * POLAR_F2_GRID_MULT);
    if (grid < 1) grid = 1;
    polar::SclParams Q = P;
    const size_t sc_bytes = Cfg::scratch_elems * sizeof(R) * (size_t)grid * WAVES;
    int rc = ensure(c, c->scratch, sc_bytes);
    if (rc) return rc;
    Q.scratch = c->scratch.p;
    if (pairs > (long long)grid * WAVES) {   // more jobs than resident wavefronts: the rest through the work queue
        rc = work_queue(c, c->scratch, &Q.queue);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

}  // namespace

int polar_tu::scl_fast2(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32, bool crc)
{
    if (P.N != 1024) return POLAR_ENOKERNEL;
    if (!r32) {
        if (in32) return POLAR_ENOKERNEL;
        return crc ? launch_fast2<double, double, true>(c, P) : launch_fast2<double, double, false>(c, P);
    }
    if (in32) return crc ? launch_fast2<float, float, true>(c, P) : launch_fast2<float, float, false>(c, P);
    return crc ? launch_fast2<float, double, true>(c, P) : launch_fast2<float, double, false>(c, P);
}
