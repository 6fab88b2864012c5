// k_fast4.hip -- k_scl_fast4 (four codewords per wavefront: measured alternative).  TEST LIBRARY ONLY.
#include "polar_host.h"
#include "scl_fast4.h"

namespace {

// four codewords per wavefront (scl_fast4.h), N = 1024, L = 8
template <typename R, typename IN, bool CRC_ON>
int launch_fast4(polar_ctx *c, const polar::SclParams &P)
{
    using Cfg = polar::Fast4Cfg<R>;
    auto kern = polar::k_scl_fast4<R, IN, CRC_ON>;
    constexpr int WAVES = Cfg::WAVES;
    const size_t lds = Cfg::total;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WAVES, lds));
    if (occ < 1) occ = 1;
    const long long quads = ((long long)P.B + Cfg::CW - 1) / Cfg::CW;
    long long blocks_needed = (quads + WAVES - 1) / WAVES;
    int grid = (int)std::min<long long>(blocks_needed, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::SclParams Q = P;
    const size_t sc_bytes = Cfg::scratch_elems * sizeof(R) * (size_t)grid * WAVES;
    int rc = ensure(c, c->scratch, sc_bytes);
    if (rc) return rc;
    Q.scratch = c->scratch.p;
    if (quads > (long long)grid * WAVES && (rc = work_queue(c, c->scratch, &Q.queue))) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

}  // namespace

#ifndef POLAR_TESTING
#error "k_fast4.hip belongs to libpolar_hip_testing.so (-DPOLAR_TESTING)"
#endif
int polar_tu::scl_fast4(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32, bool crc)
{
    if (P.N != 1024) return POLAR_ENOKERNEL;
    if (!r32) {
        if (in32) return POLAR_ENOKERNEL;
        return crc ? launch_fast4<double, double, true>(c, P) : launch_fast4<double, double, false>(c, P);
    }
    if (in32) return crc ? launch_fast4<float, float, true>(c, P) : launch_fast4<float, float, false>(c, P);
    return crc ? launch_fast4<float, double, true>(c, P) : launch_fast4<float, double, false>(c, P);
}
