// k_sc.hip -- k_sc_lanes (SC, one codeword per lane) and its launch code
#include "polar_host.h"
#include "sc_lanes.h"

namespace {

// SC, one codeword per lane (sc_lanes.h)
template <typename R, typename IN>
int launch_sc_lanes(polar_ctx *c, const polar::SclParams &P)
{
    using Cfg = polar::ScLanesCfg<R>;
    auto kern = polar::k_sc_lanes<R, IN>;
    const size_t lds = Cfg::lds_bytes(P.N);
    const int threads = 64 * Cfg::WAVES;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds));
    if (occ < 1) occ = 1;
    const long long batches = ((long long)P.B + 63) / 64;
    int grid = (int)std::min<long long>((batches + Cfg::WAVES - 1) / Cfg::WAVES, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::SclParams Q = P;
    int rc = ensure(c, c->scratch, Cfg::scratch_bytes(P.N) * (size_t)grid * Cfg::WAVES);
    if (rc) return rc;
    Q.scratch = c->scratch.p;
    if (batches > (long long)grid * Cfg::WAVES && (rc = work_queue(c, c->scratch, &Q.queue))) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

}  // namespace

int polar_tu::sc_lanes(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32)
{
    if (!r32) return in32 ? launch_sc_lanes<double, float>(c, P) : launch_sc_lanes<double, double>(c, P);
    return in32 ? launch_sc_lanes<float, float>(c, P) : launch_sc_lanes<float, double>(c, P);
}
