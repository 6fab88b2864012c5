// k_fast.hip -- k_scl_fast (L = 8: N = 128, and N = 1024 one codeword per wavefront) and its launch code
#include "polar_host.h"
#include "scl_fast.h"

namespace {


// tuned instantiations: L = 8, N in {128, 1024}
template <typename R, typename IN, int NLOG, bool CRC_ON>
int launch_fast(polar_ctx *c, const polar::SclParams &P)
{
    auto kern = polar::k_scl_fast<R, IN, NLOG, CRC_ON>;
    constexpr int WAVES = polar::FastCfg<R, NLOG>::WAVES;
    const size_t lds = polar::FastCfg<R, NLOG>::total;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WAVES, lds));
    if (occ < 1) occ = 1;
    long long blocks_needed = ((long long)P.B + WAVES - 1) / WAVES;
    int grid = (int)std::min<long long>(blocks_needed, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::SclParams Q = P;
    const size_t sc_bytes = polar::FastCfg<R, NLOG>::scratch_elems * sizeof(R) * (size_t)grid * WAVES;
    if (sc_bytes) {
        int rc = ensure(c, c->scratch, sc_bytes);
        if (rc) return rc;
        Q.scratch = c->scratch.p;
    }
    // work queue (polar_host.h): + 12 % for N = 128 in f64 (67.8 -> 76.1 M frames/s on one box); the f32 kernel at N = 128 is
    // not short of issue slots and loses 3 % to it (89.5 -> 86.5 M): fixed stride there
    constexpr bool QUEUE = !(sizeof(R) == 4 && NLOG == 7);
    if (QUEUE && (long long)P.B > (long long)grid * WAVES) {
        int rc = work_queue(c, c->scratch, &Q.queue);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

template <typename R, typename IN>
int launch_fast_n(polar_ctx *c, const polar::SclParams &P, bool crc)
{
    if (P.N == 1024) return crc ? launch_fast<R, IN, 10, true>(c, P) : launch_fast<R, IN, 10, false>(c, P);
    if (P.N == 128) return crc ? launch_fast<R, IN, 7, true>(c, P) : launch_fast<R, IN, 7, false>(c, P);
    return POLAR_ENOKERNEL;
}

}  // namespace

int polar_tu::scl_fast(polar_ctx *c, const polar::SclParams &P, bool r32, bool in32, bool crc)
{
    if (!r32) return in32 ? POLAR_ENOKERNEL : launch_fast_n<double, double>(c, P, crc);
    return in32 ? launch_fast_n<float, float>(c, P, crc) : launch_fast_n<float, double>(c, P, crc);
}
