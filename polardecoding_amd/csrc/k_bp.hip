// k_bp.hip -- flooding BP: k_bp_r4 (N = 1024), k_bp / k_bp_global (other N), k_bp_readout (BPr) and their launch code
#include "polar_host.h"
#include "bp_kernel.h"
#include "bp_r4.h"
#include "bp_w128.h"

namespace {

// N = 1024: the register-blocked kernel (bp_r4.h), two f64 codewords per CU
template <typename R, typename IN>
int launch_bp_r4(polar_ctx *c, const polar::BpParams &P)
{
    using Cfg = polar::BpR4Cfg<R>;
    auto kern = polar::k_bp_r4<R, IN>;
    const size_t lds = Cfg::lds_bytes;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cfg::THREADS, lds));
    if (occ < 1) occ = 1;
    int grid = (int)std::min<long long>((long long)P.B, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::BpParams Q = P;
    if ((long long)P.B > (long long)grid) {
        int rc = work_queue(c, c->scratch, &Q.queue);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::THREADS), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

// N = 128: one codeword per wavefront, every message in registers (bp_w128.h)
template <typename R, typename IN>
int launch_bp_w128(polar_ctx *c, const polar::BpParams &P)
{
    using Cfg = polar::BpW128Cfg<R>;
    auto kern = polar::k_bp_w128<R, IN>;
    const size_t lds = Cfg::lds_bytes;
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * Cfg::WAVES, lds));
    if (occ < 1) occ = 1;
    const long long blocks_needed = ((long long)P.B + Cfg::WAVES - 1) / Cfg::WAVES;
    int grid = (int)std::min<long long>(blocks_needed, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::BpParams Q = P;
    if ((long long)P.B > (long long)grid * Cfg::WAVES) {
        int rc = work_queue(c, c->scratch, &Q.queue);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * Cfg::WAVES), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

template <typename R, typename IN>
int launch_bp(polar_ctx *c, const polar::BpParams &P)
{
    if (P.N == 1024 && !c->force_generic) return launch_bp_r4<R, IN>(c, P);
    if (P.N == 128 && !c->force_generic) return launch_bp_w128<R, IN>(c, P);
    auto kern = polar::k_bp<R, IN>;
    const size_t lds = polar::bp_lds_bytes<R>(P.N, P.n);
    if (lds > 160 * 1024) {   // messages do not fit a CU's LDS: rows in global scratch
        auto kg = polar::k_bp_global<R, IN>;
        const size_t lds_g = 4 * (size_t)(P.N / 32) + 16 + polar::Lut<R>::bytes;
        int grid = (int)std::min<long long>((long long)P.B, (long long)2 * c->num_cu);
        if (grid < 1) grid = 1;
        int rc = ensure(c, c->scratch, sizeof(R) * 2 * (size_t)(P.n + 1) * P.N * (size_t)grid);
        if (rc) return rc;
        hipLaunchKernelGGL(kg, dim3(grid), dim3(512), lds_g, c->stream, P, reinterpret_cast<R *>(c->scratch.p));
        HIP_TRY(c, hipGetLastError());
        return POLAR_OK;
    }
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    const int threads = std::max(64, std::min(512, P.N / 2));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds));
    if (occ < 1) occ = 1;
    int grid = std::min<long long>((long long)P.B, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::BpParams Q = P;
    if ((long long)P.B > (long long)grid) {
        int rc = work_queue(c, c->scratch, &Q.queue);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

template <typename R, typename IN>
int launch_bp_readout(polar_ctx *c, const polar::BpReadoutParams &P)
{
    auto kern = polar::k_bp_readout<R, IN>;
    const size_t lds = polar::bp_readout_lds_bytes<R>(P.N, P.n);
    if (lds > 160 * 1024) return POLAR_ENOKERNEL;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    const int threads = std::max(64, std::min(256, P.N / 2));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds));
    if (occ < 1) occ = 1;
    int grid = (int)std::min<long long>((long long)P.B, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, c->stream, P);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

}  // namespace

int polar_tu::bp(polar_ctx *c, const polar::BpParams &P, bool r32, bool in32)
{
    if (r32) return in32 ? launch_bp<float, float>(c, P) : launch_bp<float, double>(c, P);
    return in32 ? launch_bp<double, float>(c, P) : launch_bp<double, double>(c, P);
}

int polar_tu::bp_readout(polar_ctx *c, const polar::BpReadoutParams &P, bool r32, bool in32)
{
    if (r32) return in32 ? launch_bp_readout<float, float>(c, P) : launch_bp_readout<float, double>(c, P);
    return in32 ? launch_bp_readout<double, float>(c, P) : launch_bp_readout<double, double>(c, P);
}
