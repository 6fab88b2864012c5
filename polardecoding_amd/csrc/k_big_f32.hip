// k_big_f32.hip -- k_scl_big<float, ...> and its launch code
#include "polar_host.h"
#include "scl_big.h"

#ifndef POLAR_BIG_CH_MINN
#define POLAR_BIG_CH_MINN 2048   // shortest code that runs the chain() kernel (N = 1024: see DESIGN.md 4.2)
#endif

namespace {

// big lists / long codes: low LLR levels in LDS, the rest in a per-wave scratch slice (scl_big.h)
template <typename R, typename IN, int LOGL, int TL, int TB, int RL = 0, int CH = 0>
int launch_big_v(polar_ctx *c, const polar::SclParams &P)
{
    using Cfg = polar::BigCfg<R, LOGL, TL, TB, RL>;
    auto kern = polar::k_scl_big<R, IN, LOGL, TL, TB, RL, CH>;
    const size_t lds = Cfg::lds_bytes;
    const int threads = 64 * Cfg::WAVES;
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    int occ = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds));
    if (occ < 1) occ = 1;
    const long long blocks_needed = ((long long)P.B + Cfg::WAVES - 1) / Cfg::WAVES;
    int grid = std::min<long long>(blocks_needed, (long long)occ * c->num_cu);
    if (grid < 1) grid = 1;
    polar::SclParams Q = P;
    int rc = ensure(c, c->scratch, Cfg::scratch_bytes(P.N) * (size_t)grid * Cfg::WAVES);
    if (rc) return rc;
    Q.scratch = c->scratch.p;
    if ((long long)P.B > (long long)grid * Cfg::WAVES && (rc = work_queue(c, c->scratch, &Q.queue))) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, c->stream, Q);
    HIP_TRY(c, hipGetLastError());
    return POLAR_OK;
}

// the LDS / scratch split that measured best per arithmetic type (profiles/README.md)
template <typename R, typename IN, int LOGL>
int launch_big(polar_ctx *c, const polar::SclParams &P)
{
    const int use = c->big_split ? c->big_split : (sizeof(R) == 8 ? (LOGL == 5 ? 371 : 35) : 46);
    // long codes, L = 32: the structure of the f64 kernel of BASELINE config 5 (chain() of the upper levels, split 4 / 7 / 1,
    // three wavefronts per SIMD: k_big_f64.hip) -- f32, N = 4096: 0.391 -> 0.453 M frames/s, same decisions as the 4 / 6 kernel
    // on the whole 2^15-frame batch (four wavefronts per SIMD spill 16 VGPRs: 0.427 M)
    if constexpr (LOGL == 5) {
        if (!c->big_split && P.N >= POLAR_BIG_CH_MINN) return launch_big_v<R, IN, LOGL, 4, 7, 1, 1>(c, P);
    }
    if constexpr (LOGL == 5) {   // L = 32: LLR level TL+1 in registers (third digit of the split code; two such levels, and
                                 // one above four LDS levels, measured slower: fewer resident wavefronts).  At the four
                                 // wavefronts per SIMD of that kernel the LDS has room for partial-sum levels 6 and 7 too
                                 // (371: two scratch round trips less per 128 leaves, +3 %)
        if (use == 351) return launch_big_v<R, IN, LOGL, 3, 5, 1>(c, P);
        if (use == 371) {
            // long codes in f64 (BASELINE config 5, N = 4096): the f chains of the upper levels in one pass, three
            // wavefronts per SIMD (scl_big.h, chain()); N = 1024 keeps the four-wavefront kernel
            if constexpr (sizeof(R) == 8) {
                if (P.N >= POLAR_BIG_CH_MINN) return launch_big_v<R, IN, LOGL, 3, 7, 1, 1>(c, P);
            }
            return launch_big_v<R, IN, LOGL, 3, 7, 1>(c, P);
        }
    }
    if (use == 57) return launch_big_v<R, IN, LOGL, 5, 7>(c, P);
    if (use == 46) return launch_big_v<R, IN, LOGL, 4, 6>(c, P);
    return launch_big_v<R, IN, LOGL, 3, 5>(c, P);
}

template <typename R, typename IN>
int launch_big_l(polar_ctx *c, const polar::SclParams &P)
{
    switch (c->logL) {
    case 1: return launch_big<R, IN, 1>(c, P);
    case 2: return launch_big<R, IN, 2>(c, P);
    case 3: return launch_big<R, IN, 3>(c, P);
    case 4: return launch_big<R, IN, 4>(c, P);
    case 5: return launch_big<R, IN, 5>(c, P);
    }
    return POLAR_ENOKERNEL;
}

}  // namespace

int polar_tu::scl_big_f32(polar_ctx *c, const polar::SclParams &P, bool in32)
{
    return in32 ? launch_big_l<float, float>(c, P) : launch_big_l<float, double>(c, P);
}
