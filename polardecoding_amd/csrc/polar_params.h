// polar_params.h -- kernel argument blocks shared by the kernel translation units and the host layer.
#pragma once
#include <stdint.h>

namespace polar {

struct SclParams {
    const void *in;            // [B][N] double or float: LLRs, or y when sigma > 0
    double sigma;              // > 0: input is y, llr = 2*y/sigma/sigma
    uint32_t *out_bits;        // [B][N/32]
    double *pm;                // [B] or null
    uint32_t *flags;           // [B] or null
    const uint32_t *frozen;    // [N/32] bit j = leaf j frozen
    const uint32_t *crc_tab;   // [N] D^{pos(j)} mod g for unfrozen leaf j (0 for frozen); null = no CRC
    int N, n;
    int B;
    int sc_mode;               // 1: plain SC decisions (SCdecode), L must be 1
    void *scratch;             // k_scl_fast, N = 1024: per-wave global scratch (FastCfg::scratch_elems each)
    unsigned long long *dbg;   // diagnostic builds only (-DPOLAR_STAMPS): per-section cycle sums
    unsigned *queue;           // persistent kernels: job counter (polar_host.h work_queue()); null = jobs by a fixed stride
};

struct BpParams {
    const void *in;          // [B][N] double or float (LLR, or y when sigma > 0)
    double sigma;
    uint32_t *out_bits;      // [B][N/32]
    const uint32_t *frozen;  // [N/32]
    int N, n, B, iters;
    unsigned *queue;         // as in SclParams
};

// BP with per-stage read-outs (reference: BPr, BPr_128.c:373-575), see bp_kernel.h
struct BpReadoutParams {
    const void *in;           // [B][N] LLR, or y when sigma > 0
    double sigma;
    uint32_t *out_bits;       // [B][N/32] final decisions (may be null)
    const uint32_t *frozen;   // [N/32]
    const uint32_t *info;     // [N/32] 1 = information position
    const uint32_t *u_bits;   // [B][N/32] sent bits
    unsigned long long *E;    // [ncp][n+1], accumulated over the frames of the launch
    int cp[8];                // iteration counts (1-based), ascending
    int ncp;
    int N, n, B, iters;
};

#ifdef __HIPCC__
// ---- work queue of the persistent kernels (host side: polar_host.h work_queue()) ----
// The first `resident` jobs are taken by index (wavefront or workgroup number), every further one from a counter in device
// memory: the returned job number is >= `resident`.  The counter hands out 0, 1, 2, ...; the values below total - resident
// are jobs, and every one of the `resident` takers ends on exactly one value at or above it, so the taker that receives
// total - 1 knows it is the last to ask and puts the counter back to zero: the next launch (or a replay of this one from
// a captured graph) finds it as this one did, without a memset in between.  queue == null: fixed stride.
__device__ __forceinline__ unsigned job_fetch(unsigned *queue, int total)
{
    const unsigned nx = atomicAdd(queue, 1u);
    if (nx == (unsigned)(total - 1)) atomicExch(queue, 0u);
    return nx;
}
__device__ __forceinline__ int next_job_wave(unsigned *queue, int cur, int resident, int total)
{
    if (!queue) return cur + resident;
    unsigned nx = 0;
    if ((threadIdx.x & 63) == 0) nx = job_fetch(queue, total);
    return resident + (int)__builtin_amdgcn_readfirstlane(nx);
}
// one job per workgroup: `slot` is one LDS word nobody else uses; every thread of the workgroup calls this
__device__ __forceinline__ int next_job_block(unsigned *queue, int cur, int resident, int total, int *slot)
{
    if (!queue) return cur + resident;
    if (threadIdx.x == 0) *slot = (int)job_fetch(queue, total);
    __syncthreads();
    const int nx = *slot;
    __syncthreads();
    return resident + nx;
}
#endif


}  // namespace polar
