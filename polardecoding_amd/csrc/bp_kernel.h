// bp_kernel.h -- flooding belief-propagation decoder (reference: BP, BP_1024.c:372-427).
//
// One codeword per workgroup; one butterfly (two CHK) per thread and stage; all messages in LDS.
// The reference keeps l[0..n][N] and r[0..n][N].  Of those only l[1..n-1] and r[1..n-1] are ever
// re-read: l[n] is the channel LLR (constant), r[0] is the frozen prior 999/0 (BP_1024.c:386-391,
// taken from the mask), r[n] is written but never read, and l[0] is read only by the final hard
// decision (:417-425).  Dropping the dead rows leaves (2(n-1)+1) N reals = 152 KB for N = 1024 in
// f64, which is what lets a whole f64 codeword sit in one CU's 160 KB LDS.
// Stage order, operand order inside CHK and inside the sums are the reference's.
//
// Synchronisation: a wavefront's 64 butterflies of a stage touch 128 elements.  With butterfly b = 64w + x
// the stages 0..6 of wave w stay inside elements [128w, 128w + 128) ("map A"); for the stages 7..n-1 the wave
// takes instead the elements whose low 7 bits lie in a 2^(14-n)-wide window, all high bits ("map B"), which is
// closed under those stages.  Consecutive stages under the same map exchange data only inside one wave (LDS
// executes a wave's operations in order), so the workgroup barrier is needed only where the map changes:
// twice per iteration for N = 1024 instead of 18 times, and the waves of a codeword drift apart enough to
// overlap one wave's table reads with another's arithmetic.
#pragma once
#include "polar_math.h"
#include "polar_lut.h"
#include "polar_params.h"

namespace polar {

// The stages are latency-bound chains (2 waves per SIMD in f64): the one-round-trip CHK wins over the compact one.
#ifdef POLAR_BP_LUT2
#define BP_CHK chk_lut
#else
#define BP_CHK chk_lut1
#endif


// first element of butterfly bb in stage i (the partner is 2^i further)
__device__ __forceinline__ int bp_elem(int bb, int i, int n)
{
    if (i < 7) return ((bb >> i) << (i + 1)) | (bb & ((1 << i) - 1));
    const int lo_bits = 14 - n, t = i - 7;
    const int x = bb & 63, c = bb >> 6;
    const int h = x >> lo_bits;
    const int hi = ((h >> t) << (t + 1)) | (h & ((1 << t) - 1));
    return (hi << 7) | (c << lo_bits) | (x & ((1 << lo_bits) - 1));
}

// barrier between two consecutive stages: workgroup-wide where the element map changes, else wave-local
__device__ __forceinline__ void bp_sync(int stage_a, int stage_b)
{
    if ((stage_a < 7) != (stage_b < 7)) __syncthreads();
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

template <typename R, typename IN>
__global__ __launch_bounds__(512) void k_bp(BpParams P)
{
    const int N = P.N, n = P.n, NW = N >> 5;
    const int tid = threadIdx.x, nt = blockDim.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *ch = reinterpret_cast<R *>(smem);
    R *lm = ch + N;                       // rows 1..n-1 of l
    R *rm = lm + (size_t)(n - 1) * N;     // rows 1..n-1 of r
    uint32_t *obits = reinterpret_cast<uint32_t *>(rm + (size_t)(n - 1) * N);  // [NW]
    unsigned char *lut_mem = reinterpret_cast<unsigned char *>(obits + NW);
    lut_mem += (16 - (reinterpret_cast<uintptr_t>(lut_mem) & 15)) & 15;
    Lut<R>::build(lut_mem, tid, nt);
    Lut<R> lut;
    lut.bind(lut_mem);
    __syncthreads();

    __shared__ int job_slot;
    for (int frame = blockIdx.x; frame < P.B; frame = next_job_block(P.queue, frame, (int)gridDim.x, P.B, &job_slot)) {
        const IN *src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
        for (int i = tid; i < N; i += nt) {
            double v = (double)src[i];
            if (P.sigma > 0) v = llr_from_y(v, P.sigma);
            ch[i] = (R)v;  // BP_1024.c:381-382
        }
        for (int i = tid; i < (n - 1) * N; i += nt) {
            lm[i] = R(0);  // BP_1024.c:378-380
            rm[i] = R(0);  // BP_1024.c:384-386
        }
        for (int i = tid; i < NW; i += nt) obits[i] = 0;
        __syncthreads();

        for (int it = 0; it < P.iters; ++it) {
            const bool last = (it + 1 == P.iters);
            // R sweep (BP_1024.c:395-404); stage n-1 only produces r[n], which nobody reads
            for (int i = 0; i + 1 < n; ++i) {
                const int s = 1 << i;
                for (int b = tid; b < N / 2; b += nt) {
                    const int j = bp_elem(b, i, n);
                    R r0, r1;
                    if (i == 0) {
                        r0 = ((P.frozen[j >> 5] >> (j & 31)) & 1) ? R(999) : R(0);
                        r1 = ((P.frozen[(j + s) >> 5] >> ((j + s) & 31)) & 1) ? R(999) : R(0);
                    } else {
                        r0 = rm[(size_t)(i - 1) * N + j];
                        r1 = rm[(size_t)(i - 1) * N + j + s];
                    }
                    const R *lrow = (i + 1 == n) ? ch : lm + (size_t)i * N;
                    const R l0 = lrow[j], l1 = lrow[j + s];
                    const R a = BP_CHK<R>(r0, l1 + r1, lut);
                    const R c = r1 + BP_CHK<R>(r0, l0, lut);
                    rm[(size_t)i * N + j] = a;
                    rm[(size_t)i * N + j + s] = c;
                }
                bp_sync(i, i + 1);  // next: R stage i+1, or L stage n-1 = i+1
            }
            // L sweep (BP_1024.c:406-415); l[0] is needed only for the final decision
            for (int i = n - 1; i >= (last ? 0 : 1); --i) {
                const int s = 1 << i;
                for (int b = tid; b < N / 2; b += nt) {
                    const int j = bp_elem(b, i, n);
                    R r0, r1;
                    bool f0 = false, f1 = false;
                    if (i == 0) {
                        f0 = (P.frozen[j >> 5] >> (j & 31)) & 1;
                        f1 = (P.frozen[(j + s) >> 5] >> ((j + s) & 31)) & 1;
                        r0 = f0 ? R(999) : R(0);
                        r1 = f1 ? R(999) : R(0);
                    } else {
                        r0 = rm[(size_t)(i - 1) * N + j];
                        r1 = rm[(size_t)(i - 1) * N + j + s];
                    }
                    const R *lrow = (i + 1 == n) ? ch : lm + (size_t)i * N;
                    const R l0 = lrow[j], l1 = lrow[j + s];
                    const R a = BP_CHK<R>(l0, l1 + r1, lut);
                    const R c = l1 + BP_CHK<R>(r0, l0, lut);
                    if (i > 0) {
                        lm[(size_t)(i - 1) * N + j] = a;
                        lm[(size_t)(i - 1) * N + j + s] = c;
                    } else {
                        // BP_1024.c:417-425: frozen -> 0, else (l + r >= 0) -> 0
                        const uint32_t b0 = (!f0 && !(a + r0 >= R(0))) ? 1u : 0u;
                        const uint32_t b1 = (!f1 && !(c + r1 >= R(0))) ? 1u : 0u;
                        if (b0) atomicOr(&obits[j >> 5], 1u << (j & 31));
                        if (b1) atomicOr(&obits[(j + s) >> 5], 1u << ((j + s) & 31));
                    }
                }
                if (i > 0) bp_sync(i, i - 1);  // after L stage 1 comes R stage 0 of the next iteration
            }
        }
        __syncthreads();
        for (int i = tid; i < NW; i += nt) P.out_bits[(size_t)frame * NW + i] = obits[i];
        __syncthreads();
    }
}

template <typename R>
constexpr size_t bp_lds_bytes(int N, int n)
{
    return sizeof(R) * (size_t)N * (1 + 2 * (n - 1)) + sizeof(uint32_t) * (size_t)(N / 32) + 16 + Lut<R>::bytes;
}

// ---- BP with per-stage read-outs (reference: BPr, BPr_128.c:373-575) -------------------------
// The same flooding schedule with every row kept (the read-out of stage i needs l[i] + r[i] for i = 0..n, so
// r[n] and l[0] are computed in every iteration as the reference does), one codeword per workgroup, N/2 threads.
// After the iterations listed in cp[] the hard decisions of every stage are carried back to the u side through
// the inverse butterflies (:417-438) and compared with the sent bits on the information set.

template <typename R, typename IN>
__global__ __launch_bounds__(256) void k_bp_readout(BpReadoutParams P)
{
    const int N = P.N, n = P.n, NW = N >> 5;
    const int tid = threadIdx.x, nt = blockDim.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    R *lm = reinterpret_cast<R *>(smem);                 // l[0..n][N]
    R *rm = lm + (size_t)(n + 1) * N;                     // r[0..n][N]
    unsigned char *bb = reinterpret_cast<unsigned char *>(rm + (size_t)(n + 1) * N);   // [N] read-out bits
    uint32_t *obits = reinterpret_cast<uint32_t *>(bb + N);                            // [NW]
    uint32_t *cnt = obits + NW;                                                        // [1]
    unsigned char *lut_mem = reinterpret_cast<unsigned char *>(cnt + 4);
    lut_mem += (16 - (reinterpret_cast<uintptr_t>(lut_mem) & 15)) & 15;
    Lut<R>::build(lut_mem, tid, nt);
    Lut<R> lut;
    lut.bind(lut_mem);
    __syncthreads();
#define LM(i, j) lm[(size_t)(i) * N + (j)]
#define RM(i, j) rm[(size_t)(i) * N + (j)]
    for (int frame = blockIdx.x; frame < P.B; frame += gridDim.x) {
        const IN *src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
        for (int i = tid; i < (n + 1) * N; i += nt) {
            lm[i] = R(0);   // BPr_128.c:379-381
            rm[i] = R(0);   // :385-387
        }
        __syncthreads();
        for (int j = tid; j < N; j += nt) {
            double v = (double)src[j];
            if (P.sigma > 0) v = llr_from_y(v, P.sigma);
            LM(n, j) = (R)v;                                                        // :382-383
            RM(0, j) = ((P.frozen[j >> 5] >> (j & 31)) & 1) ? R(999) : R(0);        // :388-393
        }
        for (int i = tid; i < NW; i += nt) obits[i] = 0;
        __syncthreads();
        int q = 0;
        for (int it = 0; it < P.iters; ++it) {
            for (int i = 0; i < n; ++i) {   // R sweep (:396-405)
                const int s = 1 << i;
                for (int b = tid; b < N / 2; b += nt) {
                    const int j = ((b >> i) << (i + 1)) | (b & (s - 1));
                    const R r0 = RM(i, j), r1 = RM(i, j + s), l0 = LM(i + 1, j), l1 = LM(i + 1, j + s);
                    RM(i + 1, j) = chk_lut<R>(r0, l1 + r1, lut);
                    RM(i + 1, j + s) = r1 + chk_lut<R>(r0, l0, lut);
                }
                __syncthreads();
            }
            for (int i = n - 1; i >= 0; --i) {   // L sweep (:406-415)
                const int s = 1 << i;
                for (int b = tid; b < N / 2; b += nt) {
                    const int j = ((b >> i) << (i + 1)) | (b & (s - 1));
                    const R r0 = RM(i, j), r1 = RM(i, j + s), l0 = LM(i + 1, j), l1 = LM(i + 1, j + s);
                    LM(i, j) = chk_lut<R>(l0, l1 + r1, lut);
                    LM(i, j + s) = l1 + chk_lut<R>(r0, l0, lut);
                }
                __syncthreads();
            }
            if (q < P.ncp && P.cp[q] == it + 1) {   // read-out (:417-438)
                for (int i = 0; i <= n; ++i) {
                    if (tid == 0) cnt[0] = 0;
                    for (int j = tid; j < N; j += nt) bb[j] = (LM(i, j) + RM(i, j) >= R(0)) ? 0 : 1;
                    __syncthreads();
                    for (int k = i; k > 0; --k) {
                        const int s = 1 << (k - 1);
                        for (int b = tid; b < N / 2; b += nt) {
                            const int j = ((b >> (k - 1)) << k) | (b & (s - 1));
                            bb[j] ^= bb[j + s];
                        }
                        __syncthreads();
                    }
                    uint32_t e = 0;
                    for (int j = tid; j < N; j += nt) {
                        const uint32_t sent = (P.u_bits[(size_t)frame * NW + (j >> 5)] >> (j & 31)) & 1u;
                        const uint32_t isinfo = (P.info[j >> 5] >> (j & 31)) & 1u;
                        e += isinfo & (sent ^ (uint32_t)bb[j]);
                    }
                    if (e) atomicAdd(&cnt[0], e);
                    __syncthreads();
                    if (tid == 0 && cnt[0]) atomicAdd(&P.E[(size_t)q * (n + 1) + i], (unsigned long long)cnt[0]);
                    __syncthreads();
                }
                ++q;
            }
        }
        for (int j = tid; j < N; j += nt) {   // :566-574
            const bool fr = (P.frozen[j >> 5] >> (j & 31)) & 1;
            if (!fr && !(LM(0, j) + RM(0, j) >= R(0))) atomicOr(&obits[j >> 5], 1u << (j & 31));
        }
        __syncthreads();
        if (P.out_bits)
            for (int i = tid; i < NW; i += nt) P.out_bits[(size_t)frame * NW + i] = obits[i];
        __syncthreads();
    }
#undef LM
#undef RM
}

template <typename R>
constexpr size_t bp_readout_lds_bytes(int N, int n)
{
    return sizeof(R) * 2 * (size_t)(n + 1) * N + (size_t)N + 4 * (size_t)(N / 32) + 16 + 16 + Lut<R>::bytes;
}

// ---- BP for block lengths whose messages do not fit one CU's LDS (N > 1024): same schedule, rows in a per-workgroup
// slice of a global scratch buffer (plain stores, sc1 loads, a workgroup barrier after every stage).  Correct and
// complete rather than fast: no reference program and no BASELINE config uses BP above N = 1024.
template <typename R, typename IN>
__global__ __launch_bounds__(512) void k_bp_global(BpParams P, R *scratch)
{
    const int N = P.N, n = P.n, NW = N >> 5;
    const int tid = threadIdx.x, nt = blockDim.x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *obits = reinterpret_cast<uint32_t *>(smem);   // [NW]
    unsigned char *lut_mem = reinterpret_cast<unsigned char *>(obits + NW);
    lut_mem += (16 - (reinterpret_cast<uintptr_t>(lut_mem) & 15)) & 15;
    Lut<R>::build(lut_mem, tid, nt);
    Lut<R> lut;
    lut.bind(lut_mem);
    R *lm = scratch + (size_t)blockIdx.x * 2 * (size_t)(n + 1) * N;   // l[0..n][N]
    R *rm = lm + (size_t)(n + 1) * N;                                  // r[0..n][N]
    auto ld = [](const R *q) -> R { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    __syncthreads();
#define LM(i, j) lm[(size_t)(i) * N + (j)]
#define RM(i, j) rm[(size_t)(i) * N + (j)]
    for (int frame = blockIdx.x; frame < P.B; frame += gridDim.x) {
        const IN *src = reinterpret_cast<const IN *>(P.in) + (size_t)frame * N;
        for (int i = tid; i < (n + 1) * N; i += nt) {
            lm[i] = R(0);
            rm[i] = R(0);
        }
        __syncthreads();
        for (int j = tid; j < N; j += nt) {
            double v = (double)src[j];
            if (P.sigma > 0) v = llr_from_y(v, P.sigma);
            LM(n, j) = (R)v;
            RM(0, j) = ((P.frozen[j >> 5] >> (j & 31)) & 1) ? R(999) : R(0);
        }
        for (int i = tid; i < NW; i += nt) obits[i] = 0;
        __syncthreads();
        for (int it = 0; it < P.iters; ++it) {
            for (int i = 0; i < n; ++i) {
                const int s = 1 << i;
                for (int b = tid; b < N / 2; b += nt) {
                    const int j = ((b >> i) << (i + 1)) | (b & (s - 1));
                    const R r0 = ld(&RM(i, j)), r1 = ld(&RM(i, j + s)), l0 = ld(&LM(i + 1, j)), l1 = ld(&LM(i + 1, j + s));
                    RM(i + 1, j) = chk_lut<R>(r0, l1 + r1, lut);
                    RM(i + 1, j + s) = r1 + chk_lut<R>(r0, l0, lut);
                }
                __syncthreads();
            }
            for (int i = n - 1; i >= 0; --i) {
                const int s = 1 << i;
                for (int b = tid; b < N / 2; b += nt) {
                    const int j = ((b >> i) << (i + 1)) | (b & (s - 1));
                    const R r0 = ld(&RM(i, j)), r1 = ld(&RM(i, j + s)), l0 = ld(&LM(i + 1, j)), l1 = ld(&LM(i + 1, j + s));
                    LM(i, j) = chk_lut<R>(l0, l1 + r1, lut);
                    LM(i, j + s) = l1 + chk_lut<R>(r0, l0, lut);
                }
                __syncthreads();
            }
        }
        for (int j = tid; j < N; j += nt) {
            const bool fr = (P.frozen[j >> 5] >> (j & 31)) & 1;
            if (!fr && !(ld(&LM(0, j)) + ld(&RM(0, j)) >= R(0))) atomicOr(&obits[j >> 5], 1u << (j & 31));
        }
        __syncthreads();
        for (int i = tid; i < NW; i += nt) P.out_bits[(size_t)frame * NW + i] = obits[i];
        __syncthreads();
    }
#undef LM
#undef RM
}

}  // namespace polar
