// count_kernel.h -- main()'s compare loop and stop rule on the device (k_count_errors, k_stop_cut).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace polar {

// ---- error accounting (main()'s compare loop, CASCL_1024_L8.c:296-305) -------------------------------
// One thread per frame word would be enough; one wave per frame keeps it trivially coalesced.
struct CountParams {
    const uint32_t *uhat;    // [B][NW]
    const uint32_t *u;       // [B][NW]
    const uint32_t *info;    // [NW] 1 = unfrozen position
    unsigned long long *counters;  // [2] block errors, bit errors
    uint32_t *frame_err;     // [B] or null
    int NW, B;
};

__global__ __launch_bounds__(256) void k_count_errors(CountParams P)
{
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    unsigned long long blk = 0, bits = 0;
    for (int f = wave; f < P.B; f += nwaves) {
        int e = 0;
        for (int w = lane; w < P.NW; w += 64)
            e += __popc((P.uhat[(size_t)f * P.NW + w] ^ P.u[(size_t)f * P.NW + w]) & P.info[w]);
        for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
        if (lane == 0) {
            if (P.frame_err) P.frame_err[f] = (uint32_t)e;
            bits += (unsigned long long)e;
            blk += (e != 0);
        }
    }
    if (lane == 0 && (blk | bits)) {
        atomicAdd(&P.counters[0], blk);
        atomicAdd(&P.counters[1], bits);
    }
}

// ---- the reference's sequential stop rule on a batch -------------------------------------------------
// main() decodes frame after frame `for (run = 0; errBlock < BLE; run++)` (SCL_1024.c:228, counters :264-275):
// the point ends WITH the frame that brings the block errors to BLE.  On a batch that was decoded as a whole this
// is a prefix count over the per-frame error counts k_count_errors wrote: out[0] = frames consumed (index of the
// `need`-th erroneous frame + 1, or B if the batch does not contain that many), out[1] / out[2] = block / bit
// errors among the consumed frames.  One workgroup: B words are read twice, which is nothing next to the decode.
// min_frames: the variant behind the published L = 32 logs (myResult_1024.zip:CASCL_L32.dat: "run = 2000" with 487
// block errors), `errBlock < BLE || run < 2000`: at least that many frames are consumed; need may then be 0.
__global__ __launch_bounds__(1024) void k_stop_cut(const uint32_t *frame_err, int B, unsigned need, int min_frames,
                                                   unsigned long long *out)
{
    __shared__ unsigned wcnt[16];
    __shared__ unsigned long long wbits[16];
    __shared__ unsigned base_s;
    __shared__ int cut_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { base_s = 0; cut_s = need ? B : 0; }
    __syncthreads();
    for (int s0 = 0; need && s0 < B; s0 += 1024) {   // pass 1: where is the need-th erroneous frame?
        const int f = s0 + tid;
        const bool bad = f < B && frame_err[f] != 0;
        const unsigned long long m = __ballot(bad);
        if (lane == 0) wcnt[w] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = base_s;
        for (int q = 0; q < w; ++q) before += wcnt[q];
        const unsigned rank = before + (unsigned)__popcll(m & ((1ull << lane) - 1ull)) + 1u;   // 1-based, if bad
        if (bad && rank == need) cut_s = f + 1;
        __syncthreads();
        if (tid == 0) {
            unsigned t = base_s;
            for (int q = 0; q < 16; ++q) t += wcnt[q];
            base_s = t;
        }
        __syncthreads();
        if (cut_s != B || base_s >= need) break;   // uniform: both are shared and were written before the barrier
    }
    const bool found = need == 0 || base_s >= need;
    const int cut = found ? min(B, max(cut_s, min_frames)) : B;
    unsigned blk = 0;
    unsigned long long bits = 0;
    for (int f = tid; f < cut; f += 1024) {   // pass 2: the counters over the consumed frames
        const uint32_t e = frame_err[f];
        blk += (e != 0);
        bits += e;
    }
    for (int o = 32; o > 0; o >>= 1) {
        blk += __shfl_down(blk, o);
        bits += __shfl_down(bits, o);
    }
    __syncthreads();
    if (lane == 0) { wcnt[w] = blk; wbits[w] = bits; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long tb = 0, tbits = 0;
        for (int q = 0; q < 16; ++q) { tb += wcnt[q]; tbits += wbits[q]; }
        out[0] = (unsigned long long)cut;
        out[1] = tb;
        out[2] = tbits;
    }
}

}  // namespace polar
