// gen_kernel.h -- device-side transmit chain: payload -> CRC multiply by g(D) -> u[I[i]] = w[i]
// -> x = u F^{(x)n} -> BPSK + AWGN -> channel LLR, one frame per wavefront, for throughput-mode Monte-Carlo.
// Restates the SHAPE of main()'s frame loop (CASCL_1024_L8.c:245-292); it is NOT the reference's sequential
// generator: the payload and the noise come from a counter-based generator (Philox4x32-10, keyed by the seed,
// counter = global frame index + element), so that a frame depends only on (seed, frame index) and batches can
// be cut and sharded freely.  Bit-exact reproduction of the reference's Ranq1 / Marsaglia stream stays on the
// host (host/polar_sim.c), because that stream cannot be indexed per frame.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace polar {

struct GenParams {
    void *out;               // [B][N] double or float: LLR (2y/s/s) or y
    uint32_t *u_bits;        // [B][N/32] transmitted u, or null
    const int *info_order;   // [A]
    uint64_t seed, first_frame;
    double sigma;
    uint32_t crc_mask;       // bit t set <=> D^t in g(D), t < 32 (taps 0..r); 1 when no CRC
    uint32_t crc_top;        // tap r when r == 32 handled via crc_r
    int crc_r;
    const uint32_t *gc_rows; // systematic CRC (CASCL_1024_sys.c:48-561): row k = D^(r+k) mod g as an r-bit mask; else null
    int N, n, K, A, B;
    int out_is_f32, out_is_y;
};

struct Philox {
    uint32_t c[4];
    __device__ __forceinline__ static void round_(uint32_t *c, uint32_t k0, uint32_t k1)
    {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ __forceinline__ Philox(uint64_t seed, uint64_t frame, uint32_t block, uint32_t stream)
    {
        c[0] = (uint32_t)frame; c[1] = (uint32_t)(frame >> 32); c[2] = block; c[3] = stream;
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            round_(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
    }
    // two uniforms in (0,1) with 53 random bits each
    __device__ __forceinline__ double u0() const { return ((double)((((uint64_t)c[0] << 32) | c[1]) >> 11) + 0.5) * 0x1.0p-53; }
    __device__ __forceinline__ double u1() const { return ((double)((((uint64_t)c[2] << 32) | c[3]) >> 11) + 0.5) * 0x1.0p-53; }
};

// lane l holds codeword / u bits j = l + 64 k as bit k of a 16-bit (N = 1024) .. 64-bit (N = 4096) word
__global__ __launch_bounds__(256) void k_generate(GenParams P)
{
    const int N = P.N, NW = N >> 5, KR = N >> 6;  // KR = bits per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    extern __shared__ unsigned char gsm[];
    unsigned char *ub = gsm + (size_t)wave * (N + 2 * 1024);       // u bytes [N]
    uint32_t *vw = reinterpret_cast<uint32_t *>(ub + N);           // payload words [K/32 + 2]
    const int waves = blockDim.x >> 6;
    for (int f = blockIdx.x * waves + wave; f < P.B; f += gridDim.x * waves) {
        const uint64_t frame = P.first_frame + (uint64_t)f;
        // payload: K random bits
        const int kw = (P.K + 31) >> 5;
        for (int w = lane; w < kw + 2; w += 64) {
            uint32_t v = 0;
            if (w < kw) {
                v = Philox(P.seed, frame, (uint32_t)w, 0u).c[0];
                if (w == kw - 1 && (P.K & 31)) v &= (1u << (P.K & 31)) - 1u;
            }
            vw[w] = v;
        }
        for (int j = lane; j < N; j += 64) ub[j] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (P.gc_rows) {
            // systematic CRC (CASCL_1024_sys.c:776-789): redundant part = sum of the generator rows of the set
            // payload bits, then the payload itself; placement u[I[i]] = w[i]
            uint32_t par = 0;
            for (int k = lane; k < P.K; k += 64)
                if ((vw[k >> 5] >> (k & 31)) & 1u) par ^= P.gc_rows[k];
            for (int o = 32; o > 0; o >>= 1) par ^= __shfl_xor(par, o);
            for (int i = lane; i < P.A; i += 64) {
                const int q = i - P.crc_r;
                const uint32_t bit = (q < 0) ? ((par >> i) & 1u) : ((vw[q >> 5] >> (q & 31)) & 1u);
                ub[P.info_order[i]] = (unsigned char)bit;
            }
        } else {
        // CRC multiply w(D) = v(D) g(D) (CASCL_1024_L8.c:251-266) and placement u[I[i]] = w[i] (:270-272)
        for (int i = lane; i < P.A; i += 64) {
            uint32_t bit = 0;
            for (int t = 0; t <= P.crc_r; ++t) {
                const bool tap = (t < 32) ? ((P.crc_mask >> t) & 1u) : (P.crc_top != 0);
                const int q = i - t;
                if (tap && q >= 0 && q < P.K) bit ^= (vw[q >> 5] >> (q & 31)) & 1u;
            }
            ub[P.info_order[i]] = (unsigned char)bit;
        }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint64_t u = 0;
        for (int k = 0; k < KR; ++k) u |= (uint64_t)(ub[lane + 64 * k] & 1) << k;
        if (P.u_bits) {
            for (int k = 0; k < KR; ++k) {
                const uint64_t m = __ballot((u >> k) & 1ull);  // bits j = 64k .. 64k+63
                if (lane == 0) {
                    P.u_bits[(size_t)f * NW + 2 * k] = (uint32_t)m;
                    P.u_bits[(size_t)f * NW + 2 * k + 1] = (uint32_t)(m >> 32);
                }
            }
        }
        // x = u F^{(x)n} (SCL_1024.c:242-250): strides < 64 across lanes, strides >= 64 inside the lane word
        uint64_t x = u;
        for (int s = 0; s < 6 && s < P.n; ++s) {
            const uint64_t o = __shfl_xor((unsigned long long)x, 1 << s);
            if (!(lane & (1 << s))) x ^= o;
        }
        for (int s = 6; s < P.n; ++s) {
            const int sh = 1 << (s - 6);
            uint64_t msk = 0;
            for (int k = 0; k < KR; ++k)
                if (!(k & sh)) msk |= 1ull << k;
            x ^= (x >> sh) & msk;
        }
        // channel: y = (1 - 2x) + sigma n, n from Box-Muller on Philox uniforms; LLR = 2 y / sigma / sigma
        for (int k2 = 0; k2 < KR; k2 += 2) {
            const Philox g(P.seed, frame, (uint32_t)(lane + 64 * (k2 >> 1)), 1u);
            const double r = sqrt(-2.0 * log(g.u0()));
            double sn, cs;
            sincospi(2.0 * g.u1(), &sn, &cs);
            const double nz[2] = {r * cs, r * sn};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = k2 + h;
                if (k >= KR) break;   // N = 64: one element per lane, the second normal of the pair is not used
                const int j = lane + 64 * k;
                const double y = (((x >> k) & 1ull) ? -1.0 : 1.0) + P.sigma * nz[h];
                const double v = P.out_is_y ? y : 2 * y / P.sigma / P.sigma;
                if (P.out_is_f32) reinterpret_cast<float *>(P.out)[(size_t)f * N + j] = (float)v;
                else reinterpret_cast<double *>(P.out)[(size_t)f * N + j] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace polar
