// tools/ubench.hip -- instruction-issue micro-benchmarks for gfx950, used to choose the CHK / PHI /
// sort formulations (DESIGN.md "What the VALU budget looks like").  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip && tools/ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e = (x);                                                        \
        if (e != hipSuccess) {                                                     \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int ITER = 20000;

// Each kernel: 8 independent instances of OP per iteration, ITER iterations, timed with s_memtime.
#define KERNEL(name, setup, body)                                                  \
    __global__ __launch_bounds__(64) void name(unsigned long long *out, unsigned long long execmask) \
    {                                                                              \
        __shared__ double lds[1024];                                               \
        lds[threadIdx.x] = threadIdx.x;                                            \
        double d0 = threadIdx.x * 0.5, d1 = 1.5, d2 = 2.5, d3 = 3.5, d4 = 4.5, d5 = 5.5, d6 = 6.5, d7 = 7.5; \
        float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7; \
        unsigned u0 = threadIdx.x * 4, u1 = 1, u2 = 2, u3 = 3, u4 = 4, u5 = 5, u6 = 6, u7 = 7; \
        unsigned long long q0 = threadIdx.x, q1 = 77;                              \
        int s0 = 0;                                                                \
        setup;                                                                     \
        asm volatile("s_mov_b64 exec, %0" ::"s"(execmask));                        \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                      \
        for (int i = 0; i < ITER; ++i) {                                           \
            body;                                                                  \
        }                                                                          \
        asm volatile("s_waitcnt lgkmcnt(0)");                                      \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                      \
        asm volatile("s_mov_b64 exec, -1");                                        \
        double acc = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + u0 + u1 + \
                     u2 + u3 + u4 + u5 + u6 + u7 + (double)q0 + (double)q1 + s0;   \
        if (acc == 12345.678) out[1] = 1;                                          \
        if (threadIdx.x == 0) out[2 + blockIdx.x] = t1 - t0;                       \
    }

KERNEL(k_add_f64, ,
       asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                    "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                    : "v"(1.0)))
KERNEL(k_add_f64_dep, ,
       asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                    "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                    : "+v"(d0)
                    : "v"(1.0)))
KERNEL(k_min_f64, ,
       asm volatile("v_min_f64 %0, %0, %8\n v_min_f64 %1, %1, %8\n v_min_f64 %2, %2, %8\n v_min_f64 %3, %3, %8\n"
                    "v_min_f64 %4, %4, %8\n v_min_f64 %5, %5, %8\n v_min_f64 %6, %6, %8\n v_min_f64 %7, %7, %8\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                    : "v"(1.0)))
KERNEL(k_add_f32, ,
       asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                    "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
                    : "v"(1.0f)))
KERNEL(k_add_f32_dep, ,
       asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                    "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                    : "+v"(f0)
                    : "v"(1.0f)))
KERNEL(k_add_u32, ,
       asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                    "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                    : "v"(3u)))
// compare + carry-accumulate (the T-index idiom): 4 pairs per iteration
KERNEL(k_cmp_f64_addc, ,
       asm volatile("v_cmp_lt_f64 vcc, %4, %0\n v_addc_co_u32 %5, vcc, %5, 0, vcc\n"
                    "v_cmp_lt_f64 vcc, %4, %1\n v_addc_co_u32 %6, vcc, %6, 0, vcc\n"
                    "v_cmp_lt_f64 vcc, %4, %2\n v_addc_co_u32 %7, vcc, %7, 0, vcc\n"
                    "v_cmp_lt_f64 vcc, %4, %3\n v_addc_co_u32 %8, vcc, %8, 0, vcc\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
                    :
                    : "vcc"))
KERNEL(k_cmp_f64, ,
       asm volatile("v_cmp_lt_f64 vcc, %4, %0\n v_cmp_lt_f64 vcc, %4, %1\n v_cmp_lt_f64 vcc, %4, %2\n"
                    "v_cmp_lt_f64 vcc, %4, %3\n v_cmp_lt_f64 vcc, %4, %0\n v_cmp_lt_f64 vcc, %4, %1\n"
                    "v_cmp_lt_f64 vcc, %4, %2\n v_cmp_lt_f64 vcc, %4, %3\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4)
                    :
                    : "vcc"))
KERNEL(k_cmp_u64, ,
       asm volatile("v_cmp_lt_u64 vcc, %4, %0\n v_cmp_lt_u64 vcc, %4, %1\n v_cmp_lt_u64 vcc, %4, %2\n"
                    "v_cmp_lt_u64 vcc, %4, %3\n v_cmp_lt_u64 vcc, %4, %0\n v_cmp_lt_u64 vcc, %4, %1\n"
                    "v_cmp_lt_u64 vcc, %4, %2\n v_cmp_lt_u64 vcc, %4, %3\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4)
                    :
                    : "vcc"))
KERNEL(k_cmp_f32_addc, ,
       asm volatile("v_cmp_lt_f32 vcc, %4, %0\n v_addc_co_u32 %5, vcc, %5, 0, vcc\n"
                    "v_cmp_lt_f32 vcc, %4, %1\n v_addc_co_u32 %6, vcc, %6, 0, vcc\n"
                    "v_cmp_lt_f32 vcc, %4, %2\n v_addc_co_u32 %7, vcc, %7, 0, vcc\n"
                    "v_cmp_lt_f32 vcc, %4, %3\n v_addc_co_u32 %8, vcc, %8, 0, vcc\n"
                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
                    :
                    : "vcc"))
KERNEL(k_cndmask, ,
       asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n"
                    "v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                    "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                    : "v"(3u)
                    : "vcc"))
KERNEL(k_bpermute, ,
       asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n"
                    "ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n"
                    "ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                    : "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7), "+v"(f0)
                    : "v"(u0)))
KERNEL(k_dpp_mov, ,
       asm volatile("v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %1, %2 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %2, %3 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %3, %4 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %4, %5 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %5, %6 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %6, %7 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    "v_mov_b32_dpp %7, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)))
KERNEL(k_readlane, ,
       asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %8, %1, 5\n v_readlane_b32 %8, %2, 7\n"
                    "v_readlane_b32 %8, %3, 9\n v_readlane_b32 %8, %4, 11\n v_readlane_b32 %8, %5, 13\n"
                    "v_readlane_b32 %8, %6, 15\n v_readlane_b32 %8, %7, 17\n"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7), "+s"(s0)))
KERNEL(k_ds_read_b64, u0 = threadIdx.x * 8,
       asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:512\n ds_read_b64 %2, %8 offset:1024\n"
                    "ds_read_b64 %3, %8 offset:1536\n ds_read_b64 %4, %8 offset:2048\n ds_read_b64 %5, %8 offset:2560\n"
                    "ds_read_b64 %6, %8 offset:3072\n ds_read_b64 %7, %8 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                    : "v"(u0)))
KERNEL(k_ds_write_b64, u0 = threadIdx.x * 8,
       asm volatile("ds_write_b64 %8, %0\n ds_write_b64 %8, %1 offset:512\n ds_write_b64 %8, %2 offset:1024\n"
                    "ds_write_b64 %8, %3 offset:1536\n ds_write_b64 %8, %4 offset:2048\n ds_write_b64 %8, %5 offset:2560\n"
                    "ds_write_b64 %8, %6 offset:3072\n ds_write_b64 %8, %7 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                    : "v"(u0)))


KERNEL(k_cndmask_sgpr, ,
       asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n"
                    "v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n"
                    "v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n"
                    "v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                    : "v"(3u)
                    : "s10", "s11"))
KERNEL(k_cmp_f32, ,
       asm volatile("v_cmp_lt_f32 vcc, %4, %0\n v_cmp_lt_f32 vcc, %4, %1\n v_cmp_lt_f32 vcc, %4, %2\n"
                    "v_cmp_lt_f32 vcc, %4, %3\n v_cmp_lt_f32 vcc, %4, %0\n v_cmp_lt_f32 vcc, %4, %1\n"
                    "v_cmp_lt_f32 vcc, %4, %2\n v_cmp_lt_f32 vcc, %4, %3\n"
                    : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4)
                    :
                    : "vcc"))
// compare into distinct SGPR pairs then select: the tab() idiom the compiler emits
KERNEL(k_cmp_f64_cnd, ,
       asm volatile("v_cmp_lt_f64 s[10:11], %4, %0\n v_cndmask_b32_e64 %5, %5, %9, s[10:11]\n"
                    "v_cmp_lt_f64 s[12:13], %4, %1\n v_cndmask_b32_e64 %6, %6, %9, s[12:13]\n"
                    "v_cmp_lt_f64 s[14:15], %4, %2\n v_cndmask_b32_e64 %7, %7, %9, s[14:15]\n"
                    "v_cmp_lt_f64 s[16:17], %4, %3\n v_cndmask_b32_e64 %8, %8, %9, s[16:17]\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
                    : "v"(3u)
                    : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17"))
// 4 compares first, then 4 selects (software-pipelined distance)
KERNEL(k_cmp_f64_cnd_far, ,
       asm volatile("v_cmp_lt_f64 s[10:11], %4, %0\n v_cmp_lt_f64 s[12:13], %4, %1\n"
                    "v_cmp_lt_f64 s[14:15], %4, %2\n v_cmp_lt_f64 s[16:17], %4, %3\n"
                    "v_cndmask_b32_e64 %5, %5, %9, s[10:11]\n v_cndmask_b32_e64 %6, %6, %9, s[12:13]\n"
                    "v_cndmask_b32_e64 %7, %7, %9, s[14:15]\n v_cndmask_b32_e64 %8, %8, %9, s[16:17]\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
                    : "v"(3u)
                    : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17"))
KERNEL(k_cmp_u32_addc, ,
       asm volatile("v_cmp_lt_u32 vcc, %4, %0\n v_addc_co_u32 %5, vcc, %5, 0, vcc\n"
                    "v_cmp_lt_u32 vcc, %4, %1\n v_addc_co_u32 %6, vcc, %6, 0, vcc\n"
                    "v_cmp_lt_u32 vcc, %4, %2\n v_addc_co_u32 %7, vcc, %7, 0, vcc\n"
                    "v_cmp_lt_u32 vcc, %4, %3\n v_addc_co_u32 %8, vcc, %8, 0, vcc\n"
                    : "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7), "+v"(f0), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3)
                    :
                    : "vcc"))
KERNEL(k_fma_f64, ,
       asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n"
                    "v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n"
                    "v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                    : "v"(1.0)))
KERNEL(k_xor_b32, ,
       asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                    "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                    : "v"(3u)))
KERNEL(k_lshl_b64, ,
       asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n"
                    "v_lshlrev_b64 %3, 3, %3\n v_lshlrev_b64 %4, 3, %4\n v_lshlrev_b64 %5, 3, %5\n"
                    "v_lshlrev_b64 %6, 3, %6\n v_lshlrev_b64 %7, 3, %7\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)))
KERNEL(k_swizzle, ,
       asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n ds_swizzle_b32 %1, %1 offset:0x041F\n"
                    "ds_swizzle_b32 %2, %2 offset:0x041F\n ds_swizzle_b32 %3, %3 offset:0x041F\n"
                    "ds_swizzle_b32 %4, %4 offset:0x041F\n ds_swizzle_b32 %5, %5 offset:0x041F\n"
                    "ds_swizzle_b32 %6, %6 offset:0x041F\n ds_swizzle_b32 %7, %7 offset:0x041F\n s_waitcnt lgkmcnt(0)\n"
                    : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)))
// table look-up in LDS with lane-dependent pseudo-random 64-entry index (bank conflicts as they come)
KERNEL(k_lds_lut64, u0 = ((threadIdx.x * 37 + 11) & 63) * 8,
       asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n"
                    "ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                    : "v"(u0)))
KERNEL(k_lds_lut_rand, u0 = ((threadIdx.x * threadIdx.x * 2654435761u) >> 26) * 8,
       asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8\n ds_read_b64 %2, %8\n ds_read_b64 %3, %8\n"
                    "ds_read_b64 %4, %8\n ds_read_b64 %5, %8\n ds_read_b64 %6, %8\n ds_read_b64 %7, %8\n"
                    "s_waitcnt lgkmcnt(0)\n"
                    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)
                    : "v"(u0)))

typedef void (*kfn)(unsigned long long *, unsigned long long);

static void run(const char *name, kfn k, int waves_per_simd, unsigned long long execmask, int ops_per_iter,
                unsigned long long *d_out, int ncu)
{
    const int blocks = ncu * 4 * waves_per_simd;
    std::vector<unsigned long long> h(2 + blocks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, execmask);  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, execmask);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < blocks; ++i) avg += (double)h[2 + i];
    avg /= blocks;
    // s_memtime ticks: assume 100 MHz constant clock if tiny, else shader clock; print both views
    const double per_op_ticks = avg / ((double)ITER * ops_per_iter);
    const double wall_ns_per_op_per_simd = (double)ms * 1e6 / ((double)ITER * ops_per_iter * waves_per_simd);
    printf("%-18s w/simd=%d exec=%016llx  ticks/op/wave=%7.3f  wall ns/op/SIMD=%7.3f  (kernel %.3f ms)\n", name,
           waves_per_simd, execmask, per_op_ticks, wall_ns_per_op_per_simd, ms);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, ncu, prop.clockRate);
    unsigned long long *d_out;
    CK(hipMalloc(&d_out, (2 + ncu * 4 * 8) * 8));
    const unsigned long long FULL = ~0ull;
    struct {
        const char *n;
        kfn k;
        int ops;
    } ks[] = {{"add_f64", k_add_f64, 8},       {"add_f64_dep", k_add_f64_dep, 8}, {"min_f64", k_min_f64, 8},
              {"add_f32", k_add_f32, 8},       {"add_f32_dep", k_add_f32_dep, 8}, {"add_u32", k_add_u32, 8},
              {"cmp_f64+addc", k_cmp_f64_addc, 8}, {"cmp_f64", k_cmp_f64, 8},     {"cmp_u64", k_cmp_u64, 8},
              {"cmp_f32+addc", k_cmp_f32_addc, 8}, {"cndmask", k_cndmask, 8},     {"bpermute", k_bpermute, 8},
              {"dpp_mov", k_dpp_mov, 8},       {"readlane", k_readlane, 8},       {"ds_read_b64", k_ds_read_b64, 8},
              {"ds_write_b64", k_ds_write_b64, 8},
              {"cndmask_sgpr", k_cndmask_sgpr, 8}, {"cmp_f32", k_cmp_f32, 8}, {"cmp_f64;cnd", k_cmp_f64_cnd, 8},
              {"cmp_f64x4;cndx4", k_cmp_f64_cnd_far, 8}, {"cmp_u32+addc", k_cmp_u32_addc, 8},
              {"fma_f64", k_fma_f64, 8}, {"xor_b32", k_xor_b32, 8}, {"lshl_b64", k_lshl_b64, 8},
              {"swizzle", k_swizzle, 8}, {"lds_lut64", k_lds_lut64, 8}, {"lds_lut_rand", k_lds_lut_rand, 8}};
    for (auto &k : ks)
        for (int w : {1, 2, 4}) run(k.n, k.k, w, FULL, k.ops, d_out, ncu);
    printf("--- partial EXEC (does the SIMD skip idle lane groups?) ---\n");
    for (unsigned long long m : {0x1ull, 0xFull, 0xFFull, 0xFFFull, 0xFFFFull, 0xFFFFFFFFull, 0x0101010101010101ull,
                                 0x0303030303030303ull, 0x0F0F0F0F0F0F0F0Full, 0x00FF00FF00FF00FFull,
                                 0x000F000F000F000Full, 0x0001000100010001ull, 0xFF000000000000FFull, FULL}) {
        run("add_f64", k_add_f64, 2, m, 8, d_out, ncu);
        run("add_f32", k_add_f32, 2, m, 8, d_out, ncu);
        run("cmp_f64+addc", k_cmp_f64_addc, 2, m, 8, d_out, ncu);
    }
    return 0;
}
