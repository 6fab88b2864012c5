/*
 * polar_hip_testing.h -- TEST-ONLY entry points of libpolar_hip.so.
 *
 * Not part of the drop-in boundary (include/polar_hip.h): nothing here corresponds to a reference interface.
 * The parity tests use these to run the SAME inputs through every kernel that can decode a configuration
 * (the tuned kernel polar_create picks, and the slower ones it would pick for other shapes) and to put single
 * operands through the kernels' check-node / metric arithmetic.  The product never calls them; the library reads
 * no environment variable.
 */
#ifndef POLAR_HIP_TESTING_H
#define POLAR_HIP_TESTING_H

#include "polar_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* which kernel a list / SC context launches */
#define POLAR_TEST_KERNEL_AUTO 0           /* polar_create's choice (default)                                      */
#define POLAR_TEST_KERNEL_GENERIC 1        /* k_scl_generic, every LLR level in LDS (if it fits, else as 2)          */
#define POLAR_TEST_KERNEL_GENERIC_SPILL 2  /* k_scl_generic, every LLR level in global scratch                       */
#define POLAR_TEST_KERNEL_BIG 3            /* k_scl_big (N >= 512, L >= 2) also where a tuned L = 8 kernel exists    */
#define POLAR_TEST_KERNEL_ONE_PER_WAVE 4   /* N = 1024, L = 8: k_scl_fast (one codeword per wavefront), not the pair */
#define POLAR_TEST_KERNEL_FOUR_PER_WAVE 5  /* N = 1024, L = 8: k_scl_fast4 (four codewords per wavefront)            */
int polar_testing_select_kernel(polar_ctx *ctx, int variant);

/* k_scl_big: LLR levels <= TL and partial-sum levels <= TB in LDS, written as the two digits TL TB:
 * 35, 46 or 57; 351 / 371 = 35 / 37 with LLR level 4 in the registers of the path's own lanes (L = 32 only; elsewhere 35);
 * 0 = the measured best for the arithmetic type and list size. */
int polar_testing_big_split(polar_ctx *ctx, int split);

/* The kernels' scalar arithmetic on caller-chosen operands, one thread per element, through the SAME device
 * functions the decoders inline (csrc/polar_math.h, csrc/polar_lut.h):
 *   op 0  chk(a, b)        compare chain, CHK of SCL_1024.c:343-374
 *   op 1  chk_lut(a, b)    table form (one compare per look-up)
 *   op 2  chk_lut1(a, b)   table form, one LDS round trip
 *   op 3  tabv(|a|)        T of SCL_1024.c:352-359 from the table      (b ignored)
 *   op 4  phi(a, u = (b != 0))      compare chain, PHI of SCL_1024.c:481-502
 *   op 5  phi_lut(a, u = (b != 0))  the form the list kernels inline: tabv(a) + max(+-a, 0)
 *   op 6  chk_cnt(a, b)    staircase counted on the VALU from the signs of |x| - threshold
 *   op 7  chk_idx(a, b)    prefix popcount over 26 cells + one threshold read (the BP kernel)
 *   op 8  chk_tab(a, b)    the same with the prefix count read from a byte table (measured alternative, not used)
 * is_f32 = 0: a, b, out are double[n]; 1: float[n].  Host pointers. */
int polar_testing_math(int op, int is_f32, const void *a, const void *b, void *out, size_t n, int device);

#ifdef __cplusplus
}
#endif
#endif
