/*
 * polar_hip.h -- C ABI of libpolar_hip.so: MI355X-native (gfx950) polar decoders.
 *
 * Drop-in boundary for the per-frame decode functions of CHEBSB/PolarDecoding.  The reference has no
 * plugin/FFI layer; its boundary is a C function symbol plus file-scope globals:
 *
 *     void SCdecode (double *y, int *u_hat);   SC_128.c:78, :395     (N,K,n #define; std, inI[] global)
 *     void BP       (double *y, int *u_hat);   BP_1024.c:114, :372   (+ iterMax)
 *     void SCLdecode(double *y, int *u_hat);   SCL_1024.c:134, :547  (+ L; PM, PMcand, surviv global)
 *     void CASCL    (double *y, int *u_hat);   CASCL_1024_L8.c:141, :601 (+ r, CRC taps inline)
 *
 * Every entry point below names the reference interface it replaces.  Plain pointers and sizes only;
 * no C++ or torch types.  All functions return 0 on success or a negative POLAR_E* code; nothing is
 * printed (the reference's printf diagnostics "Oops!" / "Wrong propagation order!" / "Error!",
 * SCL_1024.c:418, :622, :651, become the per-frame flags word).
 *
 * Thread-safety: a polar_ctx is bound to one GPU and one HIP stream and is not re-entrant (neither is
 * the reference: all its scratch is global).  Use one ctx per host thread / per GPU.
 */
#ifndef POLAR_HIP_H
#define POLAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* algo */
#define POLAR_ALGO_SC 0    /* SCdecode   SC_128.c:395-460                      */
#define POLAR_ALGO_BP 1    /* BP         BP_1024.c:372-427                     */
#define POLAR_ALGO_SCL 2   /* SCLdecode  SCL_1024.c:547-680                    */
#define POLAR_ALGO_CASCL 3 /* CASCL      CASCL_1024_L8.c:601-761               */

/* dtype: the arithmetic type the message passing runs in */
#define POLAR_F64 0 /* IEEE binary64 like the reference: bit-identical decisions (the parity gate) */
#define POLAR_F32 1 /* binary32: same operation order, FER-equivalent, not bit-identical           */

/* error codes */
#define POLAR_OK 0
#define POLAR_EINVAL (-1)   /* bad argument / unsupported configuration        */
#define POLAR_ENOMEM (-2)   /* host or device allocation failed                */
#define POLAR_EDEVICE (-3)  /* HIP runtime error (see polar_last_error)        */
#define POLAR_ENOKERNEL (-4)/* no kernel instantiation for this (N, L, dtype)  */

/* per-frame flags word */
#define POLAR_FLAG_TIE 0x1u      /* a median tie occurred (reference prints "Oops!", SCL_1024.c:621-622) */
#define POLAR_FLAG_CRC_PASS 0x2u /* CASCL: the chosen path passed the CRC (CASCL_1024_L8.c:738-746)       */
#define POLAR_FLAG_RERANK 0x4u   /* diagnostic, no reference counterpart: at some information leaf the high 32 bits of
                                    the 2L candidate metrics did not single out L survivors and the kernel ranked on
                                    the full doubles (f64 list kernels that pre-rank on 32-bit keys; same result either
                                    way -- the bit lets the tests see that rare path run)                              */

typedef struct polar_ctx polar_ctx;

/* Replaces the reference's compile-time configuration (#define N K n L r iterMax, CASCL_1024_L8.c:16-21;
 * the Q-table derived I[] / inI[], :209-217; the CRC taps written inline at :253-265 and :581-593). */
typedef struct polar_cfg {
    int N;                 /* block length, power of two, 32..4096 (BP above 1024: the messages no longer fit one
                              CU's LDS and live in global scratch -- complete, not fast)                     */
    int K;                 /* payload bits                                                               */
    int crc_r;             /* CRC length r (0 = none)                                                    */
    const int *crc_taps;   /* exponents of g(D) incl. 0 and r, e.g. {0,5,6} (CASCL_128.c:212-214)        */
    int n_taps;
    int L;                 /* list size, power of two 1..32 (SC: 1)                                      */
    int algo;              /* POLAR_ALGO_*                                                               */
    int bp_iters;          /* BP round trips (reference: iterMax = 100, BP_1024.c:16)                    */
    const int *info_order; /* I[0..K+crc_r): unfrozen positions in reliability order (I[i] = Q[N-(K+r)+i]).
                              NULL -> built from the 5G sequence like the reference does.                 */
    int dtype;             /* POLAR_F64 | POLAR_F32                                                      */
    int device;            /* HIP device ordinal                                                         */
    int crc_systematic;    /* 0: the CRC word is v(D) g(D) (CASCL_1024_L8.c:251-266).  1: CASCL_1024_sys.c:
                              systematic cyclic encoding, w[0..r) = D^r v(D) mod g, w[r..K+r) = v (:776-789), and
                              the error counters look at the K payload bits only (:820-821).  The decoder is the
                              same: that program's bit-reversed graph fed with y[bRev[j]] makes the decisions of
                              the natural-order graph fed with y.  Used by polar_generate_device,
                              polar_count_errors_device and polar_fer_batch; ignored without a CRC.            */
} polar_cfg;

int polar_create(const polar_cfg *cfg, polar_ctx **out);
void polar_destroy(polar_ctx *ctx);
const char *polar_strerror(int code);
/* text of the last HIP error seen by this ctx ("" if none) */
const char *polar_last_error(const polar_ctx *ctx);

/* --- CRC generator-matrix file ------------------------------------------------------------------------
 * The reference's CRC_6.dat (K = 64 rows x r = 6 entries, 0/1; row i = D^(r+i) mod g(D), column j = coefficient of
 * D^j; UTF-16LE with BOM, CRLF, single spaces, no final newline) and, in the same layout, the literal
 * `const int Gc[K][r]` of CASCL_1024_sys.c:48-561 that its systematic encoder sums at :776-789.  The loader takes
 * the file's encoding as it is (also UTF-8 / ASCII, also with the braces and commas of the C literal), derives
 * g(D) = D^r + row 0 and REJECTS the file (POLAR_EINVAL) unless every row i equals D^(r+i) mod g, g has a D^0
 * term, all rows have the same 1..32 entries and every entry is 0 or 1. */
typedef struct polar_crc_matrix {
    int K;            /* rows = payload bits the matrix serves                                     */
    int r;            /* columns = CRC length                                                      */
    int n_taps;       /* number of exponents of g(D)                                               */
    int taps[33];     /* exponents of g(D), ascending, incl. 0 and r: what polar_cfg.crc_taps wants */
    uint32_t *rows;   /* [K] bit j of rows[i] = entry (i, j); malloc'ed, release with ..._free     */
} polar_crc_matrix;
int polar_crc_matrix_load(const char *path, polar_crc_matrix *out);
void polar_crc_matrix_free(polar_crc_matrix *m);
/* writes the K x r matrix of g(D) in the reference's encoding, byte for byte what CRC_6.dat holds for K = 64, {0,5,6} */
int polar_crc_matrix_save(const char *path, int K, const int *taps, int n_taps);
/* polar_create with r and g(D) taken from such a file (cfg->crc_r / crc_taps / n_taps are ignored; cfg->algo must be
 * POLAR_ALGO_CASCL; cfg->K <= the file's row count).  cfg->crc_systematic = 1 is the encoder the matrix belongs to. */
int polar_create_crc_file(const polar_cfg *cfg, const char *path, polar_ctx **out);

/* --- reference-shaped single-frame call -------------------------------------------------------------
 * Identical result to  std = sigma; X(y, u_hat);  for X = SCdecode / BP / SCLdecode / CASCL
 * (SCL_1024.c:263: the per-frame call in main()).  y: N channel observations (not LLRs: the LLR
 * 2*y/std/std is formed inside, SCL_1024.c:574-578).  u_hat: N ints, fully overwritten, frozen = 0. */
int polar_decode(polar_ctx *ctx, const double *y, double sigma, int *u_hat);

/* --- north-star call shape: decode(llr_in, frozen_mask, N, L) ----------------------------------------
 * One frame of channel LLRs and a frozen mask (1 = frozen; the reference's !inI[], SCL_1024.c:198-206).
 * Plain SCL (L > 1) or SC (L == 1) in f64.  Contexts are cached per (N, L, mask). */
int polar_decode_llr(const double *llr_in, const unsigned char *frozen_mask, int N, int L, int *u_hat);

/* --- batched, host buffers ---------------------------------------------------------------------------
 * B frames.  llr_in [B][N] row-major channel LLRs (double).  frozen_mask: NULL (use cfg) or [N] override
 * for SC/BP/SCL (CASCL needs cfg's info_order for the CRC).  u_hat [B][N] ints 0/1.
 * pm_out (nullable) [B]: metric of the chosen path (SCL/CASCL; 0 otherwise).  flags (nullable) [B]. */
int polar_decode_batch(polar_ctx *ctx, const double *llr_in, const unsigned char *frozen_mask, size_t B,
                       int *u_hat, double *pm_out, unsigned *flags);

/* Same, but the input is channel observations y and the kernel forms 2*y/sigma/sigma on load
 * (SCL_1024.c:576, operation order kept). */
int polar_decode_batch_y(polar_ctx *ctx, const double *y, double sigma, size_t B, int *u_hat,
                         double *pm_out, unsigned *flags);

/* --- batched, device buffers (the measured path; asynchronous on the ctx stream) ---------------------
 * d_in: [B][N] of double (in_is_f32 = 0) or float (in_is_f32 = 1), LLRs, or y when sigma > 0.
 * d_uhat_bits: [B][N/32] uint32, bit (j & 31) of word j >> 5 = u_hat[j].
 * d_pm (nullable): [B] double.  d_flags (nullable): [B] uint32. */
int polar_decode_device(polar_ctx *ctx, const void *d_in, int in_is_f32, double sigma, size_t B,
                        uint32_t *d_uhat_bits, double *d_pm, uint32_t *d_flags);

/* --- error accounting on device (main()'s compare loop, CASCL_1024_L8.c:296-305) ---------------------
 * d_u_bits: [B][N/32] transmitted u.  Adds to d_counters[0] (block errors) and d_counters[1] (bit errors on
 * the K + r unfrozen positions).  d_frame_err (nullable): [B] uint32 bit errors per frame (needed to cut at
 * the reference's sequential stop rule). */
int polar_count_errors_device(polar_ctx *ctx, const uint32_t *d_uhat_bits, const uint32_t *d_u_bits, size_t B,
                              unsigned long long *d_counters, uint32_t *d_frame_err);

/* --- the sequential stop rule on a batch (`for (run = 0; errBlock < BLE; run++)`, SCL_1024.c:228, :264-275) -------
 * The reference ends an Eb/N0 point WITH the frame that brings the block errors to BLE; generator state and PN phase
 * carry on from there.  For a batch decoded as a whole that is a prefix count over polar_count_errors_device's
 * d_frame_err, done on the device: d_out[0] = frames consumed (position of the `need`-th erroneous frame + 1, or B if
 * the batch holds fewer), d_out[1] / d_out[2] = block / bit errors among the consumed frames.
 * min_frames: the rule behind the published L = 32 logs (myResult_1024.zip:CASCL_L32.dat, "error block = 487 run =
 * 2000"), `errBlock < BLE || run < 2000`: consume at least min_frames frames (0 = the plain rule); with
 * min_frames > 0, need may be 0 (BLE already reached, the minimum not yet). */
int polar_stop_rule_cut_device(polar_ctx *ctx, const uint32_t *d_frame_err, size_t B, unsigned need, size_t min_frames,
                               unsigned long long *d_out /* [3] */);

/* Host-buffer form, one iteration of main()'s loop over a batch: y [B][N] observations, sigma = std, u_bits [B][N/32]
 * the sent u packed like the decisions.  Decode, compare on the unfrozen positions (:266-272; payload only with
 * crc_systematic) and cut, all on the device; only three numbers come back. */
int polar_stop_rule_batch_y(polar_ctx *ctx, const double *y, double sigma, const uint32_t *u_bits, size_t B,
                            unsigned need, size_t min_frames, size_t *consumed, unsigned long long *block_errors,
                            unsigned long long *bit_errors);

/* --- BP with per-stage read-outs (BPr_128.c:373-575: `BPr(y, u_hat, u)` and its table E[7][n+1]) -----------------
 * ctx must be a POLAR_ALGO_BP context (its bp_iters = the program's iterMax, 90 in BPr_128.c:16).  After each of the
 * iteration counts checkpoints[0..ncp) (ascending, <= 8; the program uses 3, 6, 10, 20, 40, 80, :18-23) the hard
 * decisions of l + r at every stage i = 0..n are carried back to the u side and compared with the sent bits on the
 * information set; d_E[c*(n+1) + i] accumulates the mismatches over the B frames (the program's E[c][i], :437).
 * d_u_bits: [B][N/32] sent bits; d_uhat_bits (may be NULL): [B][N/32] final decisions.  N <= 512 (f64), 1024 (f32). */
int polar_bp_readout_device(polar_ctx *ctx, const void *d_in, int in_is_f32, double sigma, size_t B,
                            const uint32_t *d_u_bits, const int *checkpoints, int ncp, unsigned long long *d_E,
                            uint32_t *d_uhat_bits);
/* Host-buffer form, the shape of BPr_128.c's frame loop (:213): y [B][N] observations (sigma > 0) or LLRs (sigma = 0),
 * u [B][N] sent bits (0/1 ints), E [ncp][n+1] accumulated (+=), u_hat [B][N] (may be NULL). */
int polar_bp_readout_batch(polar_ctx *ctx, const double *in, double sigma, size_t B, const int *u,
                           const int *checkpoints, int ncp, unsigned long long *E, int *u_hat);

/* --- device-side transmit chain, throughput mode (the frame loop of main(), CASCL_1024_L8.c:245-292) -----------
 * Fills B frames: random payload -> CRC multiply by g(D) -> u[I[i]] -> x = u F^{(x)n} -> BPSK + AWGN at
 * Eb/N0 = snr_db (sigma = 10^(-snr_db/20), rate 1/2 as in the reference, :237) -> d_out[B][N] (double, or float
 * when out_is_f32; channel LLRs 2y/sigma/sigma, or y when out_is_y) and d_u_bits[B][N/32] (nullable).
 * Counter-based generator: frame (first_frame + f) depends only on (seed, frame index), so batches can be cut
 * and sharded over GPUs freely.  This is NOT the reference's sequential Ranq1 stream (that one stays on the
 * host: polar_sim.c); use it for throughput / FER runs, not for reproducing published run counts. */
int polar_generate_device(polar_ctx *ctx, unsigned long long seed, unsigned long long first_frame, double snr_db,
                          size_t B, void *d_out, int out_is_f32, int out_is_y, uint32_t *d_u_bits);

/* One throughput-mode Monte-Carlo batch entirely on the device: generate (as polar_generate_device) -> decode ->
 * count (main()'s loop body, CASCL_1024_L8.c:245-305, B times).  Adds the batch's block and bit errors to
 * *block_errors / *bit_errors (host counters).  Buffers are owned and reused by the ctx. */
int polar_fer_batch(polar_ctx *ctx, unsigned long long seed, unsigned long long first_frame, double snr_db, size_t B,
                    unsigned long long *block_errors, unsigned long long *bit_errors);

/* The same over the GPUs of one node (frames are independent, so the range first_frame .. first_frame +
 * ngpus * frames_per_gpu is cut into contiguous shards, one per GPU, each decoded by its own context on its own host thread;
 * no data-path collective).  The only exchange is the sum of the two counters, one ncclAllReduce of 2 x uint64 over RCCL /
 * xGMI per call; RCCL is loaded with dlopen() on first use (POLAR_EDEVICE if the machine has none or fewer than ngpus
 * devices).  Frame f of the range is the same frame whatever ngpus is (counter-based generator): the totals equal
 * polar_fer_batch over the whole range on one GPU.  seconds (nullable): the slowest GPU's time for its shard.
 * A polar_group holds the ngpus contexts (devices 0 .. ngpus-1; cfg->device is ignored) and the RCCL communicators, so that a
 * sweep pays for their creation once; polar_fer_multi_gpu is create + one batch + destroy. */
typedef struct polar_group polar_group;
int polar_group_create(const polar_cfg *cfg, int ngpus, polar_group **out);
void polar_group_destroy(polar_group *grp);
int polar_group_size(const polar_group *grp);
int polar_group_fer_batch(polar_group *grp, unsigned long long seed, unsigned long long first_frame, double snr_db,
                          size_t frames_per_gpu, unsigned long long *block_errors, unsigned long long *bit_errors,
                          double *seconds);
/* main()'s stop rule `for (run = 0; errBlock < BLE; run++)` (SCL_1024.c:228) over a batch decoded in shards: generate ->
 * decode -> count on every GPU as above, then ONE ncclAllGather of the per-frame error counts (4 bytes per frame) into frame
 * order and the cut of polar_stop_rule_cut_device on GPU 0.  *frames_used = frames consumed up to and including the one that
 * brings the block errors to `need` (all ngpus * frames_per_gpu if the batch holds fewer; at least min_frames),
 * *block_errors / *bit_errors = the counters over those frames (set, not added).  Independent of ngpus by construction.
 * polar_group_create runs the two collectives on known values first and refuses the group if RCCL answers differently. */
int polar_group_stop_rule_batch(polar_group *grp, unsigned long long seed, unsigned long long first_frame, double snr_db,
                                size_t frames_per_gpu, unsigned need, size_t min_frames, size_t *frames_used,
                                unsigned long long *block_errors, unsigned long long *bit_errors);
int polar_fer_multi_gpu(const polar_cfg *cfg, int ngpus, unsigned long long seed, unsigned long long first_frame,
                        double snr_db, size_t frames_per_gpu, unsigned long long *block_errors,
                        unsigned long long *bit_errors, double *seconds);

/* stream plumbing: the ctx owns a stream by default; a host framework may hand in its own
 * (hipStream_t passed as void*). */
int polar_set_stream(polar_ctx *ctx, void *hip_stream);
void *polar_get_stream(polar_ctx *ctx);
int polar_synchronize(polar_ctx *ctx);

/* Time `reps` launches of the decode kernel on B resident frames with HIP events on the ctx stream
 * (bench.py's roofline leg).  Returns average milliseconds per launch in *ms_per_launch. */
int polar_time_decode_device(polar_ctx *ctx, const void *d_in, int in_is_f32, double sigma, size_t B,
                             uint32_t *d_uhat_bits, int reps, float *ms_per_launch);

/* introspection */
/* I[0..A): the unfrozen positions in reliability order this ctx uses (the reference's global I[],
 * CASCL_1024_L8.c:99, :214-217); n must be >= A. */
int polar_info_order(const polar_ctx *ctx, int *out, int n);
int polar_ctx_info(const polar_ctx *ctx, int *N, int *K, int *A, int *L, int *algo, int *dtype);
/* name of the kernel instantiation this ctx launches (for matching rocprofv3 output) */
const char *polar_kernel_name(const polar_ctx *ctx);
/* library build id string */
const char *polar_version(void);

#ifdef __cplusplus
}
#endif
#endif
