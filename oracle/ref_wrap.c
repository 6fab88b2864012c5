/*
 * ref_wrap.c -- builds ONE reference simulator (CHEBSB/PolarDecoding, e.g. CASCL_1024_L8.c) from its
 * source WHERE IT LIES under /root/reference, unmodified, into oracle/_ref/ as a shared library and a
 * small executable.  TEST INFRASTRUCTURE ONLY.  No reference text is copied into this repository: the
 * source file is #included by path (-DREF_SRC="..."), with its main() renamed so that the real
 * decode function (SCdecode / BP / SCLdecode / CASCL) can be called frame by frame.
 *
 *   -DREF_SRC="\"/root/reference/CASCL_1024_L8.c\""   the translation unit to wrap
 *   -DREF_DECODE=CASCL                                  its decode entry point
 *   -DREF_KIND=3                                        0 SC, 1 BP, 2 SCL, 3 CASCL, 4 BP with per-stage read-outs (BPr_128.c)
 *   -DREF_BITREV                                        the program decodes on the bit-reversed graph (CASCL_1024_sys.c)
 *   -DREF_DROPIN -DREF_TAPS=0,5,6                       drop-in demonstration (oracle/Makefile target `dropin`): REF_SRC is the
 *                                                       program streamed through sed with the ONE decode call of its main()
 *                                                       reading ref_dropin(y, u_hat) -- the binding of INTEGRATION.md 2 --
 *                                                       so the reference's own main() runs on libpolar_hip.so
 *
 * What is restated here (because the reference keeps it inline in main(), SCL_1024.c:159-217) is
 * only the graph/frozen-set SET-UP, done with the reference's own connectBCB() and Q table.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* time(NULL) seeds the CASCL and BP programs (CASCL_1024_L8.c:164): make it controllable. */
static unsigned long long ref_time_value = 0;
/* Fn is read from stdin (SCL_1024.c:207-217) and is not shipped: synthesise F^{(x)n} instead. */
static long ref_scan_pos = 0;
static int ref_scan_dim = 0;
static int ref_scanf_int(int *dst)
{
    long i = ref_scan_pos / ref_scan_dim, j = ref_scan_pos % ref_scan_dim;
    *dst = ((i & j) == j) ? 1 : 0;
    ref_scan_pos++;
    return 1;
}
/* The decoders report trouble on stdout ("Oops!" on a median tie SCL_1024.c:622, "Wrong propagation order!"
 * :418, "Error!" :651): count those lines, and keep them off stdout while ref_decode() runs.  Everything else
 * (the result lines of main()) is printed as written. */
static long ref_diag_count[3];
static int ref_diag_quiet = 0;
static int ref_printf(const char *fmt, ...)
{
    static const char *const what[3] = {"Oops!", "Wrong propagation order!", "Error!"};
    int a, rv;
    va_list ap;
    for (a = 0; a < 3; a++)
        if (strncmp(fmt, what[a], strlen(what[a])) == 0) {
            ref_diag_count[a]++;
            if (ref_diag_quiet) return 0;
        }
    va_start(ap, fmt);
    rv = vprintf(fmt, ap);
    va_end(ap);
#ifdef REF_BUILD_EXE
    {   /* test harness only: REF_STOP_AFTER_LINES=k ends the program (cleanly, between two frames) once main() has printed k
         * lines -- the tail of a sweep can be hundreds of thousands of frames that a test does not need */
        static long lines = 0, limit = -1;
        if (limit < 0) {
            const char *e = getenv("REF_STOP_AFTER_LINES");
            limit = e ? atol(e) : 0;
        }
        if (strchr(fmt, '\n') && limit > 0 && ++lines >= limit) {
            fflush(stdout);
            exit(0);
        }
    }
#endif
    return rv;
}
#ifdef REF_DROPIN
#include "polar_hip.h"
static void ref_dropin(double *y, int *u_hat);   /* defined below, once the program's N, K, L, r, I[] and std are known */
#endif
#define printf ref_printf
#define time(x) ((time_t)ref_time_value)
#define scanf(fmt, p) ref_scanf_int(p)
#define main ref_main
#include REF_SRC
#undef main
#undef scanf
#undef time
#undef printf

/* From here on N, K, n (and L, r for the list decoders) are the reference's macros and `std` its
 * global noise deviation: no local identifier below may use those names. */

enum { REF_BLOCK = N, REF_INFO = K, REF_LOG = n };
#if REF_KIND == 2 || REF_KIND == 3
enum { REF_LIST = L };
#else
enum { REF_LIST = 1 };
#endif
#if REF_KIND == 3
enum { REF_CRC = r };
#else
enum { REF_CRC = 0 };
#endif

static int ref_ready = 0;

#ifdef REF_DROPIN
/* What INTEGRATION.md 2 tells a maintainer to add: a context made from the program's own constants and its I[] (filled by
 * main() before the first frame), and polar_decode() in place of the decode call. */
static void ref_dropin(double *y, int *u_hat)
{
    static polar_ctx *ctx = NULL;
    if (!ctx) {
#ifdef REF_TAPS
        static const int taps[] = {REF_TAPS};
#endif
        /* positional initialiser: the program's N, K, L, r are macros here, so the fields cannot be named
         * { N, K, crc_r, crc_taps, n_taps, L, algo, bp_iters, info_order, dtype, device, crc_systematic } */
        polar_cfg cfg = {REF_BLOCK, REF_INFO, REF_CRC,
#ifdef REF_TAPS
                         taps, (int)(sizeof taps / sizeof taps[0]),
#else
                         NULL, 0,
#endif
                         REF_LIST,
                         REF_KIND == 0 ? POLAR_ALGO_SC : REF_KIND == 1 ? POLAR_ALGO_BP : REF_KIND == 2 ? POLAR_ALGO_SCL : POLAR_ALGO_CASCL,
#if REF_KIND == 1
                         iterMax,
#else
                         0,
#endif
                         I, POLAR_F64, 0, 0};
        int rc;
        rc = polar_create(&cfg, &ctx);
        if (rc) { fprintf(stderr, "polar_create: %s\n", polar_strerror(rc)); exit(1); }
    }
    if (polar_decode(ctx, y, std, u_hat) != POLAR_OK) { fprintf(stderr, "polar_decode: %s\n", polar_last_error(ctx)); exit(1); }
}
#endif

int ref_block_length(void) { return REF_BLOCK; }
int ref_info_bits(void) { return REF_INFO; }
int ref_crc_bits(void) { return REF_CRC; }
int ref_list_size(void) { return REF_LIST; }
int ref_kind(void) { return REF_KIND; }

/* Graph + information-set set-up of main() (SCL_1024.c:143-206), using the reference's globals and
 * its connectBCB(); idempotent. */
int ref_init(void)
{
    int a, b;
    if (ref_ready) return 0;
#if REF_KIND == 2 || REF_KIND == 3
    PM = (double *)calloc(2 * REF_LIST, sizeof(double));
    PMcand = (double *)calloc(2 * REF_LIST, sizeof(double));
#endif
    V = (node ***)calloc(REF_LOG + 1, sizeof(node **));
    for (a = 0; a <= REF_LOG; a++) {
        V[a] = (node **)calloc(REF_BLOCK, sizeof(node *));
        for (b = 0; b < REF_BLOCK; b++) V[a][b] = (node *)calloc(1, sizeof(node));
    }
    for (a = 0; a <= REF_LOG; a++)
        for (b = 0; b < REF_BLOCK; b++) initV[a][b] = 0;
    for (b = 0; b < REF_BLOCK; b++) {
        V[0][b]->pU = NULL;
        V[0][b]->pL = NULL;
        connectBCB(0, b);
    }
    for (a = 1; a < REF_LOG; a++)
        for (b = 0; b < REF_BLOCK; b++) connectBCB(a, b);
    for (b = 0; b < REF_BLOCK; b++) {
        V[REF_LOG][b]->cU = NULL;
        V[REF_LOG][b]->cL = NULL;
    }
#ifdef REF_BITREV
    /* CASCL_1024_sys.c:726-735 builds bRev[] inline in main(): index with its REF_LOG bits reversed */
    for (a = 0; a < REF_BLOCK; a++) {
        int rev = 0;
        for (b = 0; b < REF_LOG; b++)
            if ((a >> b) & 1) rev |= 1 << (REF_LOG - 1 - b);
        bRev[a] = rev;
    }
#endif
    for (a = 0; a < REF_BLOCK; a++) inI[a] = 0;
    for (a = 0; a < REF_INFO + REF_CRC; a++) {
        I[a] = Q[REF_BLOCK - (REF_INFO + REF_CRC) + a];
        inI[I[a]] = 1;
    }
    ref_ready = 1;
    return 0;
}

/* One call of the reference decoder: y = channel observations, sigma -> global `std`. */
#if REF_KIND == 4
/* BPr_128.c: BPr(y, u_hat, u) also takes the sent bits and accumulates the per-stage read-outs in E[7][n+1] */
int ref_decode_u(const double *y, double sigma, const int *u, int *u_hat)
{
    if (!ref_ready) ref_init();
    std = sigma;
    REF_DECODE((double *)y, u_hat, (int *)u);
    return 0;
}
int ref_decode(const double *y, double sigma, int *u_hat) { return -1; }
void ref_readout(int *out, int reset)
{
    int a, b;
    for (a = 0; a < 7; a++)
        for (b = 0; b <= REF_LOG; b++) {
            out[a * (REF_LOG + 1) + b] = E[a][b];
            if (reset) E[a][b] = 0;
        }
}
int ref_readout_iters(int *cp) { cp[0] = i0; cp[1] = i1; cp[2] = i2; cp[3] = i3; cp[4] = i4; cp[5] = i5; return iterMax; }
#else
int ref_decode(const double *y, double sigma, int *u_hat)
{
    if (!ref_ready) ref_init();
    std = sigma;
    ref_diag_quiet = 1;
    REF_DECODE((double *)y, u_hat);
    ref_diag_quiet = 0;
    return 0;
}
#endif

/* counts of the three diagnostics since the last reset: out[0] "Oops!", out[1] "Wrong propagation order!", out[2] "Error!" */
void ref_diag(long *out, int reset)
{
    int a;
    for (a = 0; a < 3; a++) {
        out[a] = ref_diag_count[a];
        if (reset) ref_diag_count[a] = 0;
    }
}

#if REF_KIND == 2 || REF_KIND == 3
/* Put the node records back to what calloc() gave main() (SCL_1024.c:159-164): the list decoders never clear
 * paths 1..L-1, so what they do after a median tie depends on what earlier frames left there. */
void ref_reset_state(void)
{
    int a, b;
    if (!ref_ready) ref_init();
    for (a = 0; a <= REF_LOG; a++)
        for (b = 0; b < REF_BLOCK; b++) {
            memset(V[a][b]->l, 0, sizeof V[a][b]->l);
            memset(V[a][b]->b, 0, sizeof V[a][b]->b);
            memset(V[a][b]->lDone, 0, sizeof V[a][b]->lDone);
            memset(V[a][b]->bDone, 0, sizeof V[a][b]->bDone);
        }
    memset(PM, 0, 2 * REF_LIST * sizeof(double));
    memset(PMcand, 0, 2 * REF_LIST * sizeof(double));
    memset(surviv, 0, sizeof surviv);
}
void ref_path_metrics(double *out)
{
    int a;
    for (a = 0; a < REF_LIST; a++) out[a] = PM[a];
}
#endif

/* Path metric of the path the list decoders just chose (same selection rule as the decoder). */
double ref_last_pm(void)
{
#if REF_KIND == 2
    int a, best = 0;
    for (a = 1; a < REF_LIST; a++)
        if (PM[a] < PM[best]) best = a;
    return PM[best];
#elif REF_KIND == 3
    int a, best = -1;
    for (a = 0; a < REF_LIST; a++)
        if (CRcheck(a) && (best < 0 || PM[a] < PM[best])) best = a;
    if (best < 0) {
        best = 0;
        for (a = 1; a < REF_LIST; a++)
            if (PM[a] < PM[best]) best = a;
    }
    return PM[best];
#else
    return 0.0;
#endif
}

/* Decode `count` frames back to back and return the seconds spent inside the decoder only
 * (cpu_baseline, kind "reference").  y: [count][N]. */
double ref_time_decode(const double *y, double sigma, long count, int *u_hat_last)
{
    struct timespec t0, t1;
    long f;
    if (!ref_ready) ref_init();
    std = sigma;
    clock_gettime(CLOCK_MONOTONIC, &t0);
#if REF_KIND == 4
    (void)f;
#else
    for (f = 0; f < count; f++) REF_DECODE((double *)(y + f * (long)REF_BLOCK), u_hat_last);
#endif
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* Run the reference's own main() untouched (its SNR sweep, stop rule and printf lines). */
int ref_run_main(unsigned long long seed_time)
{
    ref_time_value = seed_time;
    ref_scan_pos = 0;
    ref_scan_dim = REF_BLOCK;
    setvbuf(stdout, NULL, _IOLBF, 0);
    return ref_main();
}

#ifdef REF_BUILD_EXE
int main(int argc, char **argv)
{
    unsigned long long seed = 0;
    if (argc > 1) seed = strtoull(argv[1], NULL, 10);
    return ref_run_main(seed);
}
#endif
