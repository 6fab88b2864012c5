/*
 * polar_sim.c -- C host harness over libpolar_hip.so: the reference's main() (SCL_1024.c:136-292,
 * CASCL_1024_L8.c:143-315) with the per-frame decode call replaced by batched GPU decodes.
 *
 * The transmit chain is the reference's, kept sequential on the host so that a fixed SEED reproduces the
 * published run counts: PN source (SCL_1024.c:184-197), CRC multiply by g(D) (CASCL_1024_L8.c:245-266),
 * u[I[i]] = w[i], x = u F^{(x)n} (:242-250, butterfly form), y = +-1 + n with n from Marsaglia polar on
 * Ranq1 (:295-326), std = 10^(-dB/20) (:226).  Frames are produced in order and handed over `batch` at a time
 * to polar_stop_rule_batch_y: the device decodes them (the kernel forms 2*y/std/std), compares with the sent
 * bits and applies the reference's sequential stop rule `for (run = 0; errBlock < BLE; run++)` (:228) as a
 * prefix count over its per-frame error counters; only (frames consumed, block errors, bit errors) come back.
 * The generator state is rewound to just after the last consumed frame (RNG and PN phase carry over SNR points
 * exactly as in the reference, which never resets them).
 *
 * Output lines follow the reference's printf formats (CASCL_1024_L8.c:308, SC_128.c:218-221).
 *
 *   polar_sim --algo cascl --N 1024 --K 512 --L 8 --crc 24c --seed 1242 --ble 100 --snr 1.0:2.0:0.5
 *
 * --fast: throughput mode.  Frames come from the device-side generator (polar_fer_batch: counter-based RNG,
 * not the reference's sequential stream), whole batches are counted, and a point stops after the first batch
 * that brings the block errors to >= BLE.  Same code, same decoder, different (statistically equivalent)
 * noise: use it for FER curves at 10^6..10^7 frames/s, not for reproducing published run counts.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <unistd.h>

#include "polar_hip.h"

typedef struct {
    uint64_t seed, ranv;
    int rani;
    int m; /* PN phase */
} gen_state;

static int PN[63];

static void pn_init(void) /* SCL_1024.c:184-197 */
{
    int U[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 63; i++) {
        int b = (i == 0) ? 1 : (i < 6) ? 0 : (U[4] ^ U[5]);
        PN[i] = b;
        U[5] = U[4]; U[4] = U[3]; U[3] = U[2]; U[2] = U[1]; U[1] = U[0]; U[0] = b;
    }
}

static double ranq1(gen_state *g) /* SCL_1024.c:295-309 */
{
    if (g->rani == 0) {
        g->ranv = g->seed ^ 4101842887655102017ULL;
        g->ranv ^= g->ranv >> 21; g->ranv ^= g->ranv << 35; g->ranv ^= g->ranv >> 4;
        g->ranv *= 2685821657736338717ULL;
        g->rani++;
    }
    g->ranv ^= g->ranv >> 21; g->ranv ^= g->ranv << 35; g->ranv ^= g->ranv >> 4;
    return (double)(g->ranv * 2685821657736338717ULL) * 5.42101086242752217E-20;
}

typedef struct {
    int N, K, r, A, ntaps;
    int taps[33];
    int *I; /* info order */
    int sys; /* --sys: systematic CRC encoding and K-bit error metric (CASCL_1024_sys.c) */
    const uint32_t *gc; /* --crc-file: the generator matrix rows as loaded (bit j of gc[i] = Gc[i][j]), else NULL */
} code_t;

/* The reference's generator is one sequential stream, but only its cheap part is: the xorshift steps and the
 * rejection test of the polar method (SCL_1024.c:295-326).  draw_frame() runs that part in frame order and keeps the
 * accepted (x1, x2, s) of every pair; finish_frame() -- payload, CRC, encoding and the log/sqrt of the noise, with the
 * reference's own expressions -- is then done for the frames of a batch in parallel.  Same bits as doing it all in order. */
typedef struct { double x1, x2, s; } pair_t;

static void draw_frame(gen_state *g, const code_t *c, pair_t *pr, int *m_of_frame)
{
    *m_of_frame = g->m;
    for (int i = 0; i < c->N / 2; i++) { /* SCL_1024.c:312-326, the loop of normal() */
        double x1, x2, s;
        do {
            x1 = 2 * ranq1(g) - 1;
            x2 = 2 * ranq1(g) - 1;
            s = x1 * x1 + x2 * x2;
        } while (s >= 1.0);
        pr[i].x1 = x1; pr[i].x2 = x2; pr[i].s = s;
    }
    g->m += c->K % 63; /* :273-274 */
    if (g->m >= 63) g->m -= 63;
}

static void finish_frame(const code_t *c, int m, const pair_t *pr, double sigma, unsigned char *u, double *y)
{
    const int N = c->N;
    int w[4096 + 64], d[4096 + 64];
    unsigned char x[4096];
    for (int i = 0; i < c->A; i++) w[i] = 0;
    if (c->sys && c->r > 0) {
        /* CASCL_1024_sys.c:776-789: w[r..A) = payload, w[0..r) = D^r v(D) mod g (sum of the generator rows) */
        for (int i = 0; i < c->K; i++) w[c->r + i] = PN[(m + i) % 63];
        if (c->gc) {   /* the reference's own loop: add row i of Gc for every payload bit that is 1 */
            for (int i = 0; i < c->K; i++)
                if (w[c->r + i])
                    for (int j = 0; j < c->r; j++) w[j] ^= (int)((c->gc[i] >> j) & 1u);
        } else {
        for (int i = 0; i < c->A; i++) d[i] = (i < c->r) ? 0 : w[i];
        for (int i = c->A - 1; i >= c->r; i--)
            if (d[i])
                for (int t = 0; t < c->ntaps; t++) d[i - c->r + c->taps[t]] ^= 1;
        for (int i = 0; i < c->r; i++) w[i] = d[i];
        }
    } else
    for (int i = 0; i < c->K; i++)
        if (PN[(m + i) % 63]) {
            if (c->r == 0) w[i] ^= 1;
            else for (int t = 0; t < c->ntaps; t++) w[i + c->taps[t]] ^= 1; /* CASCL_1024_L8.c:251-266 */
        }
    memset(u, 0, (size_t)N);
    for (int i = 0; i < c->A; i++) u[c->I[i]] = (unsigned char)w[i];
    memcpy(x, u, (size_t)N);
    for (int s = 1; s < N; s <<= 1)
        for (int j = 0; j < N; j++)
            if (!(j & s)) x[j] ^= x[j + s];
    for (int i = 0; i < N; i += 2) { /* SCL_1024.c:253-261 with :322-323 */
        const double x1 = pr[i / 2].x1, x2 = pr[i / 2].x2, s = pr[i / 2].s;
        const double n1 = sigma * x1 * sqrt(-2 * log(s) / s);
        const double n2 = sigma * x2 * sqrt(-2 * log(s) / s);
        y[i] = x[i] ? -1 + n1 : 1 + n1;
        y[i + 1] = x[i + 1] ? -1 + n2 : 1 + n2;
    }
}

typedef struct {
    const code_t *c; const int *m; const pair_t *pr; double sigma; unsigned char *u; double *y;
    int f0, f1;
} job_t;

static void *finish_job(void *arg)
{
    const job_t *j = (const job_t *)arg;
    const int N = j->c->N;
    for (int f = j->f0; f < j->f1; f++)
        finish_frame(j->c, j->m[f], j->pr + (size_t)f * (N / 2), j->sigma, j->u + (size_t)f * N, j->y + (size_t)f * N);
    return NULL;
}

/* a batch of frames in the reference's order; after[f] = generator state just after frame f */
static void make_batch(gen_state *g, const code_t *c, double sigma, int batch, unsigned char *u, double *y, gen_state *after,
                       pair_t *pr, int *mf)
{
    const int N = c->N;
    for (int f = 0; f < batch; f++) {
        draw_frame(g, c, pr + (size_t)f * (N / 2), &mf[f]);
        after[f] = *g;
    }
    long nt = sysconf(_SC_NPROCESSORS_ONLN);
    if (nt > 16) nt = 16;
    if (nt > batch / 64) nt = batch / 64;
    if (nt < 1) nt = 1;
    pthread_t th[16];
    job_t jobs[16];
    for (long t = 0; t < nt; t++) {
        jobs[t] = (job_t){c, mf, pr, sigma, u, y, (int)((long)batch * t / nt), (int)((long)batch * (t + 1) / nt)};
        if (t + 1 < nt) pthread_create(&th[t], NULL, finish_job, &jobs[t]);
    }
    finish_job(&jobs[nt - 1]);
    for (long t = 0; t + 1 < nt; t++) pthread_join(th[t], NULL);
}

static const int CRC24C[] = {0, 1, 2, 4, 8, 12, 13, 15, 17, 20, 21, 23, 24};
static const int CRC6[] = {0, 5, 6};

static void usage(void)
{
    fprintf(stderr, "usage: polar_sim --algo sc|bp|bpr|scl|cascl --N n --K k [--L l] [--crc 24c|6 | --crc-file m.dat] [--sys] [--seed s] [--ble b]\n"
                    "                 [--snr lo:hi:step | --snr-list a,b,..] [--batch b] [--dtype f64|f32] [--bp-iters i] [--q file] [--fn file] [--min-run m] [--fast [--gpus g]]\n");
    exit(2);
}

int main(int argc, char **argv)
{
    int N = 1024, K = 512, L = 8, algo = POLAR_ALGO_CASCL, ble = 100, batch = 4096, dtype = POLAR_F64, bp_iters = 100;
    int fast = 0, sys = 0, bpr = 0, gpus = 1;
    long min_run = 0;   /* --min-run m: `errBlock < BLE || run < m`, the rule of the published L = 32 logs (m = 2000) */
    uint64_t seed = 1024;
    double lo = 1.0, hi = 3.0, step = 0.5;
    double pts[64];
    int npts = 0;   /* --snr-list a,b,c: explicit Eb/N0 points (the published L = 32 log goes 1.0, 1.5, 2.0, 2.2) */
    const char *crc = NULL, *qfile = NULL, *fnfile = NULL, *crcfile = NULL;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(a, "--algo") && v) {
            if (!strcmp(v, "bpr")) { bpr = 1; v = "bp"; bp_iters = 90; }   /* BPr_128.c: iterMax 90 (:16) */
            algo = !strcmp(v, "sc") ? POLAR_ALGO_SC : !strcmp(v, "bp") ? POLAR_ALGO_BP
                 : !strcmp(v, "scl") ? POLAR_ALGO_SCL : !strcmp(v, "cascl") ? POLAR_ALGO_CASCL : -1;
            if (algo < 0) usage();
            i++;
        } else if (!strcmp(a, "--N") && v) { N = atoi(v); i++; }
        else if (!strcmp(a, "--K") && v) { K = atoi(v); i++; }
        else if (!strcmp(a, "--L") && v) { L = atoi(v); i++; }
        else if (!strcmp(a, "--crc") && v) { crc = v; i++; }
        else if (!strcmp(a, "--crc-file") && v) { crcfile = v; i++; }   /* CRC_6.dat / Gc[K][r] generator-matrix file */
        else if (!strcmp(a, "--seed") && v) { seed = strtoull(v, NULL, 10); i++; }
        else if (!strcmp(a, "--ble") && v) { ble = atoi(v); i++; }
        else if (!strcmp(a, "--batch") && v) { batch = atoi(v); i++; }
        else if (!strcmp(a, "--bp-iters") && v) { bp_iters = atoi(v); i++; }
        else if (!strcmp(a, "--q") && v) { qfile = v; i++; }
        else if (!strcmp(a, "--min-run") && v) { min_run = atol(v); i++; }
        else if (!strcmp(a, "--fn") && v) { fnfile = v; i++; }
        else if (!strcmp(a, "--fast")) { fast = 1; }
        else if (!strcmp(a, "--gpus") && v) { gpus = atoi(v); i++; }   /* --fast only: frames sharded over the GPUs of the node */
        else if (!strcmp(a, "--sys")) { sys = 1; }
        else if (!strcmp(a, "--dtype") && v) { dtype = !strcmp(v, "f32") ? POLAR_F32 : POLAR_F64; i++; }
        else if (!strcmp(a, "--snr-list") && v) {
            const char *q = v;
            while (*q && npts < 64) {
                char *end;
                pts[npts++] = strtod(q, &end);
                if (end == q) usage();
                q = (*end == ',') ? end + 1 : end;
            }
            i++;
        } else if (!strcmp(a, "--snr") && v) {
            if (sscanf(v, "%lf:%lf:%lf", &lo, &hi, &step) != 3) usage();
            i++;
        } else usage();
    }
    if (N > 4096 || N < 32) usage();
    if (npts == 0)
        for (double db = lo; db <= hi + 1e-12 && npts < 64; db += step) pts[npts++] = db;   /* SCL_1024.c:223 */
    code_t c;
    memset(&c, 0, sizeof c);
    c.N = N; c.K = K;
    polar_crc_matrix gcm;
    memset(&gcm, 0, sizeof gcm);
    if (algo == POLAR_ALGO_CASCL && crcfile) {
        /* g(D) and r come from the generator-matrix file (the reference's CRC_6.dat, or Gc of CASCL_1024_sys.c:48-561 as
           a file); the library has checked that every row is D^(r+i) mod g */
        if (crc) { fprintf(stderr, "--crc and --crc-file exclude each other\n"); return 1; }
        int lrc = polar_crc_matrix_load(crcfile, &gcm);
        if (lrc) { fprintf(stderr, "%s: not a CRC generator matrix (%s)\n", crcfile, polar_strerror(lrc)); return 1; }
        if (K > gcm.K) { fprintf(stderr, "%s has %d rows, --K %d needs more\n", crcfile, gcm.K, K); return 1; }
        c.ntaps = gcm.n_taps;
        memcpy(c.taps, gcm.taps, sizeof(int) * (size_t)c.ntaps);
        c.r = gcm.r;
        c.gc = gcm.rows;
    } else if (algo == POLAR_ALGO_CASCL) {
        if (!crc) crc = (N == 128) ? "6" : "24c";
        const int *t = !strcmp(crc, "6") ? CRC6 : CRC24C;
        c.ntaps = !strcmp(crc, "6") ? 3 : 13;
        memcpy(c.taps, t, sizeof(int) * (size_t)c.ntaps);
        c.r = c.taps[c.ntaps - 1];
    }
    c.A = K + c.r;
    c.sys = sys && c.r > 0;
    /* --fn file: the N x N generator matrix the reference reads from stdin (SCL_1024.c:207-217), read the same way: N*N
       whitespace-separated integers, "Illegal input!" for anything but 0 / 1.  The encoder here (and the factor graph of
       every decoder, the reference's included) is F^{(x)n}, Fn[i][j] = ((i & j) == j): any other matrix is refused. */
    if (fnfile) {
        FILE *ff = strcmp(fnfile, "-") ? fopen(fnfile, "r") : stdin;
        if (!ff) { fprintf(stderr, "cannot open %s\n", fnfile); return 1; }
        long bad = 0;
        for (long i = 0; i < (long)N * N; i++) {
            int temp;
            if (fscanf(ff, "%d", &temp) != 1) { fprintf(stderr, "%s: fewer than %d x %d entries\n", fnfile, N, N); return 1; }
            if (temp != 0 && temp != 1) printf("Illegal input!\n");
            const int want = (((i / N) & (i % N)) == (i % N)) ? 1 : 0;
            bad += (temp != want);
        }
        if (ff != stdin) fclose(ff);
        if (bad) { fprintf(stderr, "%s: %ld entries differ from the Kronecker power of [1 0; 1 1]; not supported\n", fnfile, bad); return 1; }
    }
    /* --q file: N whitespace-separated positions in ascending reliability (the shape of the reference's `const int Q[N]`
       literal, SC_1024.c:42-91); the information set is its last K + r entries, I[i] = Q[N-(K+r)+i] (CASCL_1024_L8.c:214-217) */
    int *qorder = NULL;
    if (qfile) {
        FILE *fq = fopen(qfile, "r");
        if (!fq) { fprintf(stderr, "cannot open %s\n", qfile); return 1; }
        qorder = (int *)malloc(sizeof(int) * (size_t)N);
        for (int i = 0; i < N; i++)
            if (fscanf(fq, "%d", &qorder[i]) != 1 || qorder[i] < 0 || qorder[i] >= N) {
                fprintf(stderr, "%s: need %d positions in [0, %d)\n", qfile, N, N);
                return 1;
            }
        fclose(fq);
    }

    polar_cfg cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.N = N; cfg.K = K; cfg.crc_r = c.r; cfg.crc_taps = c.r ? c.taps : NULL; cfg.n_taps = c.ntaps;
    cfg.L = L; cfg.algo = algo; cfg.bp_iters = bp_iters; cfg.info_order = qorder ? qorder + (N - c.A) : NULL; cfg.dtype = dtype; cfg.device = 0;
    cfg.crc_systematic = c.sys;
    polar_ctx *ctx = NULL;
    int rc = (c.gc) ? polar_create_crc_file(&cfg, crcfile, &ctx) : polar_create(&cfg, &ctx);
    if (rc) { fprintf(stderr, "polar_create: %s\n", polar_strerror(rc)); return 1; }
    /* the library built the frozen set from the 5G sequence like the reference (I[i] = Q[N-(K+r)+i]);
       the encoder needs the same I[] */
    c.I = (int *)malloc(sizeof(int) * (size_t)c.A);
    if (polar_info_order(ctx, c.I, c.A) != 0) { fprintf(stderr, "polar_info_order failed\n"); return 1; }

    pn_init();
    gen_state g = {seed, 0, 0, 0};
    double *y = (double *)malloc(sizeof(double) * (size_t)batch * N);
    unsigned char *u = (unsigned char *)malloc((size_t)batch * N);
    uint32_t *ubits = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)batch * (N / 32));
    gen_state *after = (gen_state *)malloc(sizeof(gen_state) * (size_t)batch);
    pair_t *pairs = (pair_t *)malloc(sizeof(pair_t) * (size_t)batch * (N / 2));
    int *mf = (int *)malloc(sizeof(int) * (size_t)batch);
    if (!y || !u || !ubits || !after || !pairs || !mf) { fprintf(stderr, "out of memory\n"); return 1; }
    printf("SEED = %llu\n", (unsigned long long)seed);
    if (fast) {
        unsigned long long first = 0;
        polar_group *grp = NULL;   /* --gpus N: one context per GPU + the RCCL communicators, made once for the sweep */
        if (gpus > 1 && (rc = polar_group_create(&cfg, gpus, &grp)) != 0) {
            fprintf(stderr, "polar_group_create(%d GPUs): %s\n", gpus, polar_strerror(rc));
            return 1;
        }
        for (int ip = 0; ip < npts; ip++) {
            const double db = pts[ip];
            unsigned long long blk = 0, bits = 0, run = 0;
            while (blk < (unsigned long long)ble) {
                if (gpus > 1) {   /* `batch` frames per GPU and round; one RCCL all-reduce of the two counters per round */
                    rc = polar_group_fer_batch(grp, seed, first, db, (size_t)batch, &blk, &bits, NULL);
                    if (rc) { fprintf(stderr, "polar_group_fer_batch: %s\n", polar_strerror(rc)); return 1; }
                    first += (unsigned long long)batch * (unsigned long long)gpus;
                    run += (unsigned long long)batch * (unsigned long long)gpus;
                    continue;
                }
                rc = polar_fer_batch(ctx, seed, first, db, (size_t)batch, &blk, &bits);
                if (rc) { fprintf(stderr, "fer_batch: %s (%s)\n", polar_strerror(rc), polar_last_error(ctx)); return 1; }
                first += (unsigned long long)batch;
                run += (unsigned long long)batch;
            }
            printf("L = %d\tbSNR = %.2lf\terror block = %llu\trun = %llu\tBLER = %le\tBER = %le\n", L, db, blk, run,
                   (double)blk / (double)run, (double)bits / (double)run / (double)(c.sys ? c.K : c.A));
            fflush(stdout);
        }
        polar_group_destroy(grp);
        polar_destroy(ctx);
        return 0;
    }
    static const int CP[6] = {3, 6, 10, 20, 40, 80};        /* BPr_128.c:18-23 */
    int *ui = bpr ? (int *)malloc(sizeof(int) * (size_t)batch * N) : NULL;
    int nlog = 0;
    while ((1 << nlog) < N) nlog++;
    for (int ip = 0; ip < npts; ip++) {
        const double db = pts[ip];
        const double sigma = pow(10, db / ((double)-20)); /* :226 */
        long run = 0, errbit = 0;
        int errblock = 0;
        unsigned long long E[6 * 16];
        memset(E, 0, sizeof E);
        while (errblock < ble || run < min_run) {
            make_batch(&g, &c, sigma, batch, u, y, after, pairs, mf);
            for (size_t k = 0; k < (size_t)batch * (N / 32); k++) { /* sent bits, packed like the decisions */
                uint32_t wd = 0;
                for (int b = 0; b < 32; b++) wd |= (uint32_t)(u[k * 32 + b] & 1) << b;
                ubits[k] = wd;
            }
            /* decode + compare (:266-272) + stop rule (:228) on the device; the batch is cut WITH the frame that brings
               the block errors to BLE */
            size_t used = 0;
            unsigned long long eb = 0, ebit = 0;
            rc = polar_stop_rule_batch_y(ctx, y, sigma, ubits, (size_t)batch, (unsigned)(errblock < ble ? ble - errblock : 0),
                                         (size_t)(run < min_run ? min_run - run : 0), &used, &eb, &ebit);
            if (rc) { fprintf(stderr, "decode: %s (%s)\n", polar_strerror(rc), polar_last_error(ctx)); return 1; }
            const int f = (int)used;
            errblock += (int)eb;
            errbit += (long)ebit;
            run += (long)used;
            if (bpr) { /* read-outs of exactly the frames that count (BPr_128.c:417-565 runs inside BPr()) */
                for (size_t k = 0; k < (size_t)f * N; k++) ui[k] = u[k];
                rc = polar_bp_readout_batch(ctx, y, sigma, (size_t)f, ui, CP, 6, E, NULL);
                if (rc) { fprintf(stderr, "readout: %s (%s)\n", polar_strerror(rc), polar_last_error(ctx)); return 1; }
            }
            if (errblock >= ble && run >= min_run) g = after[f - 1]; /* rewind to just after the frame that ended the point */
        }
        if (bpr) { /* BPr_128.c:227-258 */
            printf("bSNR = %.2lf\terror block = %d\trun = %ld\t", db, errblock, run);
            for (int q = 0; q < 6; q++) {
                printf(q ? "After %2d iterations:\n" : "\nAfter %2d iterations:\n", CP[q]);
                for (int i = 0; i <= nlog; i++) printf("%lf\t", (double)E[q * (nlog + 1) + i] / run);
                printf("\n");
            }
            printf("BLER = %lfe-2\tBER = %lfe-2\tK * BER = %lf\n", (double)errblock * 100 / run,
                   (double)errbit * 100 / K / run, (double)errbit / run);
        } else if (algo == POLAR_ALGO_SC || algo == POLAR_ALGO_BP) {
            printf("bSNR = %.2lf\terror block = %d\trun = %ld\tBLER = %lf\n", db, errblock, run, (double)errblock / run);
            printf("Error bit = %ld\tBER = %lf\n", errbit, (double)errbit / K / run);
        } else if (c.sys) { /* CASCL_1024_sys.c:832-835 */
            printf("bSNR = %.2lf\trun = %ld\tBLER = %lfe-3\t", db, run, (double)errblock / (run / 1000.0));
            printf("Error bit = %ld\tBER = %lfe-3\n", errbit, (double)errbit / K / (run / 1000.0));
        } else {
            printf("L = %d\tbSNR = %.2lf\terror block = %d\trun = %ld\tBLER = %lfe-4\n", L, db, errblock, run,
                   (double)errblock / (run / 10000.0));
        }
        fflush(stdout);
    }
    polar_destroy(ctx);
    free(y); free(u); free(ubits); free(after); free(c.I); free(qorder);
    return 0;
}
