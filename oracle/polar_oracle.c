/*
 * polar_oracle.c -- CPU oracle for the polar-decode hot path.  TEST INFRASTRUCTURE ONLY
 * (see polar_oracle.h: never linked into, or called by, the product path).
 */
#include "polar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- code construction ------------------------------------------------------------------ */

/* I[i] = Q[N-(K+r)+i], inI[I[i]] = 1 (CASCL_1024_L8.c:209-217; SCL_1024.c:198-206 with r = 0) */
po_code *po_code_create(int N, int K, int r, const int *taps, int ntaps, const int *Q)
{
    return po_code_create_q(N, K, r, taps, ntaps, Q, N);
}

/* qlen = number of entries Q really has: the 5G table stops at 1024 and a caller asking for N = 2048 with it used to
 * walk off its end.  Every position taken from Q is checked (range, listed once). */
po_code *po_code_create_q(int N, int K, int r, const int *taps, int ntaps, const int *Q, int qlen)
{
    if (N < 2 || (N & (N - 1)) || K < 1 || r < 0 || K + r > N || ntaps > PO_MAX_TAPS || ntaps < 0) return NULL;
    if (!Q || qlen < N || (r > 0 && (!taps || ntaps < 2))) return NULL;
    for (int i = 0; i < (r > 0 ? ntaps : 0); i++)
        if (taps[i] < 0 || taps[i] > r) return NULL;
    po_code *c = (po_code *)calloc(1, sizeof(po_code));
    if (!c) return NULL;
    c->N = N;
    c->n = 0;
    while ((1 << c->n) < N) c->n++;
    c->K = K;
    c->r = r;
    c->A = K + r;
    c->ntaps = (r > 0) ? ntaps : 0;
    for (int i = 0; i < c->ntaps; i++) c->taps[i] = taps[i];
    c->info_order = (int *)malloc(sizeof(int) * (size_t)c->A);
    c->frozen = (unsigned char *)malloc((size_t)N);
    if (!c->info_order || !c->frozen) { po_code_destroy(c); return NULL; }
    memset(c->frozen, 1, (size_t)N);
    for (int i = 0; i < c->A; i++) {
        const int j = Q[N - c->A + i];
        if (j < 0 || j >= N || c->frozen[j] == 0) { po_code_destroy(c); return NULL; }
        c->info_order[i] = j;
        c->frozen[j] = 0;
    }
    return c;
}

void po_code_destroy(po_code *c)
{
    if (!c) return;
    free(c->info_order);
    free(c->frozen);
    free(c);
}

void po_llr_from_y(const double *y, double sigma, double *llr, int N)
{
    for (int j = 0; j < N; j++) llr[j] = 2 * y[j] / sigma / sigma; /* SCL_1024.c:576 */
}

/* ---- CRC ---------------------------------------------------------------------------------- */

/* w(D) = v(D) g(D), non-systematic (CASCL_1024_L8.c:245-266; CASCL_128.c:205-215) */
void po_crc_encode(const po_code *c, const int *v, int *w)
{
    for (int i = 0; i < c->A; i++) w[i] = 0;
    if (c->r == 0) {
        for (int i = 0; i < c->K; i++) w[i] = v[i];
        return;
    }
    if (c->systematic) {
        /* CASCL_1024_sys.c:776-789: redundant part = sum of the rows Gc[i] = D^(r+i) mod g of the set bits,
         * then the payload itself.  Row i is obtained here by long division of D^(r+i), the way CRcheck
         * divides (:1100-1125); tests/golden/make_golden.py compares every row with the program's literal. */
        const int r = c->r;
        int *row = (int *)malloc(sizeof(int) * (size_t)c->A);
        for (int i = 0; i < c->K; i++) {
            if (v[i] != 1) continue;
            for (int j = 0; j < c->A; j++) row[j] = 0;
            row[r + i] = 1;
            for (int j = r + i; j >= r; j--)
                if (row[j] == 1)
                    for (int t = 0; t < c->ntaps; t++) row[j - r + c->taps[t]] ^= 1;
            for (int j = 0; j < r; j++) w[j] ^= row[j];
        }
        free(row);
        for (int i = r; i < c->A; i++) w[i] = v[i - r];
        return;
    }
    for (int i = 0; i < c->K; i++)
        if (v[i] == 1)
            for (int t = 0; t < c->ntaps; t++) w[i + c->taps[t]] ^= 1;
}

void po_code_set_systematic(po_code *c, int on) { c->systematic = (on && c->r > 0) ? 1 : 0; }

/* long division, pass iff remainder 0 (CASCL_1024_L8.c:569-598; CASCL_128.c:518-536) */
int po_crc_check(const po_code *c, const int *w)
{
    const int A = c->A, r = c->r;
    if (r == 0) return 1;
    int *C = (int *)malloc(sizeof(int) * (size_t)A);
    for (int i = 0; i < A; i++) C[i] = w[i];
    for (int i = A - 1; i >= r; i--)
        if (C[i] == 1)
            for (int t = 0; t < c->ntaps; t++) C[i - r + c->taps[t]] ^= 1;
    int ok = 1;
    for (int i = r - 1; i >= 0; i--)
        if (C[i] == 1) ok = 0;
    free(C);
    return ok;
}

/* x = u F^{(x)n}: butterfly form of the row-add loop (SCL_1024.c:242-250), Fn[i][j] = ((i&j)==j) */
void po_polar_encode(int N, const int *u, int *x)
{
    for (int i = 0; i < N; i++) x[i] = u[i] & 1;
    for (int s = 1; s < N; s <<= 1)
        for (int j = 0; j < N; j++)
            if (!(j & s)) x[j] ^= x[j + s];
}

/* ---- decoders: generic bodies ------------------------------------------------------------- */

/* statistics of the last po_scl_decode_* call (test instrumentation, see polar_oracle_impl.h) */
static __thread int po_last_key_fallbacks, po_last_triples, po_last_phase2, po_last_trivial, po_last_trivial_bound, po_last_k1;
void po_scl_last_stats(int *out)
{
    out[0] = po_last_key_fallbacks;
    out[1] = po_last_triples;
}
/* out[0] = information leaves decided with a full list, out[1] = those where every path just keeps its better branch
 * and the 32-bit keys show it (max of the better keys < min of the worse keys) */
void po_scl_last_prune_stats(int *out)
{
    out[0] = po_last_phase2;
    out[1] = po_last_trivial;
}
/* the same plus out[2] = leaves where the cheaper sufficient test of the pair kernel shows it: the key of
 * (max of the metrics BEFORE the leaf) + 0.65 (>= every favoured candidate, T <= 0.65) is below the key of every
 * PM + |lambda| (<= every other candidate, T >= 0); out[3] = non-trivial leaves with exactly one fork */
void po_scl_last_prune_stats4(int *out)
{
    out[0] = po_last_phase2;
    out[1] = po_last_trivial;
    out[2] = po_last_trivial_bound;
    out[3] = po_last_k1;
}

#define REAL double
#define SFX f64
#include "polar_oracle_impl.h"
#undef REAL
#undef SFX

#define REAL float
#define SFX f32
#include "polar_oracle_impl.h"
#undef REAL
#undef SFX

int po_decode_batch_f64(const po_code *c, int algo, int L, int bp_iters, const double *llr, size_t B,
                        uint32_t *u_hat_bits)
{
    const int N = c->N;
    int *uh = (int *)malloc(sizeof(int) * (size_t)N);
    if (!uh) return -1;
    int rc = 0;
    for (size_t b = 0; b < B && rc == 0; b++) {
        const double *l = llr + b * (size_t)N;
        switch (algo) {
        case 0: rc = po_sc_decode_f64(c, l, uh); break;
        case 1: rc = po_bp_decode_f64(c, l, bp_iters, uh); break;
        case 2: rc = po_scl_decode_f64(c, l, L, 0, uh, NULL, NULL); break;
        case 3: rc = po_scl_decode_f64(c, l, L, 1, uh, NULL, NULL); break;
        default: rc = -4;
        }
        if (u_hat_bits) {
            uint32_t *o = u_hat_bits + b * (size_t)(N / 32);
            for (int wd = 0; wd < N / 32; wd++) {
                uint32_t v = 0;
                for (int i = 0; i < 32; i++) v |= (uint32_t)(uh[wd * 32 + i] & 1) << i;
                o[wd] = v;
            }
        }
    }
    free(uh);
    return rc;
}

/* ---- Monte-Carlo harness ------------------------------------------------------------------- */

void po_sim_init(po_sim *s, uint64_t seed)
{
    memset(s, 0, sizeof(*s));
    s->seed = seed;
    /* PN: x^6 + x^5 + 1 style LFSR as written in SCL_1024.c:185-197 */
    int U[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 63; i++) {
        int b;
        if (i == 0) b = 1;
        else if (i < 6) b = 0;
        else b = U[4] ^ U[5];
        s->pn[i] = b;
        U[5] = U[4]; U[4] = U[3]; U[3] = U[2]; U[2] = U[1]; U[1] = U[0]; U[0] = b;
    }
}

/* SCL_1024.c:295-309 */
double po_sim_ranq1(po_sim *s)
{
    if (s->rani == 0) {
        s->ranv = s->seed ^ 4101842887655102017ULL;
        s->ranv ^= s->ranv >> 21;
        s->ranv ^= s->ranv << 35;
        s->ranv ^= s->ranv >> 4;
        s->ranv = s->ranv * 2685821657736338717ULL;
        s->rani++;
    }
    s->ranv ^= s->ranv >> 21;
    s->ranv ^= s->ranv << 35;
    s->ranv ^= s->ranv >> 4;
    return (double)(s->ranv * 2685821657736338717ULL) * 5.42101086242752217E-20;
}

/* SCL_1024.c:312-326 (Marsaglia polar) */
static void po_sim_normal(po_sim *s, double sigma, double *n1, double *n2)
{
    double x1, x2, q;
    do {
        x1 = po_sim_ranq1(s);
        x2 = po_sim_ranq1(s);
        x1 = 2 * x1 - 1;
        x2 = 2 * x2 - 1;
        q = x1 * x1 + x2 * x2;
    } while (q >= 1.0);
    *n1 = sigma * x1 * sqrt(-2 * log(q) / q);
    *n2 = sigma * x2 * sqrt(-2 * log(q) / q);
}

double po_sigma_from_db(double db) { return pow(10, db / ((double)-20)); }

void po_sim_frame(po_sim *s, const po_code *c, double sigma, int *u, double *y)
{
    const int N = c->N;
    int *v = (int *)malloc(sizeof(int) * (size_t)c->K);
    int *w = (int *)malloc(sizeof(int) * (size_t)c->A);
    int *x = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < c->K; i++) v[i] = s->pn[(s->m + i) % 63]; /* :238-240 */
    po_crc_encode(c, v, w);
    for (int i = 0; i < N; i++) u[i] = 0;
    for (int i = 0; i < c->A; i++) u[c->info_order[i]] = w[i];
    po_polar_encode(N, u, x);
    for (int i = 0; i < N; i += 2) { /* :253-261 */
        double n1, n2;
        po_sim_normal(s, sigma, &n1, &n2);
        y[i] = (x[i] == 0) ? 1 + n1 : -1 + n1;
        if (i + 1 < N) y[i + 1] = (x[i + 1] == 0) ? 1 + n2 : -1 + n2;
    }
    s->m += c->K % 63; /* :273-274 */
    if (s->m >= 63) s->m -= 63;
    free(v); free(w); free(x);
}

int po_count_bit_errors(const po_code *c, const int *u, const int *u_hat)
{
    int e = 0;
    for (int i = c->systematic ? c->r : 0; i < c->A; i++) /* CASCL_1024_sys.c:820-821: the K true info bits */
        if (u[c->info_order[i]] != u_hat[c->info_order[i]]) e++;
    return e;
}

int po_run_sweep(const po_code *c, int algo, int L, int bp_iters, uint64_t seed,
                 const double *snr_db, int n_snr, int ble, long *run_out, long *errbit_out)
{
    return po_run_sweep_min(c, algo, L, bp_iters, seed, snr_db, n_snr, ble, 0, run_out, errbit_out, NULL);
}

int po_run_sweep_min(const po_code *c, int algo, int L, int bp_iters, uint64_t seed, const double *snr_db, int n_snr,
                     int ble, long min_run, long *run_out, long *errbit_out, long *errblock_out)
{
    const int N = c->N;
    po_sim s;
    po_sim_init(&s, seed);
    int *u = (int *)malloc(sizeof(int) * (size_t)N);
    int *uh = (int *)malloc(sizeof(int) * (size_t)N);
    double *y = (double *)malloc(sizeof(double) * (size_t)N);
    double *llr = (double *)malloc(sizeof(double) * (size_t)N);
    int rc = 0;
    for (int p = 0; p < n_snr && rc == 0; p++) {
        double sigma = po_sigma_from_db(snr_db[p]);
        long run = 0, errbit = 0;
        int errblock = 0;
        for (run = 0; (errblock < ble || run < min_run) && rc == 0; run++) {
            po_sim_frame(&s, c, sigma, u, y);
            po_llr_from_y(y, sigma, llr, N);
            switch (algo) {
            case 0: rc = po_sc_decode_f64(c, llr, uh); break;
            case 1: rc = po_bp_decode_f64(c, llr, bp_iters, uh); break;
            case 2: rc = po_scl_decode_f64(c, llr, L, 0, uh, NULL, NULL); break;
            case 3: rc = po_scl_decode_f64(c, llr, L, 1, uh, NULL, NULL); break;
            default: rc = -4;
            }
            int e = po_count_bit_errors(c, u, uh);
            errbit += e;
            errblock += (e != 0);
        }
        run_out[p] = run;
        if (errbit_out) errbit_out[p] = errbit;
        if (errblock_out) errblock_out[p] = errblock;
    }
    free(u); free(uh); free(y); free(llr);
    return rc;
}
