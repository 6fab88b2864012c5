/*
 * polar_oracle_literal.c -- LITERAL model of the reference's list decoder state.  TEST INFRASTRUCTURE ONLY.
 *
 * polar_oracle_impl.h restates SCLdecode / CASCL as the algorithm they implement (array levels, lazy
 * pointers).  That is exact as long as no median tie occurs.  On a tie (SCL_1024.c:619-633: strict
 * `< med` keeps fewer than L candidates, "Oops!") the reference leaves slots that nobody refills running on
 * with stale node records: updateBit() is not called for them at that leaf, so bDone stays 0 there for the
 * rest of the frame, every later lower-node evaluation that needs that bit prints "Wrong propagation
 * order!" (:417-418) and keeps whatever l[] the node record held before -- from an earlier frame, or from a
 * simpleCopy() (:467-478), which copies values without the flags.  What the reference outputs on such a
 * frame therefore depends on the records' history, not only on (y, std).
 *
 * This file models exactly that: the full (n+1) x N x L node records (l, b, lDone, bDone) as a persistent
 * object, the recursive getLLR / updateBit on them, copyPath / simpleCopy as value copies, the diagnostics
 * as counters.  tests/test_oracle_ties.py checks it against the compiled reference (oracle/_ref) on tied
 * and untied frame sequences -- output, path metrics and the number of each diagnostic -- and uses it to
 * state where the build's own tie rule ("Median ties") departs from the reference.
 * double only: the reference has no other arithmetic.
 */
#include "polar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

struct po_lit {
    const po_code *c;
    int N, n, L;
    double *l;          /* [n+1][N][L]  V[i][j]->l[k]      SCL_1024.c:18-23 */
    int *b;             /* [n+1][N][L]  V[i][j]->b[k]                       */
    unsigned char *ld;  /* [n+1][N][L]  lDone                               */
    unsigned char *bd;  /* [n+1][N][L]  bDone                               */
    double PM[2 * PO_MAX_L], cand[2 * PO_MAX_L];
    int surviv[PO_MAX_L];
    long diag[3];       /* "Oops!" :622, "Wrong propagation order!" :418, "Error!" :651 */
};

#define IX(s, i, j, k) ((((size_t)(i) * (size_t)(s)->N) + (size_t)(j)) * (size_t)(s)->L + (size_t)(k))

po_lit *po_lit_create(const po_code *c, int L)
{
    if (!c || L < 1 || L > PO_MAX_L || (L & (L - 1))) return NULL;
    po_lit *s = (po_lit *)calloc(1, sizeof(po_lit));
    if (!s) return NULL;
    s->c = c; s->N = c->N; s->n = c->n; s->L = L;
    const size_t cells = (size_t)(c->n + 1) * (size_t)c->N * (size_t)L;
    s->l = (double *)calloc(cells, sizeof(double));
    s->b = (int *)calloc(cells, sizeof(int));
    s->ld = (unsigned char *)calloc(cells, 1);
    s->bd = (unsigned char *)calloc(cells, 1);
    if (!s->l || !s->b || !s->ld || !s->bd) { po_lit_destroy(s); return NULL; }
    return s;
}

void po_lit_destroy(po_lit *s)
{
    if (!s) return;
    free(s->l); free(s->b); free(s->ld); free(s->bd);
    free(s);
}

/* the records as calloc() leaves them in main() (SCL_1024.c:159-164) */
void po_lit_reset(po_lit *s)
{
    const size_t cells = (size_t)(s->n + 1) * (size_t)s->N * (size_t)s->L;
    memset(s->l, 0, cells * sizeof(double));
    memset(s->b, 0, cells * sizeof(int));
    memset(s->ld, 0, cells);
    memset(s->bd, 0, cells);
    memset(s->PM, 0, sizeof s->PM);
    memset(s->cand, 0, sizeof s->cand);
    memset(s->surviv, 0, sizeof s->surviv);
    memset(s->diag, 0, sizeof s->diag);
}

/* Fill every record a frame does not initialise itself (l and b of stages < n, all paths) with arbitrary values:
 * stands for "whatever earlier frames left there".  A frame whose result is the same from the calloc() state and
 * from differently poisoned states does not depend on its history. */
void po_lit_poison(po_lit *s, uint64_t seed)
{
    const size_t cells = (size_t)s->n * (size_t)s->N * (size_t)s->L;
    uint64_t v = seed * 2685821657736338717ULL + 1442695040888963407ULL;
    for (size_t i = 0; i < cells; i++) {
        v ^= v >> 21; v ^= v << 35; v ^= v >> 4;
        const uint64_t x = v * 2685821657736338717ULL;
        s->l[i] = ((double)(int64_t)(x >> 40) - 8388608.0) / 262144.0;   /* multiples of 2^-18 in [-32, 32) */
        s->b[i] = (int)(x & 1);
    }
}

void po_lit_diag(po_lit *s, long *out, int reset)
{
    for (int i = 0; i < 3; i++) {
        out[i] = s->diag[i];
        if (reset) s->diag[i] = 0;
    }
}

void po_lit_path_metrics(const po_lit *s, double *out)
{
    for (int k = 0; k < s->L; k++) out[k] = s->PM[k];
}

/* SCL_1024.c:343-374 */
static double lit_tab(double a)
{
    if (a < 0.196) return 0.65;
    if (a < 0.433) return 0.55;
    if (a < 0.71) return 0.45;
    if (a < 1.05) return 0.35;
    if (a < 1.508) return 0.25;
    if (a < 2.252) return 0.15;
    if (a < 4.5) return 0.05;
    return 0;
}

static double lit_chk(double L1, double L2)
{
    double delta = lit_tab(fabs(L1 + L2));
    delta -= lit_tab(fabs(L1 - L2));
    double A1 = fabs(L1), A2 = fabs(L2);
    int sg = ((L1 >= 0) ? 1 : -1) * ((L2 >= 0) ? 1 : -1);
    if (A1 > A2) return sg * A2 + delta;
    return sg * A1 + delta;
}

/* PHI(k, j, u), SCL_1024.c:481-502: reads the node record's l[k], whatever it holds */
static double lit_phi(const po_lit *s, int k, int j, int u)
{
    const double lam = s->l[IX(s, 0, j, k)];
    const double a = fabs(lam);
    double res = lit_tab(a);
    if ((u == 0 && lam < 0) || (u == 1 && lam > 0)) res += a;
    return res;
}

/* Stage i couples rows ju (bit i clear, "upper", leftP = 1) and jl = ju + 2^i; both have the children
 * V[i+1][ju] (cU) and V[i+1][jl] (cL) (connectBCB, SCL_1024.c:377-401). */
static void lit_get_llr(po_lit *s, int i, int j, int k)   /* getLLR, :404-421 */
{
    if (s->ld[IX(s, i, j, k)]) return;
    const int st = 1 << i, ju = j & ~st, jl = j | st;
    lit_get_llr(s, i + 1, ju, k);
    lit_get_llr(s, i + 1, jl, k);
    const double cu = s->l[IX(s, i + 1, ju, k)], cl = s->l[IX(s, i + 1, jl, k)];
    if (!(j & st))
        s->l[IX(s, i, j, k)] = lit_chk(cu, cl);
    else if (s->bd[IX(s, i, ju, k)]) {
        if (s->b[IX(s, i, ju, k)] == 0) s->l[IX(s, i, j, k)] = cl + cu;
        else s->l[IX(s, i, j, k)] = cl - cu;
    } else
        s->diag[1]++;   /* "Wrong propagation order!": l[k] keeps its old content */
    s->ld[IX(s, i, j, k)] = 1;
}

static void lit_update_bit(po_lit *s, int i, int j, int k)   /* updateBit, :424-448 */
{
    if (s->bd[IX(s, i, j, k)]) return;
    s->bd[IX(s, i, j, k)] = 1;
    if (i == s->n) return;
    const int st = 1 << i, ju = j & ~st, jl = j | st;
    if (!(j & st)) {
        if (s->bd[IX(s, i, jl, k)]) {
            s->b[IX(s, i + 1, ju, k)] = (s->b[IX(s, i, j, k)] + s->b[IX(s, i, jl, k)]) % 2;
            lit_update_bit(s, i + 1, ju, k);
        }
    } else {
        if (s->bd[IX(s, i, ju, k)]) {
            s->b[IX(s, i + 1, ju, k)] = (s->b[IX(s, i, j, k)] + s->b[IX(s, i, ju, k)]) % 2;
            lit_update_bit(s, i + 1, ju, k);
        }
        s->b[IX(s, i + 1, jl, k)] = s->b[IX(s, i, j, k)];
        lit_update_bit(s, i + 1, jl, k);
    }
}

static void lit_copy(po_lit *s, int c, int k, int with_flags)   /* copyPath :451-464 / simpleCopy :467-478 */
{
    for (int i = 0; i < s->n; i++)
        for (int j = 0; j < s->N; j++) {
            s->l[IX(s, i, j, k)] = s->l[IX(s, i, j, c)];
            s->b[IX(s, i, j, k)] = s->b[IX(s, i, j, c)];
            if (with_flags) {
                s->ld[IX(s, i, j, k)] = s->ld[IX(s, i, j, c)];
                s->bd[IX(s, i, j, k)] = s->bd[IX(s, i, j, c)];
            }
        }
}

/* QuickSort / Partition, SCL_1024.c:505-544, step for step -- because of what it does on equal keys: both scans
 * stop ON a key equal to the pivot, so with the pivot value present at two more places the swap exchanges two equal
 * keys and neither index moves: the reference never returns from that frame.  Two equal keys alone (the ordinary
 * median tie) sort fine.  Returns -1 in *hang when the loop would not end. */
static int lit_partition(double *a, int low, int high, int *hang)
{
    const double v = a[low];
    int i = low + 1, j = high;
    do {
        const int i0 = i, j0 = j;
        while (a[i] < v && i < high) i += 1;
        while (a[j] > v) j -= 1;
        if (i < j) {
            if (i == i0 && j == j0 && a[i] == a[j]) { *hang = 1; return j; }   /* state repeats for ever */
            const double t = a[i];
            a[i] = a[j];
            a[j] = t;
        }
    } while (i < j);
    a[low] = a[j];
    a[j] = v;
    return j;
}

static void lit_quicksort(double *a, int low, int high, int *hang)
{
    if (low < high && !*hang) {
        const int mid = lit_partition(a, low, high, hang);
        if (*hang) return;
        lit_quicksort(a, low, mid - 1, hang);
        lit_quicksort(a, mid + 1, high, hang);
    }
}

/* CRcheck(k), CASCL_1024_L8.c:569-598 */
static int lit_crc(const po_lit *s, int k)
{
    const po_code *c = s->c;
    int *w = (int *)malloc(sizeof(int) * (size_t)c->A);
    for (int i = 0; i < c->A; i++) w[i] = s->b[IX(s, 0, c->info_order[i], k)];
    int ok = po_crc_check(c, w);
    free(w);
    return ok;
}

/* SCLdecode (SCL_1024.c:547-680) / CASCL (CASCL_1024_L8.c:601-761) on the persistent records.
 * llr = 2*y/std/std formed by the caller (po_llr_from_y).
 * Returns 0; -3 where the reference prints "Error!" and then indexes past its arrays; -5 where the reference
 * does not return (Partition's endless loop on three equal keys).  After -3 / -5 the records are as the
 * reference's were at that point. */
int po_lit_decode(po_lit *s, const double *llr, int crc, int *u_hat, double *pm_out)
{
    const po_code *c = s->c;
    const int N = s->N, n = s->n, L = s->L;
    double *PM = s->PM, *cand = s->cand;
    int *surviv = s->surviv;
    int i, j, k, actL;
    PM[0] = 0;
    for (i = 0; i <= n; i++)
        for (j = 0; j < N; j++) s->bd[IX(s, i, j, 0)] = 0;
    for (j = 0; j < N; j++)
        if (c->frozen[j]) {
            s->b[IX(s, 0, j, 0)] = 0;
            lit_update_bit(s, 0, j, 0);
        }
    for (i = 0; i < n; i++)
        for (j = 0; j < N; j++) s->ld[IX(s, i, j, 0)] = 0;
    for (j = 0; j < N; j++)
        for (k = 0; k < L; k++) {
            s->l[IX(s, n, j, k)] = llr[j];
            s->ld[IX(s, n, j, k)] = 1;
        }
    actL = 1;
    for (j = 0; j < N && actL < L; j++) {   /* :581-605 */
        for (k = 0; k < actL; k++) lit_get_llr(s, 0, j, k);
        if (!c->frozen[j]) {
            for (k = 0; k < actL; k++) lit_copy(s, k, k + actL, 1);
            for (k = 0; k < actL; k++) {
                s->b[IX(s, 0, j, k)] = 0;
                s->b[IX(s, 0, j, k + actL)] = 1;
                PM[k + actL] = PM[k] + lit_phi(s, k, j, 1);
                PM[k] = PM[k] + lit_phi(s, k, j, 0);
                lit_update_bit(s, 0, j, k);
                lit_update_bit(s, 0, j, k + actL);
            }
            actL *= 2;
        } else {
            for (k = 0; k < actL; k++) PM[k] += lit_phi(s, k, j, 0);
        }
    }
    for (; j < N; j++) {   /* :606-666 */
        for (k = 0; k < L; k++) lit_get_llr(s, 0, j, k);
        if (!c->frozen[j]) {
            for (k = 0; k < L; k++) {
                cand[k] = PM[k] + lit_phi(s, k, j, 0);
                cand[k + L] = PM[k] + lit_phi(s, k, j, 1);
                PM[k] = cand[k];
                PM[k + L] = cand[k + L];
            }
            int hang = 0;
            lit_quicksort(cand, 0, 2 * L - 1, &hang);
            if (hang) return -5;   /* the reference is still inside Partition() */
            const double med = cand[L];
            if (cand[L - 1] == med) s->diag[0]++;   /* "Oops!" */
            for (k = 0; k < L; k++) {
                if (PM[k] < med && PM[k + L] < med) surviv[k] = 2;
                else if (PM[k] >= med && PM[k + L] < med) surviv[k] = 1;
                else if (PM[k] < med && PM[k + L] >= med) surviv[k] = 0;
                else surviv[k] = -1;
            }
            i = 0;
            for (k = 0; k < L; k++) {
                switch (surviv[k]) {
                case 0:
                    s->b[IX(s, 0, j, k)] = 0;
                    lit_update_bit(s, 0, j, k);
                    break;
                case 1:
                    s->b[IX(s, 0, j, k)] = 1;
                    lit_update_bit(s, 0, j, k);
                    PM[k] = PM[k + L];
                    break;
                case 2:
                    for (; i < L && surviv[i] != -1; i++) {}
                    if (i >= L) { s->diag[2]++; return -3; }   /* "Error!" (the reference would index past surviv[]) */
                    lit_copy(s, k, i, 0);
                    s->b[IX(s, 0, j, k)] = 0;
                    lit_update_bit(s, 0, j, k);
                    s->b[IX(s, 0, j, i)] = 1;
                    lit_update_bit(s, 0, j, i);
                    surviv[i] = -2;
                    PM[i] = PM[k + L];
                    break;
                default:
                    break;   /* -1 / -2: nothing is done for this slot at this leaf */
                }
            }
        } else {
            for (k = 0; k < L; k++) PM[k] += lit_phi(s, k, j, 0);
        }
    }
    int best = 0;
    if (crc && c->r > 0) {   /* CASCL_1024_L8.c:725-755 */
        int pass[PO_MAX_L], first = -1;
        for (k = 0; k < L; k++) {
            pass[k] = lit_crc(s, k);
            if (first < 0 && pass[k]) first = k;
        }
        if (first >= 0) {
            best = first;
            double mn = PM[first];
            for (k = 1; k < L; k++)
                if (pass[k] && PM[k] < mn) { mn = PM[k]; best = k; }
        } else {
            double mn = PM[0];
            for (k = 1; k < L; k++)
                if (PM[k] < mn) { mn = PM[k]; best = k; }
        }
    } else {   /* SCL_1024.c:667-674 */
        double mn = PM[0];
        for (k = 1; k < L; k++)
            if (PM[k] < mn) { mn = PM[k]; best = k; }
    }
    for (j = 0; j < N; j++) u_hat[j] = s->b[IX(s, 0, j, best)];
    if (pm_out) *pm_out = PM[best];
    return 0;
}
