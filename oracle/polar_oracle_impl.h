/*
 * polar_oracle_impl.h -- arithmetic-type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Included twice by polar_oracle.c with REAL = double / float and SFX = f64 / f32.
 * See polar_oracle.h for the parity statement.
 */
#define PO_CAT_(a, b) a##_##b
#define PO_CAT(a, b) PO_CAT_(a, b)
#define FN(name) PO_CAT(name, SFX)
#define PO_ABS(x) ((REAL)fabs((double)(x))) /* exact for float and double */

/* Table T: the 8-level ln(1+e^-x) staircase shared by CHK and PHI (SCL_1024.c:352-359, :490-497). */
static inline REAL FN(tab)(REAL a)
{
    if (a < (REAL)0.196) return (REAL)0.65;
    if (a < (REAL)0.433) return (REAL)0.55;
    if (a < (REAL)0.71) return (REAL)0.45;
    if (a < (REAL)1.05) return (REAL)0.35;
    if (a < (REAL)1.508) return (REAL)0.25;
    if (a < (REAL)2.252) return (REAL)0.15;
    if (a < (REAL)4.5) return (REAL)0.05;
    return (REAL)0;
}

/* Check node (SCL_1024.c:343-374, identical in all reference programs). */
static inline REAL FN(chk)(REAL L1, REAL L2)
{
    REAL sAbs = PO_ABS(L1 + L2), dAbs = PO_ABS(L1 - L2);
    REAL delta = FN(tab)(sAbs);
    delta -= FN(tab)(dAbs);
    REAL A1 = PO_ABS(L1);
    REAL A2 = PO_ABS(L2);
    int s = ((L1 >= 0) ? 1 : -1) * ((L2 >= 0) ? 1 : -1);
    if (A1 > A2) return (REAL)s * A2 + delta;
    return (REAL)s * A1 + delta;
}

/* Path-metric increment (SCL_1024.c:481-502, updatePM.pdf). */
static inline REAL FN(phi)(REAL lam, int u)
{
    REAL absL = PO_ABS(lam);
    REAL result = FN(tab)(absL);
    if ((u == 0 && lam < 0) || (u == 1 && lam > 0)) result += absL;
    return result;
}

/* element-wise CHK / T / PHI on caller-chosen operands (tests of the kernels' table arithmetic at its boundaries):
 * op 0 CHK(a, b), 1 T(|a|), 2 PHI(a, u = (b != 0)) */
void FN(po_math)(int op, const REAL *a, const REAL *b, REAL *out, size_t cnt)
{
    for (size_t i = 0; i < cnt; i++) {
        if (op == 0) out[i] = FN(chk)(a[i], b[i]);
        else if (op == 1) out[i] = FN(tab)(PO_ABS(a[i]));
        else out[i] = FN(phi)(a[i], b[i] != 0);
    }
}

/* ------------------------------------------------------------------------------------------
 * SC (SC_128.c:395-460): natural-order recursion.  alpha level t lives at
 * offset 2^t of a length-N scratch; bl = saved left-child partial sums, same layout.
 * ---------------------------------------------------------------------------------------- */
int FN(po_sc_decode)(const po_code *c, const REAL *llr, int *u_hat)
{
    const int N = c->N, n = c->n;
    REAL *alpha = (REAL *)malloc(sizeof(REAL) * (size_t)N);
    unsigned char *bl = (unsigned char *)malloc((size_t)N);
    unsigned char *cur = (unsigned char *)malloc((size_t)N);
    unsigned char *nxt = (unsigned char *)malloc((size_t)N);
    if (!alpha || !bl || !cur || !nxt) return -1;
    for (int j = 0; j < N; j++) {
        int tf;
        if (j == 0) {
            tf = n - 1;
        } else {
            int d = __builtin_ctz((unsigned)j);
            const REAL *src = (d + 1 == n) ? llr : alpha + (1 << (d + 1));
            REAL *out = alpha + (1 << d);
            const unsigned char *b = bl + (1 << d);
            int h = 1 << d;
            for (int i = 0; i < h; i++) /* getLLR lower node, SC_128.c:355-359 */
                out[i] = b[i] ? src[i + h] - src[i] : src[i + h] + src[i];
            tf = d - 1;
        }
        for (int t = tf; t >= 0; t--) { /* getLLR upper node, SC_128.c:353-354 */
            const REAL *src = (t + 1 == n) ? llr : alpha + (1 << (t + 1));
            REAL *out = alpha + (1 << t);
            int h = 1 << t;
            for (int i = 0; i < h; i++) out[i] = FN(chk)(src[i], src[i + h]);
        }
        REAL lam = alpha[1];
        int bit = c->frozen[j] ? 0 : (lam >= 0 ? 0 : 1); /* SC_128.c:426-431 */
        u_hat[j] = bit;
        /* updateBit (SC_128.c:368-392) restated as bottom-up partial-sum combine */
        cur[0] = (unsigned char)bit;
        int t = 0;
        while (t < n && ((j >> t) & 1)) {
            int h = 1 << t;
            const unsigned char *l = bl + h;
            for (int i = 0; i < h; i++) {
                nxt[i] = l[i] ^ cur[i];
                nxt[i + h] = cur[i];
            }
            unsigned char *tmp = cur; cur = nxt; nxt = tmp;
            t++;
        }
        if (t < n) memcpy(bl + (1 << t), cur, (size_t)1 << t);
    }
    free(alpha); free(bl); free(cur); free(nxt);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * BP (BP_1024.c:372-427): flooding schedule, iters round trips, no early stop (Appendix A.4).
 * l, r: [n+1][N].  Operand order inside the sums and CHK calls kept as in the reference.
 * ---------------------------------------------------------------------------------------- */
/* BPr (BPr_128.c:373-575): the same flooding BP, and after the iterations listed in cp[] (1-based counts,
 * reference: 3, 6, 10, 20, 40, 80 of iterMax = 90, :18-23) a read-out per stage i = 0..n (:417-438): hard
 * decisions of l+r at stage i, carried back to the u side through the inverse butterflies of stages i-1..0,
 * compared with the sent bits on the information set; E[c][i] += mismatches.  cp == NULL: plain BP. */
int FN(po_bpr_decode)(const po_code *c, const REAL *llr, int iters, const int *cp, int ncp, const int *u,
                      long *E /* [ncp][n+1], accumulated */, int *u_hat)
{
    const int N = c->N, n = c->n;
    REAL *l = (REAL *)calloc((size_t)(n + 1) * N, sizeof(REAL));
    REAL *r = (REAL *)calloc((size_t)(n + 1) * N, sizeof(REAL));
    if (!l || !r) return -1;
#define Lm(i, j) l[(size_t)(i) * N + (j)]
#define Rm(i, j) r[(size_t)(i) * N + (j)]
    for (int j = 0; j < N; j++) Lm(n, j) = llr[j];
    for (int j = 0; j < N; j++) Rm(0, j) = c->frozen[j] ? (REAL)999 : (REAL)0;
    for (int it = 0; it < iters; it++) {
        for (int i = 0; i < n; i++) { /* R sweep, BP_1024.c:395-404 */
            int s = 1 << i;
            for (int j = 0; j < N; j++) {
                if (j & s) continue;
                REAL a = FN(chk)(Rm(i, j), Lm(i + 1, j + s) + Rm(i, j + s));
                REAL b = Rm(i, j + s) + FN(chk)(Rm(i, j), Lm(i + 1, j));
                Rm(i + 1, j) = a;
                Rm(i + 1, j + s) = b;
            }
        }
        for (int i = n - 1; i >= 0; i--) { /* L sweep, BP_1024.c:406-415 */
            int s = 1 << i;
            for (int j = 0; j < N; j++) {
                if (j & s) continue;
                REAL a = FN(chk)(Lm(i + 1, j), Lm(i + 1, j + s) + Rm(i, j + s));
                REAL b = Lm(i + 1, j + s) + FN(chk)(Rm(i, j), Lm(i + 1, j));
                Lm(i, j) = a;
                Lm(i, j + s) = b;
            }
        }
        for (int q = 0; q < ncp; q++) {
            if (cp[q] != it + 1) continue;
            int *b = (int *)malloc(sizeof(int) * (size_t)N);
            for (int i = 0; i <= n; i++) {
                for (int j = 0; j < N; j++) b[j] = (Lm(i, j) + Rm(i, j) >= 0) ? 0 : 1;
                for (int k = i; k > 0; k--) { /* :423-429: left = (upper ^ lower, lower) of the stage k-1 butterflies */
                    int s = 1 << (k - 1);
                    for (int j = 0; j < N; j++)
                        if (!(j & s)) b[j] = (b[j + s] + b[j]) % 2;
                }
                for (int j = 0; j < c->K; j++)
                    if (b[c->info_order[j]] != u[c->info_order[j]]) E[(size_t)q * (n + 1) + i] += 1;
            }
            free(b);
        }
    }
    for (int j = 0; j < N; j++) /* BP_1024.c:417-425 */
        u_hat[j] = c->frozen[j] ? 0 : ((Lm(0, j) + Rm(0, j) >= 0) ? 0 : 1);
#undef Lm
#undef Rm
    free(l); free(r);
    return 0;
}

int FN(po_bp_decode)(const po_code *c, const REAL *llr, int iters, int *u_hat)
{
    return FN(po_bpr_decode)(c, llr, iters, NULL, 0, NULL, NULL, u_hat);
}

/* ------------------------------------------------------------------------------------------
 * SCL / CA-SCL (SCL_1024.c:547-680, CASCL_1024_L8.c:601-761), lock-step list, Appendix A.3.
 * copyPath/simpleCopy (SCL_1024.c:451-478) are replaced by a per-level pointer table: every
 * slot owns a buffer per level; a clone copies the table; writes go to the slot's own buffer.
 * ---------------------------------------------------------------------------------------- */
typedef struct FN(scl_ws) {
    int N, n, L;
    REAL *alpha;        /* [L][N]  level t at offset 2^t                 */
    unsigned char *bl;  /* [L][N]  saved left partial sums, same layout  */
    unsigned char *uh;  /* [L][N]  decided bits                          */
    unsigned char *cur, *nxt; /* [N] scratch                             */
    int *ptr;           /* [L][n]                                        */
    REAL PM[2 * PO_MAX_L], cand[2 * PO_MAX_L];
    int surviv[PO_MAX_L];
} FN(scl_ws);

static const REAL *FN(scl_src)(const FN(scl_ws) *w, const REAL *llr, int k, int t)
{
    if (t == w->n) return llr;
    return w->alpha + (size_t)w->ptr[k * w->n + t] * w->N + ((size_t)1 << t);
}

/* getLLR(V[0][j], k) (SCL_1024.c:404-421) for leaf j of path k; returns lambda */
static REAL FN(scl_leaf_llr)(FN(scl_ws) *w, const REAL *llr, int k, int j)
{
    const int n = w->n;
    REAL *own = w->alpha + (size_t)k * w->N;
    int tf;
    if (j == 0) {
        tf = n - 1;
    } else {
        int d = __builtin_ctz((unsigned)j);
        const REAL *src = FN(scl_src)(w, llr, k, d + 1);
        REAL *out = own + (1 << d);
        const unsigned char *b = w->bl + (size_t)k * w->N + (1 << d);
        int h = 1 << d;
        for (int i = 0; i < h; i++)
            out[i] = b[i] ? src[i + h] - src[i] : src[i + h] + src[i];
        w->ptr[k * n + d] = k;
        tf = d - 1;
    }
    for (int t = tf; t >= 0; t--) {
        const REAL *src = FN(scl_src)(w, llr, k, t + 1);
        REAL *out = own + (1 << t);
        int h = 1 << t;
        for (int i = 0; i < h; i++) out[i] = FN(chk)(src[i], src[i + h]);
        w->ptr[k * n + t] = k;
    }
    return own[1];
}

/* updateBit (SCL_1024.c:424-448) restated: fold decided bit into the saved partial sums */
static void FN(scl_set_bit)(FN(scl_ws) *w, int k, int j, int bit)
{
    const int n = w->n;
    unsigned char *bl = w->bl + (size_t)k * w->N;
    unsigned char *cur = w->cur, *nxt = w->nxt;
    w->uh[(size_t)k * w->N + j] = (unsigned char)bit;
    cur[0] = (unsigned char)bit;
    int t = 0;
    while (t < n && ((j >> t) & 1)) {
        int h = 1 << t;
        const unsigned char *l = bl + h;
        for (int i = 0; i < h; i++) {
            nxt[i] = l[i] ^ cur[i];
            nxt[i + h] = cur[i];
        }
        unsigned char *tmp = cur; cur = nxt; nxt = tmp;
        t++;
    }
    if (t < n) memcpy(bl + (1 << t), cur, (size_t)1 << t);
}

/* copyPath / simpleCopy (SCL_1024.c:451-478): slot dst becomes a clone of slot src before leaf j is decided */
static void FN(scl_clone)(FN(scl_ws) *w, int src, int dst, int j)
{
    memcpy(w->ptr + dst * w->n, w->ptr + src * w->n, sizeof(int) * (size_t)w->n);
    memcpy(w->bl + (size_t)dst * w->N, w->bl + (size_t)src * w->N, (size_t)w->N);
    memcpy(w->uh + (size_t)dst * w->N, w->uh + (size_t)src * w->N, (size_t)j);
}

static inline uint32_t FN(key32)(REAL v)
{
    if (sizeof(REAL) == 4) { float f = (float)v; uint32_t u; memcpy(&u, &f, 4); return u; }
    double d = (double)v; uint64_t u; memcpy(&u, &d, 8); return (uint32_t)(u >> 32);
}

static int FN(cmp_real)(const void *a, const void *b)
{
    REAL x = *(const REAL *)a, y = *(const REAL *)b;
    return (x > y) - (x < y);
}

int FN(po_scl_decode)(const po_code *c, const REAL *llr, int L, int crc, int *u_hat,
                      REAL *pm_out, int *ties_out)
{
    const int N = c->N, n = c->n;
    if (L < 1 || L > PO_MAX_L || (L & (L - 1))) return -2;
    FN(scl_ws) w;
    w.N = N; w.n = n; w.L = L;
    w.alpha = (REAL *)malloc(sizeof(REAL) * (size_t)L * N);
    w.bl = (unsigned char *)calloc((size_t)L * N, 1);
    w.uh = (unsigned char *)calloc((size_t)L * N, 1);
    w.cur = (unsigned char *)malloc((size_t)N);
    w.nxt = (unsigned char *)malloc((size_t)N);
    w.ptr = (int *)calloc((size_t)L * n, sizeof(int));
    if (!w.alpha || !w.bl || !w.uh || !w.cur || !w.nxt || !w.ptr) return -1;
    REAL *PM = w.PM, *cand = w.cand;
    int *surviv = w.surviv;
    REAL lam[PO_MAX_L];
    int ties = 0;
    int act = 1;
    po_last_key_fallbacks = 0;
    po_last_triples = 0;
    po_last_phase2 = 0;
    po_last_trivial = 0;
    po_last_trivial_bound = 0;
    po_last_k1 = 0;
    PM[0] = 0; /* SCL_1024.c:556 */
    for (int j = 0; j < N; j++) {
        for (int k = 0; k < act; k++) lam[k] = FN(scl_leaf_llr)(&w, llr, k, j);
        if (c->frozen[j]) { /* SCL_1024.c:601-604, :662-665 */
            for (int k = 0; k < act; k++) {
                PM[k] += FN(phi)(lam[k], 0);
                FN(scl_set_bit)(&w, k, j, 0);
            }
        } else if (act < L) { /* phase 1: doubling, SCL_1024.c:586-600 */
            for (int k = 0; k < act; k++) FN(scl_clone)(&w, k, k + act, j);
            for (int k = 0; k < act; k++) {
                PM[k + act] = PM[k] + FN(phi)(lam[k], 1);
                PM[k] = PM[k] + FN(phi)(lam[k], 0);
                FN(scl_set_bit)(&w, k, j, 0);
                FN(scl_set_bit)(&w, k + act, j, 1);
            }
            act *= 2;
        } else { /* phase 2: prune, SCL_1024.c:610-661 */
            {   /* instrumentation: the sufficient test on bounds (see po_scl_last_prune_stats4) */
                REAL mxpm = PM[0];
                for (int k = 1; k < L; k++) if (PM[k] > mxpm) mxpm = PM[k];
                const uint32_t bkey = FN(key32)(mxpm + (REAL)0.65);
                int ok = 1;
                for (int k = 0; k < L; k++) ok &= (bkey < FN(key32)(PM[k] + PO_ABS(lam[k])));
                po_last_trivial_bound += ok;
            }
            for (int k = 0; k < L; k++) {
                cand[k] = PM[k] + FN(phi)(lam[k], 0);
                cand[k + L] = PM[k] + FN(phi)(lam[k], 1);
                PM[k] = cand[k];
                PM[k + L] = cand[k + L];
            }
            { /* instrumentation for the tests of the kernels' two-step ranking (32-bit keys first: the float's bits /
               * the double's high word, full width only if those do not single out exactly L candidates) */
                int keep = 0;
                for (int a = 0; a < 2 * L; a++) {
                    int cnt = 0;
                    for (int m = 0; m < 2 * L; m++) cnt += (FN(key32)(cand[m]) <= FN(key32)(cand[a]));
                    keep += (cnt <= L);
                }
                if (keep != L) po_last_key_fallbacks++;
                /* leaves where every path simply keeps its better branch, decided on the 32-bit keys alone */
                uint32_t mx = 0, mn = 0xffffffffu;
                for (int a = 0; a < L; a++) {
                    uint32_t k0 = FN(key32)(cand[a]), k1 = FN(key32)(cand[a + L]);
                    uint32_t lo = k0 < k1 ? k0 : k1, hi = k0 < k1 ? k1 : k0;
                    if (lo > mx) mx = lo;
                    if (hi < mn) mn = hi;
                }
                po_last_phase2++;
                if (mx < mn) po_last_trivial++;
            }
            qsort(cand, (size_t)2 * L, sizeof(REAL), FN(cmp_real)); /* QuickSort, :619 */
            REAL med = cand[L];
            if (cand[L - 1] == med) ties++; /* the reference prints "Oops!" (:621-622) */
            { /* three or more keys equal to the pivot value somewhere: see polar_oracle_literal.c (Partition) */
                int run = 1;
                for (int a = 1; a < 2 * L; a++) {
                    run = (cand[a] == cand[a - 1]) ? run + 1 : 1;
                    if (run >= 3) { po_last_triples++; break; }
                }
            }
            for (int k = 0; k < L; k++) { /* :624-633 */
                if (PM[k] < med && PM[k + L] < med) surviv[k] = 2;
                else if (PM[k] >= med && PM[k + L] < med) surviv[k] = 1;
                else if (PM[k] < med && PM[k + L] >= med) surviv[k] = 0;
                else surviv[k] = -1;
            }
            {
                int nb = 0;
                for (int k = 0; k < L; k++) nb += (surviv[k] == 2);
                po_last_k1 += (nb == 1);
            }
            int i = 0;
            for (int k = 0; k < L; k++) { /* :636-661 */
                switch (surviv[k]) {
                case 0:
                    FN(scl_set_bit)(&w, k, j, 0);
                    break;
                case 1:
                    FN(scl_set_bit)(&w, k, j, 1);
                    PM[k] = PM[k + L];
                    break;
                case 2:
                    for (; i < L && surviv[i] != -1; i++) {}
                    if (i >= L) { /* cannot happen: #both == #dead unless tie, and then #dead > #both */
                        free(w.alpha); free(w.bl); free(w.uh); free(w.cur); free(w.nxt); free(w.ptr);
                        return -3;
                    }
                    FN(scl_clone)(&w, k, i, j);
                    FN(scl_set_bit)(&w, k, j, 0);
                    FN(scl_set_bit)(&w, i, j, 1);
                    surviv[i] = -2;
                    PM[i] = PM[k + L];
                    break;
                default:
                    break;
                }
            }
            /* Tie rule of this build (DESIGN.md "median ties"): a dead slot that was not refilled
             * continues as its own 0-branch with PM = c0 (the reference leaves PM[k] = c0 too, but its
             * node flags go stale).  Never reached in fp64 on AWGN inputs. */
            for (int k = 0; k < L; k++)
                if (surviv[k] == -1) FN(scl_set_bit)(&w, k, j, 0);
        }
    }
    /* selection */
    int best = 0;
    if (crc && c->r > 0) { /* CASCL_1024_L8.c:725-755 */
        int first = -1;
        int pass[PO_MAX_L];
        int *cw = (int *)malloc(sizeof(int) * (size_t)c->A);
        for (int k = 0; k < act; k++) {
            for (int i = 0; i < c->A; i++) cw[i] = w.uh[(size_t)k * N + c->info_order[i]];
            pass[k] = po_crc_check(c, cw);
            if (first < 0 && pass[k]) first = k;
        }
        free(cw);
        if (first >= 0) {
            best = first;
            REAL mn = PM[first];
            for (int k = 1; k < act; k++)
                if (pass[k] && PM[k] < mn) { mn = PM[k]; best = k; }
        } else {
            REAL mn = PM[0];
            for (int k = 1; k < act; k++)
                if (PM[k] < mn) { mn = PM[k]; best = k; }
        }
    } else { /* SCL_1024.c:667-674 */
        REAL mn = PM[0];
        for (int k = 1; k < act; k++)
            if (PM[k] < mn) { mn = PM[k]; best = k; }
    }
    for (int j = 0; j < N; j++) u_hat[j] = w.uh[(size_t)best * N + j];
    if (pm_out) *pm_out = PM[best];
    if (ties_out) *ties_out = ties;
    free(w.alpha); free(w.bl); free(w.uh); free(w.cur); free(w.nxt); free(w.ptr);
    return 0;
}

#undef PO_ABS
#undef FN
#undef PO_CAT
#undef PO_CAT_
