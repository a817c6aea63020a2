/*
 * oracle/cscore_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A CPU restatement (plain C99 + pthreads, flat arrays, no Python.h) of the
 * algorithm in the reference's only native component,
 * /root/reference/motifscan/motif/cscore.c.  It exists so that the HIP path in
 * motifscan_amd/csrc can be checked bit-for-bit on any box (the reference tree
 * does not travel to the GPU box) and so bench.py has a CPU baseline when
 * oracle/_ref (the real cscore.c, built in the build container and shipped as a
 * binary) is absent: "cpu_baseline.kind" is "reference" when bench.py timed
 * oracle/_ref and "port" only when it had to fall back to this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (motifscan_amd/) must never call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against
 *   (a) the literal known answers of the reference's own tests
 *       (tests/test_motif_score.py:6-32), and
 *   (b) golden vectors produced by the real reference (cscore.c compiled
 *       unmodified into oracle/_ref/, driven by tests/golden/make_golden.py),
 * bit-for-bit on scores, positions, strands and order.
 *
 * Semantics restated (reference file:line):
 *   base codes       cscore.c:81-114   A/a 0, C/c 1, G/g 2, T/t 3, anything else -1
 *   max_raw_score    cscore.c:36-48    sum over columns of max(0, column max), left to right, fp64
 *   score kernel     cscore.c:174-229  first W bases only, fwd / rev / max(fwd, rev), divided by max_raw
 *   scan kernel      cscore.c:317-397  every start j in [0, L-W], both strands accumulated in column
 *                                      order c = 0..W-1, fwd uses M[row][c], rev uses M[3-row][W-1-c],
 *                                      non-ACGT adds nothing, score/max_raw, hit iff score-cutoff >= -1e-10,
 *                                      '+' (1) emitted before '-' (2) at the same j
 *   hit order        cscore.c:336-390  per PWM: sequence ascending, start ascending, '+' before '-'
 *   threading        cscore.c:181-186, 323-328  work queue whose unit is one whole PWM
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- codes -- */

/* cscore.c:92-111 */
static inline int8_t base_code(char ch) {
    switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
    }
}

void oracle_encode(const char *bases, int64_t n, int8_t *codes) {
    for (int64_t i = 0; i < n; i++) codes[i] = base_code(bases[i]);
}

/* cscore.c:36-48.  m is one PWM, row-major [4][w]. */
double oracle_max_raw_score(const double *m, int32_t w) {
    double total = 0;
    for (int32_t c = 0; c < w; c++) {
        double best = 0;           /* initialised to 0, NOT to the first entry */
        for (int r = 0; r < 4; r++)
            if (m[(int64_t)r * w + c] > best) best = m[(int64_t)r * w + c];
        total += best;
    }
    return total;
}

/* --------------------------------------------------------------- shared -- */

typedef struct {
    int64_t seq;
    int64_t pos;
    double score;
    int32_t strand;
} hit_t;

typedef struct {
    hit_t *v;
    int64_t n, cap;
} hitvec_t;

static int hitvec_push(hitvec_t *h, int64_t seq, int64_t pos, double score, int32_t strand) {
    if (h->n == h->cap) {
        int64_t ncap = h->cap ? h->cap * 2 : 64;
        hit_t *nv = (hit_t *) realloc(h->v, (size_t) ncap * sizeof(hit_t));
        if (!nv) return -1;
        h->v = nv;
        h->cap = ncap;
    }
    hit_t *e = &h->v[h->n++];
    e->seq = seq; e->pos = pos; e->score = score; e->strand = strand;
    return 0;
}

typedef struct {
    /* inputs */
    const double *vals;        /* concatenated PWMs, each row-major [4][w] */
    const int64_t *val_off;    /* P+1 offsets into vals (in doubles) */
    const int32_t *widths;     /* P */
    const double *cutoffs;     /* P (scan only) */
    int32_t P;
    const int8_t *codes;       /* concatenated base codes */
    const int64_t *seq_off;    /* R+1 */
    int64_t R;
    int strand;
    /* work queue (cscore.c:181-186) */
    pthread_mutex_t mu;
    int32_t next;
    /* outputs */
    hitvec_t *hits;            /* P (scan) */
    double *scores;            /* P*R (score) */
    int failed;
} job_t;

static int32_t job_take(job_t *jb) {
    pthread_mutex_lock(&jb->mu);
    int32_t k = jb->next < jb->P ? jb->next++ : -1;
    pthread_mutex_unlock(&jb->mu);
    return k;
}

/* ----------------------------------------------------------------- scan -- */

/* cscore.c:336-390 for one PWM */
static int scan_one(job_t *jb, int32_t p) {
    const int32_t w = jb->widths[p];
    const double *m = jb->vals + jb->val_off[p];
    const double *row[4] = {m, m + w, m + 2 * (int64_t) w, m + 3 * (int64_t) w};
    const double maxraw = oracle_max_raw_score(m, w);
    const double cutoff = jb->cutoffs[p];
    const int strand = jb->strand;
    hitvec_t *out = &jb->hits[p];

    for (int64_t i = 0; i < jb->R; i++) {
        const int8_t *s = jb->codes + jb->seq_off[i];
        const int64_t len = jb->seq_off[i + 1] - jb->seq_off[i];
        if (len < w) continue;                             /* cscore.c:337-339 */
        for (int64_t j = 0; j + w <= len; j++) {
            double fwd = 0, rev = 0;
            for (int32_t c = 0; c < w; c++) {
                int8_t r = s[j + c];
                if (r != -1) {
                    if (strand & 1) fwd += row[r][c];
                    if (strand & 2) rev += row[3 - r][w - 1 - c];
                }
            }
            if (strand & 1) {
                fwd = fwd / maxraw;
                if (fwd - cutoff >= -1e-10)
                    if (hitvec_push(out, i, j, fwd, 1)) return -1;
            }
            if (strand & 2) {
                rev = rev / maxraw;
                if (rev - cutoff >= -1e-10)
                    if (hitvec_push(out, i, j, rev, 2)) return -1;
            }
        }
    }
    return 0;
}

static void *scan_worker(void *arg) {
    job_t *jb = (job_t *) arg;
    for (;;) {
        int32_t p = job_take(jb);
        if (p < 0) break;
        if (scan_one(jb, p)) { jb->failed = 1; break; }
    }
    return NULL;
}

/* ---------------------------------------------------------------- score -- */

/* cscore.c:191-224 for one PWM.  The reference reads the first W codes with no
 * length check (undefined behaviour when a sequence is shorter than W); the
 * restatement treats the missing bases as non-ACGT (they add nothing). */
static void score_one(job_t *jb, int32_t p) {
    const int32_t w = jb->widths[p];
    const double *m = jb->vals + jb->val_off[p];
    const double *row[4] = {m, m + w, m + 2 * (int64_t) w, m + 3 * (int64_t) w};
    const double maxraw = oracle_max_raw_score(m, w);
    const int strand = jb->strand;
    for (int64_t i = 0; i < jb->R; i++) {
        const int8_t *s = jb->codes + jb->seq_off[i];
        const int64_t len = jb->seq_off[i + 1] - jb->seq_off[i];
        double fwd = 0, rev = 0;
        for (int32_t c = 0; c < w; c++) {
            int8_t r = c < len ? s[c] : (int8_t) -1;
            if (r != -1) {
                if (strand & 1) fwd += row[r][c];
                if (strand & 2) rev += row[3 - r][w - 1 - c];
            }
        }
        double sc = 0;
        switch (strand) {
            case 1: sc = fwd; break;
            case 2: sc = rev; break;
            case 3: sc = fwd > rev ? fwd : rev; break;
        }
        jb->scores[(int64_t) p * jb->R + i] = sc / maxraw;
    }
}

static void *score_worker(void *arg) {
    job_t *jb = (job_t *) arg;
    for (;;) {
        int32_t p = job_take(jb);
        if (p < 0) break;
        score_one(jb, p);
    }
    return NULL;
}

/* --------------------------------------------------------------- driver -- */

static int run_threads(job_t *jb, void *(*fn)(void *), int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads == 1) { fn(jb); return jb->failed ? -1 : 0; }
    pthread_t *th = (pthread_t *) malloc(sizeof(pthread_t) * (size_t) n_threads);
    if (!th) return -1;
    int started = 0;
    for (; started < n_threads; started++)
        if (pthread_create(&th[started], NULL, fn, jb) != 0) break;
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    free(th);
    return (started == n_threads && !jb->failed) ? 0 : -1;
}

static int job_init(job_t *jb, const double *vals, const int32_t *widths, const double *cutoffs,
                    int32_t P, const char *bases, const int64_t *seq_off, int64_t R, int strand,
                    int64_t **val_off_out, int8_t **codes_out) {
    memset(jb, 0, sizeof(*jb));
    int64_t *val_off = (int64_t *) malloc(sizeof(int64_t) * ((size_t) P + 1));
    int64_t n_bases = R > 0 ? seq_off[R] : 0;
    int8_t *codes = (int8_t *) malloc((size_t) (n_bases > 0 ? n_bases : 1));
    if (!val_off || !codes) { free(val_off); free(codes); return -1; }
    val_off[0] = 0;
    for (int32_t p = 0; p < P; p++) val_off[p + 1] = val_off[p] + 4 * (int64_t) widths[p];
    oracle_encode(bases, n_bases, codes);
    jb->vals = vals; jb->val_off = val_off; jb->widths = widths; jb->cutoffs = cutoffs; jb->P = P;
    jb->codes = codes; jb->seq_off = seq_off; jb->R = R; jb->strand = strand;
    pthread_mutex_init(&jb->mu, NULL);
    *val_off_out = val_off;
    *codes_out = codes;
    return 0;
}

/*
 * Scan.  Inputs: P PWMs concatenated (each row-major [4][w_p]), cutoffs[P],
 * R sequences concatenated as ASCII with offsets[R+1].  Outputs (malloc'd,
 * release with oracle_free): motif_off[P+1] and SoA hit arrays in the
 * reference's order.  Returns 0 on success.
 */
int oracle_scan(const double *vals, const int32_t *widths, const double *cutoffs, int32_t P,
                const char *bases, const int64_t *seq_off, int64_t R, int strand, int n_threads,
                int64_t **motif_off, int64_t **hit_seq, int64_t **hit_pos, double **hit_score,
                int32_t **hit_strand) {
    job_t jb;
    int64_t *val_off; int8_t *codes;
    if (job_init(&jb, vals, widths, cutoffs, P, bases, seq_off, R, strand, &val_off, &codes)) return -1;
    jb.hits = (hitvec_t *) calloc((size_t) (P > 0 ? P : 1), sizeof(hitvec_t));
    int rc = jb.hits ? run_threads(&jb, scan_worker, n_threads) : -1;

    int64_t *off = (int64_t *) malloc(sizeof(int64_t) * ((size_t) P + 1));
    int64_t total = 0;
    if (rc == 0 && off) {
        off[0] = 0;
        for (int32_t p = 0; p < P; p++) { total += jb.hits[p].n; off[p + 1] = total; }
    } else rc = -1;
    size_t n = (size_t) (total > 0 ? total : 1);
    int64_t *hs = (int64_t *) malloc(n * sizeof(int64_t));
    int64_t *hp = (int64_t *) malloc(n * sizeof(int64_t));
    double *hv = (double *) malloc(n * sizeof(double));
    int32_t *hd = (int32_t *) malloc(n * sizeof(int32_t));
    if (rc == 0 && hs && hp && hv && hd) {
        int64_t k = 0;
        for (int32_t p = 0; p < P; p++)
            for (int64_t i = 0; i < jb.hits[p].n; i++, k++) {
                hs[k] = jb.hits[p].v[i].seq; hp[k] = jb.hits[p].v[i].pos;
                hv[k] = jb.hits[p].v[i].score; hd[k] = jb.hits[p].v[i].strand;
            }
        *motif_off = off; *hit_seq = hs; *hit_pos = hp; *hit_score = hv; *hit_strand = hd;
    } else {
        rc = -1;
        free(off); free(hs); free(hp); free(hv); free(hd);
    }
    if (jb.hits) { for (int32_t p = 0; p < P; p++) free(jb.hits[p].v); free(jb.hits); }
    pthread_mutex_destroy(&jb.mu);
    free(val_off); free(codes);
    return rc;
}

/* Score: scores_out is caller-allocated [P][R]. */
int oracle_score(const double *vals, const int32_t *widths, int32_t P, const char *bases,
                 const int64_t *seq_off, int64_t R, int strand, int n_threads, double *scores_out) {
    job_t jb;
    int64_t *val_off; int8_t *codes;
    if (job_init(&jb, vals, widths, NULL, P, bases, seq_off, R, strand, &val_off, &codes)) return -1;
    jb.scores = scores_out;
    int rc = run_threads(&jb, score_worker, n_threads);
    pthread_mutex_destroy(&jb.mu);
    free(val_off); free(codes);
    return rc;
}

void oracle_free(void *p) { free(p); }
