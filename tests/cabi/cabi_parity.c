/*
 * C-only parity check of the drop-in boundary (test infrastructure): a plain C program binds
 * include/motifscan_amd.h exactly as a compiled host (the reference's cscore.c is C) would, scans seeded
 * random input on the GPU and compares every hit -- sequence index, position, strand, fp64 score, order --
 * with the oracle's C restatement of cscore.c:317-476 (oracle/cscore_oracle.c) on the same input.
 *
 * Build + run (tests/test_gpu_parity.py::test_c_program_through_the_cabi does this):
 *   gcc -O2 -std=c99 tests/cabi/cabi_parity.c -Iinclude -Lmotifscan_amd -lmotifscan_amd -Loracle -loracle -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "motifscan_amd.h"

int oracle_scan(const double *vals, const int32_t *widths, const double *cutoffs, int32_t P, const char *bases,
                const int64_t *seq_off, int64_t R, int strand, int n_threads, int64_t **motif_off, int64_t **hit_seq,
                int64_t **hit_pos, double **hit_score, int32_t **hit_strand);
void oracle_free(void *p);

static uint64_t rng_state = 0x9E3779B97F4A7C15ULL;
static uint64_t rnd(void) {                       /* xorshift64* */
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return rng_state * 0x2545F4914F6CDD1DULL;
}
static double unif(void) { return (double) (rnd() >> 11) / 9007199254740992.0; }

#define CHECK(call) do { int rc__ = (call); if (rc__) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, ms_last_error()); return 2; } } while (0)

int main(void) {
    enum { P = 40, R = 300 };
    int32_t widths[P];
    int64_t n_vals = 0;
    for (int p = 0; p < P; p++) { widths[p] = 5 + (int32_t) (rnd() % 30); n_vals += 4 * widths[p]; }   /* 5..34: both paths */
    double *vals = malloc(sizeof(double) * (size_t) n_vals), cutoffs[P];
    int64_t o = 0;
    for (int p = 0; p < P; p++) {
        for (int i = 0; i < 4 * widths[p]; i++) vals[o + i] = round((unif() * 4.0 - 2.6) * 1e5) / 1e5;   /* log-odds-like, 5 decimals */
        for (int c = 0; c < widths[p]; c++) vals[o + (int64_t) (rnd() % 4) * widths[p] + c] = round((0.5 + unif() * 1.5) * 1e5) / 1e5;
        cutoffs[p] = 0.25 + 0.3 * unif();
        o += 4 * widths[p];
    }
    int64_t seq_off[R + 1];
    seq_off[0] = 0;
    for (int r = 0; r < R; r++) seq_off[r + 1] = seq_off[r] + (int64_t) (rnd() % 700);
    const int64_t n_bases = seq_off[R];
    char *bases = malloc((size_t) n_bases + 1);
    for (int64_t i = 0; i < n_bases; i++) bases[i] = "ACGTacgtNR"[rnd() % 100 < 96 ? rnd() % 8 : 8 + rnd() % 2];
    bases[n_bases] = 0;

    int ndev = 0;
    CHECK(ms_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
    CHECK(ms_set_device(0));
    ms_pwmset *pw = NULL; ms_seqset *sq = NULL;
    CHECK(ms_pwmset_create(vals, widths, cutoffs, P, &pw));
    CHECK(ms_seqset_create(bases, seq_off, R, 0, &sq));
    long long total = 0;
    for (int strand = 1; strand <= 3; strand++) {
        ms_result *res = NULL;
        CHECK(ms_scan(pw, sq, strand, 0, &res));
        int64_t n = 0, off[P + 1];
        CHECK(ms_result_num_hits(res, &n));
        CHECK(ms_result_motif_offsets(res, off));
        int64_t *seq = malloc(8 * (size_t) (n + 1)), *pos = malloc(8 * (size_t) (n + 1));
        double *score = malloc(8 * (size_t) (n + 1));
        int8_t *sd = malloc((size_t) n + 1);
        CHECK(ms_result_hits(res, seq, pos, score, sd));

        int64_t *w_off, *w_seq, *w_pos; double *w_score; int32_t *w_sd;
        if (oracle_scan(vals, widths, cutoffs, P, bases, seq_off, R, strand, 4, &w_off, &w_seq, &w_pos, &w_score, &w_sd)) { fprintf(stderr, "oracle failed\n"); return 4; }
        if (w_off[P] != n) { fprintf(stderr, "strand %d: %lld hits, oracle %lld\n", strand, (long long) n, (long long) w_off[P]); return 1; }
        for (int p = 0; p <= P; p++) if (off[p] != w_off[p]) { fprintf(stderr, "strand %d: motif offset %d differs\n", strand, p); return 1; }
        for (int64_t i = 0; i < n; i++)
            if (seq[i] != w_seq[i] || pos[i] != w_pos[i] || sd[i] != w_sd[i] || memcmp(&score[i], &w_score[i], 8) != 0) {
                fprintf(stderr, "strand %d: hit %lld differs\n", strand, (long long) i);
                return 1;
            }
        total += n;
        oracle_free(w_off); oracle_free(w_seq); oracle_free(w_pos); oracle_free(w_score); oracle_free(w_sd);
        free(seq); free(pos); free(score); free(sd);
        ms_result_free(res);
    }
    ms_seqset_free(sq);
    ms_pwmset_free(pw);
    free(vals); free(bases);
    printf("cabi_parity: %lld hits identical over strands 1, 2, 3 (version %d)\n", total, ms_version());
    return total > 1000 ? 0 : 5;
}
