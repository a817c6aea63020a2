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

    /* ---- batch stream (ms_stream_*): the same regions in three batches, pinned input, compact copy-out ---------------- */
    long long streamed = 0;
    {
        char *pin = NULL;
        CHECK(ms_host_alloc((size_t) n_bases + 1, (void **) &pin));
        memcpy(pin, bases, (size_t) n_bases);
        int64_t *w_off, *w_seq, *w_pos; double *w_score; int32_t *w_sd;
        if (oracle_scan(vals, widths, cutoffs, P, bases, seq_off, R, 3, 4, &w_off, &w_seq, &w_pos, &w_score, &w_sd)) return 4;
        ms_stream *st = NULL;
        CHECK(ms_stream_create(pw, 3, MS_STREAM_PACKED, 2, &st));
        const int cut[4] = {0, 90, 91, R};
        int64_t boff[3][R + 1];
        for (int b = 0; b < 3; b++) {
            for (int r = cut[b]; r <= cut[b + 1]; r++) boff[b][r - cut[b]] = seq_off[r] - seq_off[cut[b]];
            CHECK(ms_stream_submit(st, pin + seq_off[cut[b]], boff[b], cut[b + 1] - cut[b]));
        }
        int64_t seen[P];                           /* hits of motif p already matched (batches arrive in order) */
        for (int p = 0; p < P; p++) seen[p] = 0;
        for (int b = 0; b < 3; b++) {
            ms_result *res = NULL;
            CHECK(ms_stream_next(st, &res));
            if (!res) { fprintf(stderr, "stream ran dry\n"); return 1; }
            int64_t n = 0, off[P + 1];
            const uint64_t *coord; const double *score;
            CHECK(ms_result_num_hits(res, &n));
            CHECK(ms_result_motif_offsets(res, off));
            CHECK(ms_result_hits_packed_host(res, &coord, &score));
            for (int p = 0; p < P; p++)
                for (int64_t i = off[p]; i < off[p + 1]; i++) {
                    const int64_t k = w_off[p] + seen[p]++;
                    const int64_t seq = (int64_t) (coord[i] >> 32) + cut[b], pos = (int64_t) ((coord[i] & 0xFFFFFFFFu) >> 1);
                    const int sd = (int) (coord[i] & 1u) + 1;
                    if (k >= w_off[p + 1] || seq != w_seq[k] || pos != w_pos[k] || sd != w_sd[k] || memcmp(&score[i], &w_score[k], 8) != 0) {
                        fprintf(stderr, "stream batch %d motif %d: hit %lld differs\n", b, p, (long long) i);
                        return 1;
                    }
                }
            streamed += n;
            ms_result_free(res);
        }
        for (int p = 0; p < P; p++) if (seen[p] != w_off[p + 1] - w_off[p]) { fprintf(stderr, "stream: motif %d lost hits\n", p); return 1; }
        ms_result *none = (ms_result *) 1;
        CHECK(ms_stream_next(st, &none));
        if (none != NULL) return 1;
        /* the control-set form (cli/scan.py:81-89 -> stats.py:29-31): the same batch counts-only -- its per-motif region counts and hit
         * number are those of the batch that carried its hits out */
        {
            int64_t c_hits[P], c_only[P], n_hits = 0, n_only = 0;
            ms_result *r1 = NULL, *r2 = NULL;
            CHECK(ms_stream_submit(st, pin, boff[0], cut[1] - cut[0]));
            CHECK(ms_stream_submit_counts_only(st, pin, boff[0], cut[1] - cut[0]));
            CHECK(ms_stream_next(st, &r1));
            CHECK(ms_stream_next(st, &r2));
            if (!r1 || !r2) return 1;
            CHECK(ms_result_region_counts(r1, c_hits));
            CHECK(ms_result_region_counts(r2, c_only));
            CHECK(ms_result_num_hits(r1, &n_hits));
            CHECK(ms_result_num_hits(r2, &n_only));
            if (n_hits != n_only || memcmp(c_hits, c_only, sizeof(c_hits)) != 0) { fprintf(stderr, "counts-only batch differs\n"); return 1; }
            ms_result_free(r1); ms_result_free(r2);
        }
        double stage[12];
        uint64_t pool[6];
        CHECK(ms_stream_stats(st, stage));
        CHECK(ms_device_pool_stats(pool));
        for (int k = 0; k < 3; k++)
            if (stage[4 * k] != 5.0 || stage[4 * k + 1] < 0.0) { fprintf(stderr, "stream: stage %d saw %.0f batches\n", k, stage[4 * k]); return 1; }
        if (pool[0] + pool[1] == 0) { fprintf(stderr, "block pool saw no request\n"); return 1; }
        ms_stream_free(st);
        ms_host_free(pin);
        oracle_free(w_off); oracle_free(w_seq); oracle_free(w_pos); oracle_free(w_score); oracle_free(w_sd);
    }

    /* ---- span planning + scan-once for overlapping regions against the per-region scan ----------------------------------- */
    long long once_sites = 0;
    {
        const int64_t chrom_len[2] = {n_bases / 2, n_bases - n_bases / 2};
        int64_t n_spans = 0;
        CHECK(ms_sweep_spans(chrom_len, 2, 200, 50, 5000, NULL, 0, &n_spans));
        ms_span *spans = malloc(sizeof(ms_span) * (size_t) n_spans);
        CHECK(ms_sweep_spans(chrom_len, 2, 200, 50, 5000, spans, n_spans, &n_spans));
        int64_t wins = 0;
        for (int64_t k = 0; k < n_spans; k++) {
            if (spans[k].first_window != wins || spans[k].end - spans[k].begin > 5000) { fprintf(stderr, "bad span %lld\n", (long long) k); return 1; }
            wins += spans[k].n_windows;
        }
        if (wins != (chrom_len[0] - 200) / 50 + 1 + (chrom_len[1] - 200) / 50 + 1) { fprintf(stderr, "span windows do not add up\n"); return 1; }
        free(spans);
        const int64_t goff[3] = {0, chrom_len[0], n_bases};
        ms_genome *gn = NULL;
        CHECK(ms_genome_create(bases, goff, 2, &gn));
        enum { NR = 400 };
        int32_t rc_[NR]; int64_t rs[NR], re[NR];
        for (int r = 0; r < NR; r++) {
            rc_[r] = (int32_t) (rnd() % 2);
            rs[r] = (int64_t) (rnd() % (uint64_t) (chrom_len[rc_[r]] - 10));
            re[r] = rs[r] + (int64_t) (rnd() % 600);
            if (re[r] > chrom_len[rc_[r]]) re[r] = chrom_len[rc_[r]];
        }
        ms_seqset *cut_ = NULL;
        ms_result *a = NULL, *b = NULL;
        CHECK(ms_seqset_from_genome(gn, rc_, rs, re, NR, &cut_));
        CHECK(ms_scan(pw, cut_, 3, 0, &a));
        CHECK(ms_scan_regions_once(pw, gn, rc_, rs, re, NR, 3, 0, &b));
        int64_t na = 0, nb = 0;
        CHECK(ms_result_num_hits(a, &na));
        CHECK(ms_result_num_hits(b, &nb));
        if (na != nb) { fprintf(stderr, "scan-once: %lld sites, per-region scan %lld\n", (long long) nb, (long long) na); return 1; }
        const int64_t *sa, *pa, *sb, *pb; const double *va, *vb; const int8_t *da, *db;
        CHECK(ms_result_hits_host(a, &sa, &pa, &va, &da));
        CHECK(ms_result_hits_host(b, &sb, &pb, &vb, &db));
        if (memcmp(sa, sb, 8 * (size_t) na) || memcmp(pa, pb, 8 * (size_t) na) || memcmp(va, vb, 8 * (size_t) na) || memcmp(da, db, (size_t) na)) {
            fprintf(stderr, "scan-once differs from the per-region scan\n");
            return 1;
        }
        once_sites = na;
        ms_result_free(a); ms_result_free(b);
        ms_seqset_free(cut_);
        ms_genome_free(gn);
    }
    ms_pwmset_free(pw);
    free(vals); free(bases);
    printf("cabi_parity: %lld hits identical over strands 1, 2, 3; %lld through a 3-batch stream (compact form); %lld sites scan-once == per-region "
           "(version %d)\n", total, streamed, once_sites, ms_version());
    return total > 1000 && streamed > 300 && once_sites > 300 ? 0 : 5;
}
