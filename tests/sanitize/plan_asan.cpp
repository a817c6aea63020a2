// tests/sanitize/plan_asan.cpp -- the host side of the pre-filter (motifscan_amd/csrc/ms_plan.cpp: thresholds, the e2m3
// quantiser, paired rows, the row-tile DP, the operand image) under AddressSanitizer + UndefinedBehaviorSanitizer, compiled
// by plain g++ from the SAME source file the library is built from.  CPU build container only (`make -C motifscan_amd/csrc
// sanitize`; tests/test_sanitizers.py runs it, once with the benchmark motif set dumped to a raw file).
//
//     plan_asan [motifs.bin]      motifs.bin: int32 n, int32 widths[n], double cutoffs[n][n_sets], int32 n_sets, double values[...]
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../motifscan_amd/csrc/ms_internal.h"

namespace ms {
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::vfprintf(stderr, fmt, ap);
    va_end(ap);
    std::fputc('\n', stderr);
}
}  // namespace ms

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
    } while (0)

namespace {

double c_max_raw(const double *m, int W) {          // cscore.c:36-48: column maxima start at 0
    double s = 0;
    for (int c = 0; c < W; c++) {
        double mx = 0;
        for (int b = 0; b < 4; b++) if (m[(size_t) b * W + c] > mx) mx = m[(size_t) b * W + c];
        s += mx;
    }
    return s;
}

long n_plans = 0;

void plan_all(const std::vector<double> &values, const std::vector<int32_t> &widths, const std::vector<double> &cutoffs) {
    const int32_t n = (int32_t) widths.size();
    std::vector<int64_t> off((size_t) n + 1, 0);
    for (int32_t p = 0; p < n; p++) off[(size_t) p + 1] = off[(size_t) p] + 4 * (int64_t) widths[(size_t) p];
    CHECK((size_t) off[(size_t) n] == values.size());
    std::vector<double> max_raw((size_t) n);
    for (int32_t p = 0; p < n; p++) max_raw[(size_t) p] = c_max_raw(values.data() + off[(size_t) p], widths[(size_t) p]);
    for (int strand = 1; strand <= 3; strand++)
        for (int pair = 0; pair < 2; pair++)
            for (size_t lds : {(size_t) 8 * 1024, (size_t) 24 * 1024, (size_t) 70 * 1024, (size_t) 139 * 1024}) {
                ms::PrefilterPlan plan;
                const int rc = ms::build_plan(values.data(), off.data(), widths.data(), cutoffs.data(), max_raw.data(), n, strand, lds,
                                              pair != 0, &plan);
                CHECK(rc == MS_OK || rc == MS_ERR_INVALID || rc == MS_ERR_NOMEM);
                if (rc != MS_OK) continue;
                n_plans++;
                CHECK(plan.fast_motifs.size() + plan.exact_motifs.size() == (size_t) n);
                CHECK(plan.group_fields.size() == plan.group_kb.size() * ms::kGroupFields);
                for (int32_t f : plan.group_fields) CHECK(f >= -1 && f < n);
                for (const auto &t : plan.tiles)                          // every tile's operand image lies inside the table buffer
                    CHECK(((size_t) t.table_off16 + t.table_len16) * 16 <= plan.tables.size() * sizeof(uint32_t) && t.n_classes <= ms::kMaxClasses);
            }
}

}  // namespace

int main(int argc, char **argv) {
    std::mt19937_64 rng(20250310);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    // (1) random log-odds-like sets over every width the plan distinguishes (1 ... 70: > 63 goes to the exact kernel), cutoffs
    //     from far below to above the attainable range, plus the degenerate matrices of SURVEY.md Q4 (max_raw = 0, -inf, huge)
    for (int round = 0; round < 12; round++) {
        std::vector<double> values;
        std::vector<int32_t> widths;
        std::vector<double> cutoffs;
        const int n = round == 0 ? 0 : (round == 1 ? 1 : 40 + 37 * round);
        for (int p = 0; p < n; p++) {
            const int W = 1 + (int) (rng() % 70);
            widths.push_back(W);
            const int kind = (int) (rng() % 16);
            for (int b = 0; b < 4; b++)
                for (int c = 0; c < W; c++) {
                    double v = std::round(1e5 * std::log((0.002 + uni(rng)) / 0.25)) / 1e5;
                    if (kind == 0) v = -std::fabs(v) - 0.1;                         // max_raw = 0
                    if (kind == 1 && (b + c) % 5 == 0) v = -INFINITY;
                    if (kind == 2) v *= 1e6;
                    if (kind == 3) v *= 1e-8;
                    if (kind == 4) v = 0.0;
                    values.push_back(v);
                }
            const double cuts[] = {0.35 + 0.5 * uni(rng), -0.3, 0.0, 1.0, 1.0 + 1e-10, 2.0, 1e-300, NAN};
            cutoffs.push_back(kind >= 8 ? cuts[0] : cuts[rng() % 8]);
        }
        plan_all(values, widths, cutoffs);
    }
    // (2) the benchmark motif set at each of its cutoff columns, when the test hands it over
    if (argc > 1) {
        FILE *f = std::fopen(argv[1], "rb");
        CHECK(f);
        int32_t n = 0, n_sets = 0;
        CHECK(std::fread(&n, 4, 1, f) == 1 && std::fread(&n_sets, 4, 1, f) == 1 && n > 0 && n_sets > 0);
        std::vector<int32_t> widths((size_t) n);
        CHECK(std::fread(widths.data(), 4, (size_t) n, f) == (size_t) n);
        std::vector<double> cuts((size_t) n * (size_t) n_sets);
        CHECK(std::fread(cuts.data(), 8, cuts.size(), f) == cuts.size());
        size_t nv = 0;
        for (int32_t w : widths) nv += 4 * (size_t) w;
        std::vector<double> values(nv);
        CHECK(std::fread(values.data(), 8, nv, f) == nv);
        std::fclose(f);
        for (int s = 0; s < n_sets; s++) {
            std::vector<double> c((size_t) n);
            for (int32_t p = 0; p < n; p++) c[(size_t) p] = cuts[(size_t) p * (size_t) n_sets + (size_t) s];
            plan_all(values, widths, c);
        }
    }
    std::printf("plan_asan: ok (%ld plans)\n", n_plans);
    return 0;
}
