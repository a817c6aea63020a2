// tests/sanitize/host_asan.cpp -- round 6's host-side code under AddressSanitizer + UndefinedBehaviorSanitizer, compiled by plain g++ from the
// SAME source files the library is built from: the host packer (motifscan_amd/csrc/ms_hostpack.cpp: convert_seq + region hints, AVX2 and scalar
// paths, ragged tails) against a byte-at-a-time restatement of cscore.c:81-114, and the NUMA look-ups (ms_numa.cpp: cpulist parser, sysfs
// reads) on well-formed, malformed and missing inputs.  CPU build container only (`make -C motifscan_amd/csrc sanitize`).
#include <sched.h>
#include <sys/stat.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace ms {
void host_pack_units(const uint8_t *bases, int64_t n_bases, int64_t u0, int64_t u1, uint32_t *codes, uint32_t *nmask);
void host_region_hints(const int64_t *offsets, int64_t R, int64_t b0, int64_t b1, int32_t *blk2reg, int32_t *info, bool all_far);
int parse_cpulist(const char *text, cpu_set_t *set);
int numa_node_of_bdf(const char *bdf, const char *root);
int numa_cpus_of_node(int node, const char *root, cpu_set_t *set);
int numa_node_count(const char *root);
}  // namespace ms

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

static int code_of(uint8_t ch) {                     // cscore.c:92-111
    switch (ch) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return -1; }
}

static void write_file(const std::string &path, const char *text) {
    FILE *f = fopen(path.c_str(), "w");
    CHECK(f != nullptr);
    fputs(text, f);
    fclose(f);
}

int main() {
    std::mt19937_64 rng(7);
    const char alphabet[] = "ACGTacgtNnRYKMSW-*.@`{\x01\xff";
    int n_cases = 0;
    for (int64_t n : {0LL, 1LL, 31LL, 32LL, 33LL, 63LL, 64LL, 65LL, 1000LL, 4097LL, 70001LL}) {
        // exact-size buffers: a read or write past either end is the sanitizer's to find
        std::vector<uint8_t> bases((size_t) n);
        for (auto &b : bases) b = (uint8_t) alphabet[rng() % (sizeof(alphabet) - 1)];
        const int64_t units = (n + 31) / 32;
        std::vector<uint32_t> codes((size_t) (2 * units)), nmask((size_t) units);
        const int64_t cut = units / 3;                // two calls, as two threads would split the units
        ms::host_pack_units(bases.data(), n, 0, cut, codes.data(), nmask.data());
        ms::host_pack_units(bases.data(), n, cut, units, codes.data(), nmask.data());
        for (int64_t i = 0; i < 32 * units; i++) {
            const uint64_t cw = (uint64_t) codes[(size_t) (2 * (i / 32))] | ((uint64_t) codes[(size_t) (2 * (i / 32) + 1)] << 32);
            const int code = (int) ((cw >> (2 * (i % 32))) & 3), isn = (int) ((nmask[(size_t) (i / 32)] >> (i % 32)) & 1);
            if (i < n) { const int want = code_of(bases[(size_t) i]); CHECK(want < 0 ? (isn == 1 && code == 0) : (isn == 0 && code == want)); }
            else CHECK(code == 0 && isn == 0);
        }
        // region hints over ragged regions that tile [0, n)
        std::vector<int64_t> off{0};
        while (off.back() < n) off.push_back(std::min<int64_t>(n, off.back() + (int64_t) (rng() % 300)));
        const int64_t R = (int64_t) off.size() - 1, blocks = (n + 63) / 64 + 1;
        if (R > 0) {
            std::vector<int32_t> blk((size_t) blocks), info((size_t) (4 * blocks));
            ms::host_region_hints(off.data(), R, 0, blocks / 2, blk.data(), info.data(), false);
            ms::host_region_hints(off.data(), R, blocks / 2, blocks, blk.data(), info.data(), false);
            for (int64_t b = 0; b < blocks; b++) {
                int64_t r = 0;
                while (r + 1 < R && off[(size_t) (r + 1)] <= 64 * b) r++;
                CHECK(blk[(size_t) b] == (int32_t) r && info[(size_t) (4 * b)] == (int32_t) r && info[(size_t) (4 * b + 1)] == (int32_t) (off[(size_t) r] - 64 * b));
            }
        }
        n_cases++;
    }
    // ---- NUMA look-ups
    cpu_set_t set;
    CHECK(ms::parse_cpulist("0-3,8,10-11\n", &set) == 7 && CPU_ISSET(8, &set) && !CPU_ISSET(9, &set));
    CHECK(ms::parse_cpulist("5", &set) == 1 && ms::parse_cpulist("", &set) == 0);
    CHECK(ms::parse_cpulist("3-1", &set) == -1 && ms::parse_cpulist("a-b", &set) == -1 && ms::parse_cpulist("1,,2", &set) == -1 && ms::parse_cpulist("1-", &set) == -1);
    CHECK(ms::parse_cpulist("0-100000", &set) > 0);                      // beyond CPU_SETSIZE: clipped, not written
    char tmpl[] = "/tmp/ms_numa_XXXXXX";
    const char *root = mkdtemp(tmpl);
    CHECK(root != nullptr);
    const std::string r(root);
    for (const char *d : {"/sys", "/sys/bus", "/sys/bus/pci", "/sys/bus/pci/devices", "/sys/bus/pci/devices/0000:c1:00.0", "/sys/devices", "/sys/devices/system",
                          "/sys/devices/system/node", "/sys/devices/system/node/node1"})
        CHECK(mkdir((r + d).c_str(), 0700) == 0);
    write_file(r + "/sys/bus/pci/devices/0000:c1:00.0/numa_node", "1\n");
    write_file(r + "/sys/devices/system/node/node1/cpulist", "64-127,192-255\n");
    write_file(r + "/sys/devices/system/node/online", "0-1\n");
    CHECK(ms::numa_node_of_bdf("0000:C1:00.0", root) == 1 && ms::numa_node_of_bdf("0000:00:00.0", root) == -1);
    CHECK(ms::numa_cpus_of_node(1, root, &set) == 128 && ms::numa_cpus_of_node(0, root, &set) == 0 && ms::numa_cpus_of_node(-1, root, &set) == 0);
    CHECK(ms::numa_node_count(root) == 2 && ms::numa_node_count("/nonexistent") == 1);
    std::printf("host_asan: ok (%d packer cases)\n", n_cases);
    return 0;
}
