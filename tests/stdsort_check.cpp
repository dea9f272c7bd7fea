// Compares kasa_amd/csrc/stdsort_order.h with std::sort itself (tests/test_host_cpu.py compiles and runs this).
// With an argument: prints 300 tie-heavy cases "rel rel ...|id id ..." (std::sort's order) for the Python restatement.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <tuple>
#include <vector>
#include "../kasa_amd/csrc/stdsort_order.h"

int main(int argc, char **)
{
    std::mt19937_64 rng(12345);
    long checked = 0, heap = 0;
    const int rounds = argc > 1 ? 300 : 60000;
    for (int round = 0; round < rounds; ++round) {
        const int n = 1 + (int)(rng() % (round % 7 == 0 ? 1500 : 300));
        const int distinct = 1 + (int)(rng() % (round % 3 == 0 ? 4 : (round % 3 == 1 ? 40 : 100000)));   // many ties, some ties, almost none
        std::vector<std::tuple<size_t, float, double>> res(n);          // what Compare::scoringFunc sorts: (taxon, score, relative score)
        std::vector<double> rel(n);
        for (int i = 0; i < n; ++i) rel[i] = (double)(rng() % distinct) * 0.125;
        if (round % 11 == 0) std::sort(rel.begin(), rel.end());
        if (round % 13 == 0) std::sort(rel.begin(), rel.end(), std::greater<double>());
        for (int i = 0; i < n; ++i) res[i] = std::make_tuple((size_t)i, (float)rel[i], rel[i]);
        std::sort(res.begin(), res.end(), [](const std::tuple<size_t, float, double> &a, const std::tuple<size_t, float, double> &b) { return std::get<2>(a) > std::get<2>(b); });
        if (argc > 1) {
            for (int i = 0; i < n; ++i) std::printf("%s%.3f", i ? " " : "", rel[i]);
            std::printf("|");
            for (int i = 0; i < n; ++i) std::printf("%s%zu", i ? " " : "", std::get<0>(res[i]));
            std::printf("\n");
            continue;
        }
        std::vector<uint16_t> ids(n);
        for (int i = 0; i < n; ++i) ids[i] = (uint16_t)i;
        const bool ok = stdsort_order(ids.data(), n, [&](uint16_t x, uint16_t y) { return rel[x] > rel[y]; });
        if (!ok) { ++heap; continue; }
        for (int i = 0; i < n; ++i)
            if (ids[i] != (uint16_t)std::get<0>(res[i])) { std::printf("MISMATCH round %d n %d at %d\n", round, n, i); return 1; }
        // ... and the prefix-only form: the first `need` positions, whatever is done with the rest
        const int need = 1 + (int)(rng() % (round % 2 ? 8 : 80));
        for (int i = 0; i < n; ++i) ids[i] = (uint16_t)i;
        int covered = 0;
        if (!stdsort_order(ids.data(), n, [&](uint16_t x, uint16_t y) { return rel[x] > rel[y]; }, need, &covered)) { std::printf("PREFIX gave up, round %d\n", round); return 1; }
        if (covered < std::min(need, n)) { std::printf("PREFIX covers %d of %d, round %d\n", covered, need, round); return 1; }
        for (int i = 0; i < covered; ++i)
            if (ids[i] != (uint16_t)std::get<0>(res[i])) { std::printf("PREFIX MISMATCH round %d n %d need %d at %d\n", round, n, need, i); return 1; }
        ++checked;
    }
    if (argc > 1) return 0;
    std::printf("OK %ld arrays identical to std::sort, %ld left to the heap sort\n", checked, heap);
    return checked > 50000 ? 0 : 2;
}
