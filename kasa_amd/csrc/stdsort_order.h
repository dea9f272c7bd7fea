// stdsort_order.h -- where libstdc++'s std::sort leaves elements that compare equal.
//
// The reference ranks the hits of a read with std::sort over their relative scores (source/modes/Compare.hpp:1525-1527).
// std::sort is not stable, but it is deterministic: which of two taxa with the same relative score is printed first is
// decided by the library's algorithm (introsort: median-of-three pivot moved to the front, unguarded Hoare partition,
// ranges of at most 16 left to one final insertion sort).  To print what the reference prints, the device ranking has to
// walk the same path.  This header restates that algorithm over an array of element ids -- every comparison and every
// move of libstdc++'s __introsort_loop / __final_insertion_sort (bits/stl_algo.h, unchanged from GCC 4.9 to 13) happens
// here in the same order, so `a` ends up as the ids in std::sort's output order.  The heap sort libstdc++ falls back to
// after 2 log2(n) partitioning levels is not restated: the function then returns false and the caller takes another way
// (it needs an adversarial input).  tests/test_host_cpu.py runs it against std::sort itself on tie-heavy inputs;
// kasa_amd/report.py:_stdsort_order is the same in Python for the Python host.
#pragma once
#include <stdint.h>

#if !defined(__HIPCC__) && !defined(__host__)
#define __host__
#define __device__
#endif

// less(x, y): the element with id x goes before the one with id y.  a[0 .. n) = the ids in input order, n < 65536.
// need: only the positions [0, need) have to come out as std::sort leaves them (the device wants the few hits a writer
// prints).  The partitioning leaves everything right of a cut not smaller than everything left of it, and neither the
// sorting of a right-hand range nor the final insertion of its elements moves anything across the cut; so the ranges
// are finished from the left and the rest is dropped once `need` positions are covered.  *covered (may be NULL) receives
// how many leading positions are final (>= min(need, n)).
// The ranges still to be partitioned wait on a stack; where it lives is the caller's choice (the device keeps it in LDS).
struct StdsortLocalStack {
    static constexpr int CAP = 34;
    int32_t f[CAP], l[CAP], d[CAP];
    __host__ __device__ inline void put(int at, int first, int last, int depth) { f[at] = first; l[at] = last; d[at] = depth; }
    __host__ __device__ inline void get(int at, int &first, int &last, int &depth) const { first = f[at]; last = l[at]; depth = d[at]; }
};

template <class Ids, class Less, class Stack = StdsortLocalStack>
__host__ __device__ inline bool stdsort_order(Ids a, int n, Less less, int need = 0x7fffffff, int *covered = nullptr, Stack st = Stack())
{
    if (covered) *covered = n;
    if (n <= 1) return true;
    auto swap_at = [&](int x, int y) { const uint16_t t = a[x]; a[x] = a[y]; a[y] = t; };
    int lg = 0;
    while ((n >> (lg + 1)) != 0) ++lg;
    // ranges still to be partitioned (the recursion of __introsort_loop on the right-hand parts)
    int top = 0;
    st.put(0, 0, n, 2 * lg);
    ++top;
    while (top > 0) {
        --top;
        int first, last, depth;
        st.get(top, first, last, depth);
        while (last - first > 16) {
            if (depth == 0) return false;                              // libstdc++ heap-sorts this range
            --depth;
            const int mid = first + (last - first) / 2, A = first + 1, B = mid, C = last - 1;
            // the median of a[A], a[B], a[C] goes to the front
            if (less(a[A], a[B])) {
                if (less(a[B], a[C])) swap_at(first, B);
                else if (less(a[A], a[C])) swap_at(first, C);
                else swap_at(first, A);
            } else if (less(a[A], a[C])) swap_at(first, A);
            else if (less(a[B], a[C])) swap_at(first, C);
            else swap_at(first, B);
            // partition of (first, last) around the front element; no bounds checks: the median guards both ends
            int lo = first + 1, hi = last;
            for (;;) {
                while (less(a[lo], a[first])) ++lo;
                --hi;
                while (less(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                swap_at(lo, hi);
                ++lo;
            }
            if (top >= Stack::CAP) return false;
            st.put(top, lo, last, depth);
            ++top;
            last = lo;
        }
        if (last >= need && last < n) { n = last; break; }             // [0, last) is partitioned into small ranges: enough
    }
    if (covered) *covered = n;
    // one insertion sort over everything: guarded for the first 16, unguarded after (something not larger is to the left)
    auto insert_unguarded = [&](int i) {
        const uint16_t v = a[i];
        int j = i - 1;
        while (less(v, a[j])) { a[j + 1] = a[j]; --j; }
        a[j + 1] = v;
    };
    const int guarded = n > 16 ? 16 : n;
    for (int i = 1; i < guarded; ++i) {
        if (less(a[i], a[0])) {
            const uint16_t v = a[i];
            for (int j = i; j > 0; --j) a[j] = a[j - 1];
            a[0] = v;
        } else insert_unguarded(i);
    }
    for (int i = guarded; i < n; ++i) insert_unguarded(i);
    return true;
}
