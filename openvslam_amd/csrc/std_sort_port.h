// std_sort_port.h -- the introsort algorithm of GNU libstdc++'s std::sort, written out so that it
// can run inside a kernel.
//
// match::angle_checker orders its 30 histogram bins with std::sort and a "larger bin first"
// comparator (src/openvslam/match/angle_checker.h:126-134).  std::sort is not stable, so which of
// several equally filled bins ends up among the first three depends on the algorithm.  To give the
// same answer as a GCC build of the reference on ties, the device code runs the same algorithm
// libstdc++ runs: introsort (median-of-three quicksort down to 16 elements, heapsort when the
// depth limit 2 * floor(log2(n)) is exhausted) followed by a final insertion sort.
// The host build of this header is checked against std::sort in tests (tests/test_sort_port.py
// via the ovs_test_sort_by_size entry point).
#ifndef OVS_STD_SORT_PORT_H
#define OVS_STD_SORT_PORT_H

#if defined(__HIPCC__)
#define OVS_HD __host__ __device__
#else
#define OVS_HD
#endif

namespace ovs_sort {

// Sorts idx[0..n) with comp(a, b) = size[a] > size[b]; n <= 64.
struct by_size_desc {
    const unsigned* size;
    OVS_HD bool operator()(unsigned a, unsigned b) const { return size[a] > size[b]; }
};

template <typename Comp>
OVS_HD inline void iter_swap(unsigned* a, unsigned* b) {
    const unsigned t = *a;
    *a = *b;
    *b = t;
}

template <typename Comp>
OVS_HD inline void move_median_to_first(unsigned* result, unsigned* a, unsigned* b, unsigned* c, Comp comp) {
    if (comp(*a, *b)) {
        if (comp(*b, *c)) {
            iter_swap<Comp>(result, b);
        }
        else if (comp(*a, *c)) {
            iter_swap<Comp>(result, c);
        }
        else {
            iter_swap<Comp>(result, a);
        }
    }
    else if (comp(*a, *c)) {
        iter_swap<Comp>(result, a);
    }
    else if (comp(*b, *c)) {
        iter_swap<Comp>(result, c);
    }
    else {
        iter_swap<Comp>(result, b);
    }
}

template <typename Comp>
OVS_HD inline unsigned* unguarded_partition(unsigned* first, unsigned* last, unsigned* pivot, Comp comp) {
    while (true) {
        while (comp(*first, *pivot)) {
            ++first;
        }
        --last;
        while (comp(*pivot, *last)) {
            --last;
        }
        if (!(first < last)) {
            return first;
        }
        iter_swap<Comp>(first, last);
        ++first;
    }
}

template <typename Comp>
OVS_HD inline unsigned* unguarded_partition_pivot(unsigned* first, unsigned* last, Comp comp) {
    unsigned* mid = first + (last - first) / 2;
    move_median_to_first(first, first + 1, mid, last - 1, comp);
    return unguarded_partition(first + 1, last, first, comp);
}

// ---- heap fallback (std::__partial_sort(first, last, last) = __heap_select + __sort_heap) ----
template <typename Comp>
OVS_HD inline void push_heap_(unsigned* first, long holeIndex, long topIndex, unsigned value, Comp comp) {
    long parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && comp(first[parent], value)) {
        first[holeIndex] = first[parent];
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    first[holeIndex] = value;
}

template <typename Comp>
OVS_HD inline void adjust_heap(unsigned* first, long holeIndex, long len, unsigned value, Comp comp) {
    const long topIndex = holeIndex;
    long secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (comp(first[secondChild], first[secondChild - 1])) {
            secondChild--;
        }
        first[holeIndex] = first[secondChild];
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        first[holeIndex] = first[secondChild - 1];
        holeIndex = secondChild - 1;
    }
    push_heap_(first, holeIndex, topIndex, value, comp);
}

template <typename Comp>
OVS_HD inline void make_heap_(unsigned* first, unsigned* last, Comp comp) {
    const long len = last - first;
    if (len < 2) {
        return;
    }
    long parent = (len - 2) / 2;
    while (true) {
        const unsigned value = first[parent];
        adjust_heap(first, parent, len, value, comp);
        if (parent == 0) {
            return;
        }
        parent--;
    }
}

template <typename Comp>
OVS_HD inline void heap_sort_all(unsigned* first, unsigned* last, Comp comp) {
    // __heap_select(first, last, last) reduces to make_heap; then __sort_heap
    make_heap_(first, last, comp);
    while (last - first > 1) {
        --last;
        const unsigned value = *last;
        *last = *first;
        adjust_heap(first, 0, last - first, value, comp);
    }
}

template <typename Comp>
OVS_HD inline void unguarded_linear_insert(unsigned* last, Comp comp) {
    const unsigned val = *last;
    unsigned* next = last;
    --next;
    while (comp(val, *next)) {
        *last = *next;
        last = next;
        --next;
    }
    *last = val;
}

template <typename Comp>
OVS_HD inline void insertion_sort(unsigned* first, unsigned* last, Comp comp) {
    if (first == last) {
        return;
    }
    for (unsigned* i = first + 1; i != last; ++i) {
        if (comp(*i, *first)) {
            const unsigned val = *i;
            for (unsigned* p = i; p != first; --p) {
                *p = *(p - 1);
            }
            *first = val;
        }
        else {
            unguarded_linear_insert(i, comp);
        }
    }
}

// One pending range of the introsort loop, as offsets from the start of the array.
struct sort_frame {
    int first, last, depth;
};
// The explicit stack holds at most one entry per level of the depth limit 2 * floor(log2(n)) plus
// one: 13 entries for n <= 64.
constexpr int kSortStackEntries = 16;

// `stack`: kSortStackEntries entries of working memory (a kernel passes LDS: a local array that is
// indexed dynamically would live in scratch memory).
template <typename Comp>
OVS_HD inline void std_sort(unsigned* first, unsigned* last, Comp comp, sort_frame* stack) {
    const long threshold = 16;
    if (first == last) {
        return;
    }
    // __introsort_loop, with the recursion on the right part turned into an explicit stack
    long n = last - first;
    int lg = 0;
    while ((n >> (lg + 1)) != 0) {
        ++lg;
    }
    int sp = 0;
    stack[sp++] = sort_frame{0, (int)n, 2 * lg};
    while (sp > 0) {
        const sort_frame f = stack[--sp];
        unsigned* f_first = first + f.first;
        unsigned* f_last = first + f.last;
        int depth = f.depth;
        while (f_last - f_first > threshold) {
            if (depth == 0) {
                heap_sort_all(f_first, f_last, comp);
                f_last = f_first;
                break;
            }
            --depth;
            unsigned* cut = unguarded_partition_pivot(f_first, f_last, comp);
            // the recursive call sorts [cut, last) first, then the loop continues on [first, cut);
            // partitions of disjoint ranges commute, so the order of processing does not matter
            stack[sp++] = sort_frame{(int)(cut - first), (int)(f_last - first), depth};
            f_last = cut;
        }
    }
    // __final_insertion_sort
    if (last - first > threshold) {
        insertion_sort(first, first + threshold, comp);
        for (unsigned* i = first + threshold; i != last; ++i) {
            unguarded_linear_insert(i, comp);
        }
    }
    else {
        insertion_sort(first, last, comp);
    }
}

// index_sort_by_size of angle_checker.h:126-134: idx = 0..n-1 sorted by size descending
OVS_HD inline void index_sort_by_size(const unsigned* size, unsigned* idx, int n, sort_frame* stack) {
    for (int i = 0; i < n; ++i) {
        idx[i] = (unsigned)i;
    }
    by_size_desc comp{size};
    std_sort(idx, idx + n, comp, stack);
}
// host convenience: the stack is a local array
inline void index_sort_by_size(const unsigned* size, unsigned* idx, int n) {
    sort_frame stack[kSortStackEntries];
    index_sort_by_size(size, idx, n, stack);
}

} // namespace ovs_sort

#endif
