// ovs_match.hip -- popcount Hamming kernels that are not window searches: pairwise / all-pairs
// distances (match::compute_descriptor_distance_32, src/openvslam/match/base.h:18-39) and the
// brute-force top-2 search over a resident descriptor database.
//
// None of this is a dense contraction, so there is no MFMA here: distances are v_xor + v_bcnt over
// 8 dwords, selections are per-lane running minima merged by the lexicographic (distance, row) key.
#include "ovs_match_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

using namespace ovsm;

namespace {

// ---- plain Hamming kernels ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hamming_pairs(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int n, u32* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        u32 da[8], db[8];
        load_desc(a + 32 * (size_t)i, da);
        load_desc(b + 32 * (size_t)i, db);
        out[i] = hamming8(da, db);
    }
}

// out[i] = hamming(a[ia[i]], b[ib[i]])
__global__ __launch_bounds__(256) void k_hamming_indexed(const uint8_t* __restrict__ a, const u32* __restrict__ ia,
                                                         const uint8_t* __restrict__ b, const u32* __restrict__ ib, long long n,
                                                         unsigned short* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        u32 da[8], db[8];
        load_desc(a + 32 * (size_t)ia[i], da);
        load_desc(b + 32 * (size_t)ib[i], db);
        out[i] = (unsigned short)hamming8(da, db);
    }
}

// out[i][j] for a 64 x 64 tile per workgroup: b descriptors staged in LDS
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t* __restrict__ a, int na, const uint8_t* __restrict__ b, int nb,
                                                        unsigned short* __restrict__ out) {
    __shared__ u32 sb[64 * 9];
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    for (int t = tid; t < 64 * 8; t += 256) {
        const int r = t >> 3, wi = t & 7;
        sb[r * 9 + wi] = (j0 + r < nb) ? reinterpret_cast<const u32*>(b)[(size_t)(j0 + r) * 8 + wi] : 0u;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < 64; r += 4) {
        const int i = i0 + r;
        if (i >= na) {
            break;
        }
        u32 da[8];
        load_desc(a + 32 * (size_t)i, da);
        const int j = j0 + lane;
        if (j < nb) {
            out[(size_t)i * nb + j] = (unsigned short)hamming8(da, &sb[lane * 9]);
        }
    }
}

// ---- database top-2 search -------------------------------------------------------------------------
// Queries are staged in LDS (64 per workgroup, one per lane of every wave); each wave streams a
// slice of the database rows, every lane keeps the running best / second best of its query under
// the serial update rule (strict '<'), waves and workgroups (database slices) are merged by the
// lexicographic (distance, row) minimum, which equals the serial scan over all rows.
constexpr int kKnnRowsPerBlock = 4096;

// (selects, not branches: the compiler turns the branchy form into one store through a computed
// address, which puts b0 / b1 into scratch memory)
__device__ __forceinline__ void knn_push(u64 key, u64& b0, u64& b1) {
    const bool lt0 = key < b0, lt1 = key < b1;
    b1 = lt0 ? b0 : (lt1 ? key : b1);
    b0 = lt0 ? key : b0;
}

__global__ __launch_bounds__(256) void k_knn2_partial(const uint8_t* __restrict__ db, long long n_rows, const uint8_t* __restrict__ queries, int nq,
                                                      u64* __restrict__ partial /* [slices][nq][2] */) {
    __shared__ u32 rows[256 * 8];
    __shared__ u64 wb[4][64][2];
    const int q0 = blockIdx.x * 64;
    const long long r0 = (long long)blockIdx.y * kKnnRowsPerBlock;
    const long long r1 = min(r0 + (long long)kKnnRowsPerBlock, n_rows);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 qd[8];
    const int qi = q0 + lane;
    if (qi < nq) {
        load_desc(queries + 32 * (size_t)qi, qd);
    }
    else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            qd[i] = 0;
        }
    }
    // best and second best of this lane's slice of rows: rows arrive in increasing order, so strict
    // `<` on the distance alone keeps the lowest row among equals (the 64-bit keys are only built
    // at the end); updates become rare after the first rows, so the wave skips them with one ballot
    u32 d0 = 0xFFFFFFFFu, d1 = 0xFFFFFFFFu, i0 = 0, i1 = 0;
    for (long long base = r0; base < r1; base += 256) {
        __syncthreads();
        {
            // 256 rows x 32 B, coalesced 16-B loads: thread t loads halves t and t + 256
            const long long row_a = base + (tid >> 1);
            const uint4 z = make_uint4(0, 0, 0, 0);
            uint4 va = z, vb = z;
            if (row_a < r1) {
                va = reinterpret_cast<const uint4*>(db)[row_a * 2 + (tid & 1)];
            }
            const long long row_b = base + 128 + (tid >> 1);
            if (row_b < r1) {
                vb = reinterpret_cast<const uint4*>(db)[row_b * 2 + (tid & 1)];
            }
            reinterpret_cast<uint4*>(rows)[tid] = va;
            reinterpret_cast<uint4*>(rows)[256 + tid] = vb;
        }
        __syncthreads();
        const int lim = (int)min((long long)256, r1 - base);
        // wave w scans rows w*64 .. w*64+63 of the chunk; all lanes read the same row (broadcast)
        const int jb = wave * 64, je = min(jb + 64, lim);
        for (int j = jb; j < je; ++j) {
            const u32* r = &rows[j * 8];
            u32 d = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                d += __popc(qd[i] ^ r[i]);
            }
            if (__ballot(d < d1) != 0) {
                const u32 row = (u32)(base - r0) + (u32)j;
                if (d < d0) {
                    d1 = d0;
                    i1 = i0;
                    d0 = d;
                    i0 = row;
                }
                else if (d < d1) {
                    d1 = d;
                    i1 = row;
                }
            }
        }
    }
    wb[wave][lane][0] = d0 == 0xFFFFFFFFu ? kNoKey : (((u64)d0 << 48) | (u64)(r0 + i0));
    wb[wave][lane][1] = d1 == 0xFFFFFFFFu ? kNoKey : (((u64)d1 << 48) | (u64)(r0 + i1));
    __syncthreads();
    if (wave == 0 && qi < nq) {
        u64 m0 = kNoKey, m1 = kNoKey;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            knn_push(wb[w][lane][0], m0, m1);
            knn_push(wb[w][lane][1], m0, m1);
        }
        u64* o = partial + ((size_t)blockIdx.y * nq + qi) * 2;
        o[0] = m0;
        o[1] = m1;
    }
}

// ---- the same search on the matrix cores ------------------------------------------------------------
// With query bits as +1 / -1 bytes and database bits as 1 / 0 bytes the integer dot product over the
// 256 bit positions is D = |q and r| - |not q and r|, and the Hamming distance is |q| - D, so a tile
// of 16 rows x 16 queries is four v_mfma_i32_16x16x64_i8.  A workgroup of four waves shares the
// expanded rows through LDS (double buffered, stored in fragment order: one ds_read_b128 per lane and
// fragment, conflict free); every wave keeps its own 128 queries as B fragments in registers.  The
// accumulator layout puts a query on the lane (column = lane & 15) and four consecutive rows in the
// four registers, so the serial best / second-best rule runs per lane on D (larger D = smaller
// distance; strict '>' keeps the lowest row among equals); a wave skips a tile with one compare of
// the tile maximum when no lane improves.  The four lane groups of a query, the row slices and the
// shards are merged with the (distance, row) keys as before.
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kMmRows = 64;          // database rows per iteration (four 16-row tiles)
constexpr int kMmChunks = kMmRows * 16 / 256; // 16-bit chunks each thread expands per iteration
#ifndef OVS_MM_TILES
#define OVS_MM_TILES 4
#define OVS_MM_OCC 3
#endif
constexpr int kMmTiles = OVS_MM_TILES; // 16-query tiles per wave
constexpr int kMmBlocksPerCu = OVS_MM_OCC;
constexpr int kMmQueriesPerWave = 16 * kMmTiles, kMmQueriesPerBlock = 4 * kMmQueriesPerWave;
constexpr int kMmMaxSliceRows = 1 << 30;
constexpr int kMmNoD = -(1 << 20);

// 16 bits -> 16 bytes of 0 / 1 (bit 4 j + e -> byte e of dword j)
__device__ __forceinline__ v4i expand16(u32 bits16) {
    v4i v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = (int)(__umul24((bits16 >> (4 * j)) & 15u, 0x00204081u) & 0x01010101u);
    }
    return v;
}

__global__ __launch_bounds__(256, kMmBlocksPerCu) void k_knn2_mfma(const uint8_t* __restrict__ db, long long n_rows, int rows_per_slice,
                                                      const uint8_t* __restrict__ queries, int nq, u64* __restrict__ partial /* [slices][nq][2] */) {
    __shared__ v4i atile[2][16 * kMmRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int q0 = blockIdx.x * kMmQueriesPerBlock + wave * kMmQueriesPerWave;
    const long long r0 = (long long)blockIdx.y * rows_per_slice;
    const long long r1 = min(r0 + (long long)rows_per_slice, n_rows);
    const int nblk = (int)((r1 - r0 + kMmRows - 1) / kMmRows);

    // B fragments: query q0 + 16 t + n, k step s: bits [64 s + 16 kq, + 16) as +1 (bit set) / -1 bytes
    v4i bq[kMmTiles][4];
    int pq[kMmTiles];
#pragma unroll
    for (int t = 0; t < kMmTiles; ++t) {
        const int qi = q0 + 16 * t + n;
        u32 qd[8];
        if (qi < nq) {
            load_desc(queries + 32 * (size_t)qi, qd);
        }
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                qd[i] = 0;
            }
        }
        pq[t] = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            pq[t] += __popc(qd[i]);
        }
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            // halfword kq of the 64 bits of this k step (a shift: indexing qd by kq would put it in scratch memory)
            const u64 bits64 = ((u64)qd[2 * sidx + 1] << 32) | qd[2 * sidx];
            const v4i e = expand16((u32)(bits64 >> (16 * kq)) & 0xFFFFu);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bq[t][sidx][j] = qi < nq ? (int)~((u32)e[j] * 0xFEu) : 0;
            }
        }
    }
    // per lane: best / second best (D and row offset inside the slice) among the rows this lane group
    // has seen, and thr = the query's second-best D over all four lane groups as of its last update.
    // A row can only enter the query's final top two if its D exceeds thr (rows arrive in increasing
    // order, so an equal D from a later row loses), which makes thr an exact and much tighter gate
    // than the lane's own second best.
    int D0[kMmTiles], D1[kMmTiles], thr[kMmTiles];
    u32 I0[kMmTiles], I1[kMmTiles];
#pragma unroll
    for (int t = 0; t < kMmTiles; ++t) {
        D0[t] = D1[t] = thr[t] = kMmNoD;
        I0[t] = I1[t] = 0;
    }

    // expansion: a block of kMmRows rows is 16 * kMmRows chunks of 16 bits, one per LDS slot.  Wave w
    // fills the fragments of k step w of every 16-row tile, lane l the slot it will later be read
    // from by lane l (row l & 15, halfword 4 w + (l >> 4)): a wave writes 1 KB of consecutive LDS.
    // The bits are fetched two iterations ahead of their use.
    const uint16_t* db16 = reinterpret_cast<const uint16_t*>(db);
    const u32 slice_rows = (u32)(r1 - r0);
    static_assert(kMmChunks == kMmRows / 16, "one chunk per thread and 16-row tile");
    const int exp_slot = (wave << 6) + lane, exp_hw = 4 * wave + kq;
    auto load_bits = [&](int blk, u32* out) {
#pragma unroll
        for (int c = 0; c < kMmChunks; ++c) {
            const u32 row = (u32)blk * kMmRows + 16u * c + (u32)n;
            out[c] = (blk < nblk && row < slice_rows) ? (u32)db16[(r0 + row) * 16 + exp_hw] : 0u;
        }
    };
    int slot[kMmChunks];
#pragma unroll
    for (int c = 0; c < kMmChunks; ++c) {
        slot[c] = (c << 8) + exp_slot;
    }
    u32 bits0[kMmChunks], bits1[kMmChunks];
    load_bits(0, bits0);
#pragma unroll
    for (int c = 0; c < kMmChunks; ++c) {
        atile[0][slot[c]] = expand16(bits0[c]);
    }
    load_bits(1, bits0);
    load_bits(2, bits1);
    __syncthreads();
#pragma unroll 1
    for (int b = 0; b < nblk; ++b) {
        const int cur = b & 1;
        if (b + 1 < nblk) {
#pragma unroll
            for (int c = 0; c < kMmChunks; ++c) {
                atile[cur ^ 1][slot[c]] = expand16(bits0[c]);
                bits0[c] = bits1[c];
            }
            load_bits(b + 3, bits1);
        }
        const bool full = (u32)(b + 1) * kMmRows <= slice_rows;
        u32 dirty = 0; // tiles whose best / second best changed in this iteration (wave uniform)
#pragma unroll
        for (int mt = 0; mt < kMmRows / 16; ++mt) {
            v4i af[4];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                af[sidx] = atile[cur][((mt * 4 + sidx) << 6) + lane];
            }
            // all tiles' products first, then the selection: the matrix pipe works on the later
            // tiles while the vector unit handles the earlier ones
            v4i acc[kMmTiles];
#pragma unroll
            for (int t = 0; t < kMmTiles; ++t) {
                acc[t] = v4i{0, 0, 0, 0};
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) {
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[sidx], bq[t][sidx], acc[t], 0, 0, 0);
                }
            }
            // accumulator register i of lane group kq is row 4 kq + i of the 16-row tile
            const u32 row_b = (u32)b * kMmRows + 16u * mt + 4u * kq;
#pragma unroll
            for (int t = 0; t < kMmTiles; ++t) {
                const int m = max(max(acc[t][0], acc[t][1]), max(acc[t][2], acc[t][3]));
                if (__ballot(m > thr[t]) != 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u32 row = row_b + i;
                        int D = acc[t][i];
                        if (!full) {
                            D = row < slice_rows ? D : kMmNoD;
                        }
                        const bool gt0 = D > D0[t], gt1 = D > D1[t];
                        D1[t] = gt0 ? D0[t] : (gt1 ? D : D1[t]);
                        I1[t] = gt0 ? I0[t] : (gt1 ? row : I1[t]);
                        D0[t] = gt0 ? D : D0[t];
                        I0[t] = gt0 ? row : I0[t];
                    }
                    dirty |= 1u << t;
                }
            }
        }
        // refresh the gate of the tiles that changed: second best of the query over its four lane groups
#pragma unroll
        for (int t = 0; t < kMmTiles; ++t) {
            if (dirty & (1u << t)) {
                int a0 = D0[t], a1 = D1[t];
#pragma unroll
                for (int d = 16; d <= 32; d <<= 1) {
                    const int b0 = __shfl_xor(a0, d), b1 = __shfl_xor(a1, d);
                    const int n1 = max(min(a0, b0), max(a1, b1));
                    a0 = max(a0, b0);
                    a1 = n1;
                }
                thr[t] = a1;
            }
        }
        __syncthreads();
    }
    // the four lane groups of a query hold disjoint rows: merge and write the slice's result
#pragma unroll
    for (int t = 0; t < kMmTiles; ++t) {
        u64 k0 = D0[t] == kMmNoD ? kNoKey : (((u64)(u32)(pq[t] - D0[t]) << 48) | (u64)(r0 + I0[t]));
        u64 k1 = D1[t] == kMmNoD ? kNoKey : (((u64)(u32)(pq[t] - D1[t]) << 48) | (u64)(r0 + I1[t]));
#pragma unroll
        for (int d = 16; d <= 32; d <<= 1) {
            const u64 o0 = ((u64)__shfl_xor((u32)(k0 >> 32), d) << 32) | __shfl_xor((u32)k0, d);
            const u64 o1 = ((u64)__shfl_xor((u32)(k1 >> 32), d) << 32) | __shfl_xor((u32)k1, d);
            knn_push(o0, k0, k1);
            knn_push(o1, k0, k1);
        }
        const int qi = q0 + 16 * t + n;
        if (kq == 0 && qi < nq) {
            u64* o = partial + ((size_t)blockIdx.y * nq + qi) * 2;
            o[0] = k0;
            o[1] = k1;
        }
    }
}

__global__ __launch_bounds__(256) void k_knn2_finish(const u64* __restrict__ partial, int slices, int nq, long long row_offset,
                                                     ovs_knn2* __restrict__ out) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) {
        return;
    }
    u64 m0 = kNoKey, m1 = kNoKey;
    for (int s = 0; s < slices; ++s) {
        const u64* p = partial + ((size_t)s * nq + qi) * 2;
        knn_push(p[0], m0, m1);
        knn_push(p[1], m0, m1);
    }
    // (field by field: a local struct copied as a whole goes through scratch memory)
    ovs_knn2* o = out + qi;
    o->dist[0] = m0 == kNoKey ? kMaxDist : (u32)(m0 >> 48);
    o->dist[1] = m1 == kNoKey ? kMaxDist : (u32)(m1 >> 48);
    o->row[0] = m0 == kNoKey ? -1 : (long long)(m0 & 0xFFFFFFFFFFFFull) + row_offset;
    o->row[1] = m1 == kNoKey ? -1 : (long long)(m1 & 0xFFFFFFFFFFFFull) + row_offset;
}

__global__ __launch_bounds__(256) void k_knn2_merge(const ovs_knn2* __restrict__ lists, int n_lists, int nq, ovs_knn2* __restrict__ out) {
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) {
        return;
    }
    u64 m0 = kNoKey, m1 = kNoKey;
    for (int s = 0; s < n_lists; ++s) {
        const ovs_knn2 e = lists[(size_t)s * nq + qi];
        for (int k = 0; k < 2; ++k) {
            if (0 <= e.row[k]) {
                knn_push(((u64)e.dist[k] << 48) | (u64)e.row[k], m0, m1);
            }
        }
    }
    ovs_knn2* o = out + qi;
    o->dist[0] = m0 == kNoKey ? kMaxDist : (u32)(m0 >> 48);
    o->dist[1] = m1 == kNoKey ? kMaxDist : (u32)(m1 >> 48);
    o->row[0] = m0 == kNoKey ? -1 : (long long)(m0 & 0xFFFFFFFFFFFFull);
    o->row[1] = m1 == kNoKey ? -1 : (long long)(m1 & 0xFFFFFFFFFFFFull);
}


// landmark::compute_descriptor: one wavefront per landmark.  For every row r of the landmark's
// distance matrix the lanes split the columns, build the histogram of the 257 possible distances
// in LDS and find the bin holding element k = floor(0.5 * (n - 1)) of the sorted row; lane 0 keeps
// the first row with the smallest median.
__global__ __launch_bounds__(64) void k_landmark_median(const uint8_t* __restrict__ descs, const u32* __restrict__ offsets, int n_landmarks,
                                                        int* __restrict__ best_idx) {
    __shared__ u32 hist[264];
    const int l = blockIdx.x, lane = threadIdx.x;
    const u32 first = offsets[l], n = offsets[l + 1] - first;
    if (n == 0) {
        if (lane == 0) {
            best_idx[l] = -1;
        }
        return;
    }
    const uint8_t* base = descs + 32 * (size_t)first;
    const u32 k = (u32)(0.5 * (double)(n - 1));
    u32 best_median = kMaxDist, best = 0;
    for (u32 r = 0; r < n; ++r) {
        for (int b = lane; b < 264; b += 64) {
            hist[b] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u32 dr[8];
        load_desc(base + 32 * (size_t)r, dr);
        for (u32 j = lane; j < n; j += 64) {
            u32 dj[8];
            load_desc(base + 32 * (size_t)j, dj);
            atomicAdd(&hist[hamming8(dr, dj)], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lane b owns bins 5b .. 5b+4 (320 >= 257): inclusive prefix over lanes, then inside the lane
        u32 own[5], sum = 0;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int bin = 5 * lane + t;
            own[t] = bin < 264 ? hist[bin] : 0u;
            sum += own[t];
        }
        const u32 inc = wave_incl_scan_u32(sum);
        u32 run = inc - sum; // elements in lower bins
        u32 median = 0xFFFFFFFFu;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            if (median == 0xFFFFFFFFu && run <= k && k < run + own[t]) {
                median = (u32)(5 * lane + t);
            }
            run += own[t];
        }
        // exactly one lane found it
        const u64 who = __ballot(median != 0xFFFFFFFFu);
        median = __shfl(median, __ffsll((long long)who) - 1);
        if (median < best_median) {
            best_median = median;
            best = r;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        best_idx[l] = (int)best;
    }
}


} // namespace

// ---- scratch memory of the entry point that allocates nothing -----------------------------------------
// ovs_desc_db_search_noalloc_dev keeps the per-slice partial results of a search in this array instead
// of a buffer from hipMalloc.  Every GPU the library runs on holds its own copy of the array (at its own
// address) until the process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress: the address kernels are given is looked up once per device, with that device
// current, and kept in g_scratch_ptr.
constexpr size_t kNoAllocScratchBytes = 4u << 20;
static __device__ __align__(256) unsigned char scratch[kNoAllocScratchBytes];

namespace {
constexpr int kMaxScratchDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxScratchDevices] = {nullptr};

// address of this device's copy of `scratch`; `device` must be the current device
int scratch_for_device(int device, void** out) {
    if (device < 0 || device >= kMaxScratchDevices) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "device ordinal beyond the scratch table");
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        OVS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) {
            return ovs::fail(OVS_ERR_HIP, "hipGetSymbolAddress returned a null address");
        }
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return OVS_OK;
}
} // namespace

struct ovs_desc_db {
    int device = 0;
    ovs::dev_buffer own;
    const uint8_t* d_rows = nullptr;
    long long n_rows = 0;
    long long row_offset = 0;
    ovs::dev_buffer partial;
    ovs::dev_buffer st_q, st_out;
};

extern "C" {

int ovs_hamming_pairs(const uint8_t* a, const uint8_t* b, int n, uint32_t* out, int device) {
    if (n < 0 || (n > 0 && (!a || !b || !out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) return rc;
    const size_t nn = n;
    if ((rc = ctx->arena.reserve(2 * al(nn * 32) + al(nn * 4))) != OVS_OK) return rc;
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* da = cv.take<uint8_t>(nn * 32);
    auto* db = cv.take<uint8_t>(nn * 32);
    auto* dout = cv.take<u32>(nn);
    OVS_HIP(hipMemcpyAsync(da, a, nn * 32, hipMemcpyHostToDevice, ctx->stream));
    OVS_HIP(hipMemcpyAsync(db, b, nn * 32, hipMemcpyHostToDevice, ctx->stream));
    OVS_LAUNCH("k_hamming_pairs", k_hamming_pairs, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, da, db, n, dout);
    OVS_HIP(hipMemcpyAsync(out, dout, nn * 4, hipMemcpyDeviceToHost, ctx->stream));
    OVS_HIP(hipStreamSynchronize(ctx->stream));
    return OVS_OK;
}

int ovs_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out, int device) {
    if (na < 0 || nb < 0 || ((na > 0 && nb > 0) && (!a || !b || !out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    if (na == 0 || nb == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) return rc;
    const size_t sa = na, sb = nb;
    if ((rc = ctx->arena.reserve(al(sa * 32) + al(sb * 32) + al(sa * sb * 2))) != OVS_OK) return rc;
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* da = cv.take<uint8_t>(sa * 32);
    auto* db = cv.take<uint8_t>(sb * 32);
    auto* dout = cv.take<unsigned short>(sa * sb);
    OVS_HIP(hipMemcpyAsync(da, a, sa * 32, hipMemcpyHostToDevice, ctx->stream));
    OVS_HIP(hipMemcpyAsync(db, b, sb * 32, hipMemcpyHostToDevice, ctx->stream));
    OVS_LAUNCH("k_hamming_matrix", k_hamming_matrix, dim3((nb + 63) / 64, (na + 63) / 64), dim3(256), 0, ctx->stream, da, na, db, nb, dout);
    OVS_HIP(hipMemcpyAsync(out, dout, sa * sb * 2, hipMemcpyDeviceToHost, ctx->stream));
    OVS_HIP(hipStreamSynchronize(ctx->stream));
    return OVS_OK;
}

// ---- descriptor database ---------------------------------------------------------------------------
int ovs_desc_db_create(int device, ovs_desc_db** out) {
    if (!out) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    {
        // set-up of the entry point that allocates nothing: this device's address of `scratch`
        ovs::device_guard guard(device);
        void* p = nullptr;
        if ((rc = scratch_for_device(device, &p)) != OVS_OK) {
            return rc;
        }
    }
    auto* db = new ovs_desc_db();
    db->device = device;
    *out = db;
    return OVS_OK;
}

void ovs_desc_db_destroy(ovs_desc_db* db) {
    if (!db) {
        return;
    }
    ovs::device_guard guard(db->device);
    db->own.release();
    db->partial.release();
    db->st_q.release();
    db->st_out.release();
    delete db;
}

int ovs_desc_db_device_bytes(const ovs_desc_db* db, int64_t* bytes) {
    if (!db || !bytes) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null argument");
    }
    *bytes = static_cast<int64_t>(db->own.bytes + db->partial.bytes + db->st_q.bytes + db->st_out.bytes);
    return OVS_OK;
}

int ovs_desc_db_upload(ovs_desc_db* db, const uint8_t* descs, int64_t n_rows, int64_t global_row_offset) {
    if (!db || n_rows < 0 || (n_rows > 0 && !descs)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    ovs::device_guard guard(db->device);
    int rc = db->own.reserve(std::max<size_t>(static_cast<size_t>(n_rows) * 32, 256));
    if (rc != OVS_OK) {
        return rc;
    }
    if (n_rows) {
        OVS_HIP(hipMemcpy(db->own.ptr, descs, static_cast<size_t>(n_rows) * 32, hipMemcpyHostToDevice));
    }
    db->d_rows = db->own.as<uint8_t>();
    db->n_rows = n_rows;
    db->row_offset = global_row_offset;
    return OVS_OK;
}

int ovs_desc_db_attach_dev(ovs_desc_db* db, const uint8_t* d_descs, int64_t n_rows, int64_t global_row_offset) {
    if (!db || n_rows < 0 || (n_rows > 0 && !d_descs)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (reinterpret_cast<uintptr_t>(d_descs) & 15) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "database rows must be 16-byte aligned");
    }
    db->d_rows = d_descs;
    db->n_rows = n_rows;
    db->row_offset = global_row_offset;
    return OVS_OK;
}

namespace {
// launch plan of a search: which kernel, how the rows are cut into slices
struct knn_plan {
    bool mfma;
    int slices, rows_per_slice;
    size_t partial_bytes;
};

knn_plan plan_search(const ovs_desc_db* db, int nq) {
    knn_plan p;
    // 128 queries and more: matrix-core kernel, row slices sized so that the grid is about two
    // workgroups per compute unit (long slices keep the per-lane best / second best warm)
    p.mfma = nq >= 128 && db->n_rows > 0;
    p.rows_per_slice = kKnnRowsPerBlock;
    if (p.mfma) {
        const int qblocks = (nq + kMmQueriesPerBlock - 1) / kMmQueriesPerBlock;
        const long long want = std::max<long long>(1, 256 * kMmBlocksPerCu / qblocks);
        const long long by_rows = std::max<long long>(1, (db->n_rows + 1023) / 1024);
        const long long ns = std::min(want, by_rows);
        long long rps = (db->n_rows + ns - 1) / ns;
        rps = std::min<long long>((rps + kMmRows - 1) / kMmRows * kMmRows, kMmMaxSliceRows);
        p.rows_per_slice = static_cast<int>(rps);
        p.slices = static_cast<int>((db->n_rows + rps - 1) / rps);
    }
    else {
        p.slices = static_cast<int>(std::max<long long>(1, (db->n_rows + kKnnRowsPerBlock - 1) / kKnnRowsPerBlock));
    }
    p.partial_bytes = static_cast<size_t>(p.slices) * nq * 2 * sizeof(u64);
    return p;
}

// `partial`: plan.partial_bytes of device memory, fully written by the search kernel before the
// finish kernel reads it (nothing has to be cleared)
int launch_search(const ovs_desc_db* db, const knn_plan& plan, const uint8_t* d_queries, int nq, u64* partial, ovs_knn2* d_out, hipStream_t st) {
    if (plan.mfma) {
        OVS_LAUNCH("k_knn2_mfma", k_knn2_mfma, dim3((nq + kMmQueriesPerBlock - 1) / kMmQueriesPerBlock, plan.slices), dim3(256), 0, st,
                   db->d_rows, db->n_rows, plan.rows_per_slice, d_queries, nq, partial);
    }
    else {
        OVS_LAUNCH("k_knn2_partial", k_knn2_partial, dim3((nq + 63) / 64, plan.slices), dim3(256), 0, st,
                   db->d_rows, db->n_rows, d_queries, nq, partial);
    }
    OVS_LAUNCH("k_knn2_finish", k_knn2_finish, dim3((nq + 255) / 256), dim3(256), 0, st,
               partial, plan.slices, nq, db->row_offset, d_out);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}
} // namespace

int ovs_desc_db_search_dev(ovs_desc_db* db, const uint8_t* d_queries, int nq, ovs_knn2* d_out, void* stream_) {
    if (!db || nq < 0 || (nq > 0 && (!d_queries || !d_out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (nq == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(db->device);
    const knn_plan plan = plan_search(db, nq);
    int rc = db->partial.reserve(plan.partial_bytes);
    if (rc != OVS_OK) {
        return rc;
    }
    return launch_search(db, plan, d_queries, nq, db->partial.as<u64>(), d_out, static_cast<hipStream_t>(stream_));
}

int ovs_desc_db_search_noalloc_dev(ovs_desc_db* db, const uint8_t* d_queries, int nq, ovs_knn2* d_out, void* stream_) {
    if (!db || nq < 0 || (nq > 0 && (!d_queries || !d_out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (nq == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(db->device);
    const knn_plan plan = plan_search(db, nq);
    if (plan.partial_bytes > kNoAllocScratchBytes) {
        return ovs::fail(OVS_ERR_CAPACITY, "ovs_desc_db_search_noalloc_dev: this search needs more than the 4 MiB of built-in scratch memory "
                                           "(slices x queries x 16 bytes); use ovs_desc_db_search_dev");
    }
    void* partial = nullptr;
    int rc = scratch_for_device(db->device, &partial); // looked up at ovs_desc_db_create; a table read here
    if (rc != OVS_OK) {
        return rc;
    }
    return launch_search(db, plan, d_queries, nq, static_cast<u64*>(partial), d_out, static_cast<hipStream_t>(stream_));
}

int ovs_desc_db_search(ovs_desc_db* db, const uint8_t* queries, int nq, ovs_knn2* out) {
    if (!db || nq < 0 || (nq > 0 && (!queries || !out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (nq == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(db->device);
    int rc;
    if ((rc = db->st_q.reserve(static_cast<size_t>(nq) * 32)) != OVS_OK) return rc;
    if ((rc = db->st_out.reserve(static_cast<size_t>(nq) * sizeof(ovs_knn2))) != OVS_OK) return rc;
    OVS_HIP(hipMemcpy(db->st_q.ptr, queries, static_cast<size_t>(nq) * 32, hipMemcpyHostToDevice));
    if ((rc = ovs_desc_db_search_dev(db, db->st_q.as<uint8_t>(), nq, db->st_out.as<ovs_knn2>(), nullptr)) != OVS_OK) return rc;
    OVS_HIP(hipDeviceSynchronize());
    OVS_HIP(hipMemcpy(out, db->st_out.ptr, static_cast<size_t>(nq) * sizeof(ovs_knn2), hipMemcpyDeviceToHost));
    return OVS_OK;
}

int ovs_knn2_merge_dev(const ovs_knn2* d_lists, int n_lists, int nq, ovs_knn2* d_out, void* stream_) {
    if (n_lists <= 0 || nq < 0 || (nq > 0 && (!d_lists || !d_out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (nq == 0) {
        return OVS_OK;
    }
    OVS_LAUNCH("k_knn2_merge", k_knn2_merge, dim3((nq + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_),
               d_lists, n_lists, nq, d_out);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

int ovs_knn2_merge(const ovs_knn2* lists, int n_lists, int nq, ovs_knn2* out) {
    // host buffers: the merge of a few lists per query is a trivial host loop
    if (n_lists <= 0 || nq < 0 || (nq > 0 && (!lists || !out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    for (int qi = 0; qi < nq; ++qi) {
        u64 m0 = kNoKey, m1 = kNoKey;
        for (int s = 0; s < n_lists; ++s) {
            const ovs_knn2& e = lists[static_cast<size_t>(s) * nq + qi];
            for (int k = 0; k < 2; ++k) {
                if (0 <= e.row[k]) {
                    const u64 key = (static_cast<u64>(e.dist[k]) << 48) | static_cast<u64>(e.row[k]);
                    if (key < m0) {
                        m1 = m0;
                        m0 = key;
                    }
                    else if (key < m1) {
                        m1 = key;
                    }
                }
            }
        }
        ovs_knn2 r;
        r.dist[0] = m0 == kNoKey ? kMaxDist : static_cast<u32>(m0 >> 48);
        r.row[0] = m0 == kNoKey ? -1 : static_cast<long long>(m0 & 0xFFFFFFFFFFFFull);
        r.dist[1] = m1 == kNoKey ? kMaxDist : static_cast<u32>(m1 >> 48);
        r.row[1] = m1 == kNoKey ? -1 : static_cast<long long>(m1 & 0xFFFFFFFFFFFFull);
        out[qi] = r;
    }
    return OVS_OK;
}

int ovs_landmark_compute_descriptors(const uint8_t* descs, const uint32_t* offsets, int n_landmarks,
                                     int32_t* best_idx, int device) {
    if (n_landmarks < 0 || (n_landmarks > 0 && (!offsets || !best_idx))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (n_landmarks == 0) {
        return OVS_OK;
    }
    const size_t total = offsets[n_landmarks];
    for (int l = 0; l < n_landmarks; ++l) {
        if (offsets[l + 1] < offsets[l]) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "offsets must be non-decreasing");
        }
        if (offsets[l + 1] - offsets[l] > 4096) {
            return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 4096 observations per landmark are not supported");
        }
    }
    if (total > 0 && !descs) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) return rc;
    const size_t nl = n_landmarks;
    if ((rc = ctx->arena.reserve(al(std::max<size_t>(total, 1) * 32) + al((nl + 1) * 4) + al(nl * 4))) != OVS_OK) return rc;
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_desc = cv.take<uint8_t>(std::max<size_t>(total, 1) * 32);
    auto* d_off = cv.take<u32>(nl + 1);
    auto* d_best = cv.take<int>(nl);
    if (total) {
        OVS_HIP(hipMemcpyAsync(d_desc, descs, total * 32, hipMemcpyHostToDevice, ctx->stream));
    }
    OVS_HIP(hipMemcpyAsync(d_off, offsets, (nl + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
    OVS_LAUNCH("k_landmark_median", k_landmark_median, dim3(n_landmarks), dim3(64), 0, ctx->stream, d_desc, d_off, n_landmarks, d_best);
    OVS_HIP(hipMemcpyAsync(best_idx, d_best, nl * 4, hipMemcpyDeviceToHost, ctx->stream));
    OVS_HIP(hipStreamSynchronize(ctx->stream));
    return OVS_OK;
}

} // extern "C"
