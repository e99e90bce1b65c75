// ovs_match.hip -- binary-descriptor matching for gfx950: popcount Hamming kernels behind the
// match::{projection, stereo, robust, bow_tree} entry points of include/ovs.h.
//
// None of this is a dense contraction, so there is no MFMA here: distances are v_xor + v_bcnt over
// 8 dwords, selections are wave-level min / argmin reductions.
//
// Projection-window matchers (src/openvslam/match/projection.cc) share three kernels:
//   k_grid_build       counting sort of the searched frame's keypoints into the camera grid, cells
//                      flattened x-major with ascending keypoint index inside a cell: the iteration
//                      order of data::get_keypoints_in_cell (src/openvslam/data/common.cc:156-181)
//   k_proj_candidates  one wavefront per query: scans the window's cells, applies the gates that
//                      do not depend on earlier queries and keeps the K smallest
//                      (distance, scan position) keys
//   k_proj_resolve     one wavefront per matching problem: walks the queries in the reference's
//                      order, takes the first candidate not yet claimed (strict '<' + scan order ==
//                      lexicographic minimum of the key), falls back to a full rescan of the window
//                      when a list is exhausted, then runs the rotation-histogram check
// which reproduces the loop-carried dependency `frm.landmarks_[best] = lm` exactly.
#include "ovs_internal.h"
#include "std_sort_port.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr u32 kThrLow = 50;   // HAMMING_DIST_THR_LOW, match/base.h:13
constexpr u32 kThrHigh = 100; // HAMMING_DIST_THR_HIGH, match/base.h:14
constexpr u32 kMaxDist = 256; // MAX_HAMMING_DIST, match/base.h:15
constexpr int kTopK = 4;
constexpr u64 kNoKey = ~0ull;

enum proj_kind {
    kCurrLast = 0,   // projection::match_current_and_last_frames
    kFrameLms = 1,   // projection::match_frame_and_landmarks
    kFrameKeyfrm = 2, // projection::match_frame_and_keyframe
    kSim3 = 3,       // projection::match_by_Sim3_transform
    kOneWay = 4      // one direction of projection::match_keyframes_mutually
};

struct frame_dev {
    int n;
    const ovs_keypoint* kps;
    const uint8_t* desc;
    const float* sxr; // stereo_x_right or nullptr
    const float* scale_factors;
    int num_levels;
    ovs_grid grid;
    const u32* cell_start; // num_cols * num_rows + 1
    const u32* cell_items; // n
};

struct query_dev {
    int n;
    const float* reproj_xy;
    const float* x_right;
    const int* scale_level;
    const float* angle;
    const uint8_t* desc;
    const uint8_t* valid;
    const uint8_t* has_obs;
};

struct proj_params {
    int kind;
    float margin;
    int direction;
    int last_levels;
    float lowe;
    u32 thr;
    int check_orientation;
};

__device__ __forceinline__ int cv_floor_dd(double v) {
    const int i = (int)v;
    return i - (i > v);
}
__device__ __forceinline__ int cv_ceil_dd(double v) {
    const int i = (int)v;
    return i + (i < v);
}

__device__ __forceinline__ u32 hamming8(const u32* a, const u32* b) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d += __popc(a[i] ^ b[i]);
    }
    return d;
}

__device__ __forceinline__ void load_desc(const uint8_t* p, u32* out) {
    const uint4 lo = *reinterpret_cast<const uint4*>(p);
    const uint4 hi = *reinterpret_cast<const uint4*>(p + 16);
    out[0] = lo.x; out[1] = lo.y; out[2] = lo.z; out[3] = lo.w;
    out[4] = hi.x; out[5] = hi.y; out[6] = hi.z; out[7] = hi.w;
}

__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const u32 lo = __shfl_xor((u32)v, d);
        const u32 hi = __shfl_xor((u32)(v >> 32), d);
        const u64 o = ((u64)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ u32 wave_sum_u32(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d);
    }
    return v;
}

// ---- grid ---------------------------------------------------------------------------------------
// data::get_cell_indices, src/openvslam/data/common.h:59-64
__device__ __forceinline__ bool cell_of(const ovs_grid& g, float x, float y, int& cx, int& cy) {
    cx = cv_floor_dd((double)(x - g.min_x) * g.inv_cell_width);
    cy = cv_floor_dd((double)(y - g.min_y) * g.inv_cell_height);
    return 0 <= cx && cx < (int)g.num_cols && 0 <= cy && cy < (int)g.num_rows;
}

// One workgroup per frame.  counts -> exclusive scan -> scatter -> per-cell ascending sort.
__global__ __launch_bounds__(256) void k_grid_build(const ovs_keypoint* __restrict__ kps_all, const int* __restrict__ counts, int fixed_n,
                                                    int cap, ovs_grid g, u32* __restrict__ cell_start_all, u32* __restrict__ cell_items_all) {
    extern __shared__ u32 sm[];
    const int f = blockIdx.x;
    const int n = counts ? max(min(counts[f], cap), 0) : fixed_n;
    const int ncell = (int)(g.num_cols * g.num_rows);
    const ovs_keypoint* kps = kps_all + (size_t)f * cap;
    u32* cell_start = cell_start_all + (size_t)f * (ncell + 1);
    u32* cell_items = cell_items_all + (size_t)f * cap;
    u32* cnt = sm;            // ncell + 1
    u32* cur = sm + ncell + 1; // ncell
    __shared__ u32 red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int c = tid; c <= ncell; c += 256) {
        cnt[c] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        int cx, cy;
        if (cell_of(g, kps[i].x, kps[i].y, cx, cy)) {
            atomicAdd(&cnt[cx * g.num_rows + cy], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of cnt[0..ncell]
    {
        const int per = (ncell + 1 + 255) / 256;
        const int b = min(tid * per, ncell + 1), e = min(b + per, ncell + 1);
        u32 sum = 0;
        for (int i = b; i < e; ++i) {
            sum += cnt[i];
        }
        u32 inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 t = __shfl_up(inc, d);
            if (lane >= d) {
                inc += t;
            }
        }
        if (lane == 63) {
            red[wave] = inc;
        }
        __syncthreads();
        u32 woff = 0;
        for (int w = 0; w < wave; ++w) {
            woff += red[w];
        }
        u32 run = woff + inc - sum;
        for (int i = b; i < e; ++i) {
            const u32 v = cnt[i];
            cnt[i] = run;
            run += v;
        }
    }
    __syncthreads();
    for (int c = tid; c <= ncell; c += 256) {
        cell_start[c] = cnt[c];
        if (c < ncell) {
            cur[c] = cnt[c];
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        int cx, cy;
        if (cell_of(g, kps[i].x, kps[i].y, cx, cy)) {
            const u32 pos = atomicAdd(&cur[cx * g.num_rows + cy], 1u);
            cell_items[pos] = (u32)i;
        }
    }
    __syncthreads();
    // insertion order inside a cell is ascending keypoint index (common.cc:111-116)
    for (int c = tid; c < ncell; c += 256) {
        const u32 b = cnt[c], e = cnt[c + 1];
        for (u32 i = b + 1; i < e; ++i) {
            const u32 v = cell_items[i];
            u32 j = i;
            while (j > b && cell_items[j - 1] > v) {
                cell_items[j] = cell_items[j - 1];
                --j;
            }
            cell_items[j] = v;
        }
    }
}

// ---- window scan --------------------------------------------------------------------------------
struct window {
    int min_cx, max_cx, min_cy, max_cy;
    bool empty;
};

// the cell range of data::get_keypoints_in_cell, src/openvslam/data/common.cc:134-152
__device__ __forceinline__ window window_of(const ovs_grid& g, float ref_x, float ref_y, float margin) {
    window w;
    w.empty = true;
    w.min_cx = max(0, cv_floor_dd((double)(ref_x - g.min_x - margin) * g.inv_cell_width));
    if ((int)g.num_cols <= w.min_cx) {
        return w;
    }
    w.max_cx = min((int)g.num_cols - 1, cv_ceil_dd((double)(ref_x - g.min_x + margin) * g.inv_cell_width));
    if (w.max_cx < 0) {
        return w;
    }
    w.min_cy = max(0, cv_floor_dd((double)(ref_y - g.min_y - margin) * g.inv_cell_height));
    if ((int)g.num_rows <= w.min_cy) {
        return w;
    }
    w.max_cy = min((int)g.num_rows - 1, cv_ceil_dd((double)(ref_y - g.min_y + margin) * g.inv_cell_height));
    if (w.max_cy < 0) {
        return w;
    }
    w.empty = w.max_cx < w.min_cx || w.max_cy < w.min_cy;
    return w;
}

struct query_ctx {
    float rx, ry, margin, xr;
    int min_level, max_level; // gate of get_keypoints_in_cell
    bool ugate;               // unsigned level gate of projection.cc:390,515,598
    u32 sl;
    bool stereo_gate;
};

__device__ __forceinline__ bool make_query_ctx(const proj_params& pp, const frame_dev& f, const query_dev& q, int qi, query_ctx& c) {
    const int sl = q.scale_level[qi];
    if (sl < 0 || f.num_levels <= sl) {
        return false;
    }
    c.rx = q.reproj_xy[2 * qi];
    c.ry = q.reproj_xy[2 * qi + 1];
    c.margin = __fmul_rn(pp.margin, f.scale_factors[sl]);
    c.xr = q.x_right ? q.x_right[qi] : -1.0f;
    c.sl = (u32)sl;
    c.ugate = false;
    c.stereo_gate = false;
    switch (pp.kind) {
        case kCurrLast:
            if (pp.direction == OVS_DIR_FORWARD) {
                c.min_level = sl;
                c.max_level = pp.last_levels - 1;
            }
            else if (pp.direction == OVS_DIR_BACKWARD) {
                c.min_level = 0;
                c.max_level = sl;
            }
            else {
                c.min_level = sl - 1;
                c.max_level = sl + 1;
            }
            c.stereo_gate = true;
            break;
        case kFrameLms:
            c.min_level = sl - 1;
            c.max_level = sl;
            c.stereo_gate = true;
            break;
        case kFrameKeyfrm:
            c.min_level = sl - 1;
            c.max_level = sl + 1;
            break;
        default:
            c.min_level = -1;
            c.max_level = -1;
            c.ugate = true;
            break;
    }
    return true;
}

// key = distance (9 bits) | cell sequence in the scan (17) | position inside the cell (18) | index (20)
__device__ __forceinline__ u64 make_key(u32 dist, u32 cellseq, u32 j, u32 idx) {
    return ((u64)dist << 55) | ((u64)cellseq << 38) | ((u64)j << 20) | idx;
}
__device__ __forceinline__ u32 key_dist(u64 k) { return (u32)(k >> 55); }
__device__ __forceinline__ u32 key_idx(u64 k) { return (u32)(k & 0xFFFFFu); }

// Wave-cooperative scan of one query window.  Returns the K smallest keys (all lanes hold them)
// among the keypoints that pass the window / level / stereo gates and are neither statically
// occupied nor set in `blocked` (a bitmap in LDS, may be null); `total` counts the qualifying ones.
template <int K>
__device__ void window_topk(const frame_dev& f, const query_ctx& c, const u32* qd, const uint8_t* occupied,
                            const u32* blocked, u64 (&top)[K], u32& total) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        top[k] = kNoKey;
    }
    total = 0;
    const window w = window_of(f.grid, c.rx, c.ry, c.margin);
    if (w.empty) {
        return;
    }
    const bool check_level = (0 < c.min_level) || (0 <= c.max_level);
    const int ncy = w.max_cy - w.min_cy + 1;
    const int ncell = (w.max_cx - w.min_cx + 1) * ncy;
    u64 loc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        loc[k] = kNoKey;
    }
    u32 cnt = 0;
    for (int ci = lane; ci < ncell; ci += 64) {
        const int cx = w.min_cx + ci / ncy, cy = w.min_cy + ci % ncy;
        const int cell = cx * (int)f.grid.num_rows + cy;
        const u32 b = f.cell_start[cell], e = f.cell_start[cell + 1];
        for (u32 j = b; j < e; ++j) {
            const u32 idx = f.cell_items[j];
            const ovs_keypoint kp = f.kps[idx];
            if (check_level) {
                if (kp.octave < c.min_level) {
                    continue;
                }
                if (0 <= c.max_level && c.max_level < kp.octave) {
                    continue;
                }
            }
            const float dist_x = __fsub_rn(kp.x, c.rx);
            const float dist_y = __fsub_rn(kp.y, c.ry);
            if (!(fabsf(dist_x) < c.margin && fabsf(dist_y) < c.margin)) {
                continue;
            }
            if (occupied && occupied[idx]) {
                continue;
            }
            if (blocked && ((blocked[idx >> 5] >> (idx & 31)) & 1u)) {
                continue;
            }
            if (c.ugate) {
                const u32 oct = (u32)kp.octave;
                if (oct < c.sl - 1u || c.sl < oct) {
                    continue;
                }
            }
            if (c.stereo_gate && f.sxr) {
                const float sx = f.sxr[idx];
                if (sx > 0) {
                    const float err = fabsf(__fsub_rn(c.xr, sx));
                    if (c.margin < err) {
                        continue;
                    }
                }
            }
            u32 cd[8];
            load_desc(f.desc + 32 * (size_t)idx, cd);
            u64 key = make_key(hamming8(qd, cd), (u32)ci, j - b, idx);
            ++cnt;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (key < loc[k]) {
                    const u64 t = loc[k];
                    loc[k] = key;
                    key = t;
                }
            }
        }
    }
    total = wave_sum_u32(cnt);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const u64 m = wave_min_u64(loc[0]);
        top[k] = m;
        if (loc[0] == m && m != kNoKey) {
#pragma unroll
            for (int t = 0; t + 1 < K; ++t) {
                loc[t] = loc[t + 1];
            }
            loc[K - 1] = kNoKey;
        }
    }
}

// One wavefront per query; 4 queries per workgroup.  blockIdx.y = problem (frame pair).
struct proj_problem {
    frame_dev frm;
    query_dev q;
    const uint8_t* occupied; // static occupancy of the searched frame or nullptr
    u64* lists;              // q.n * kTopK
    u32* totals;             // q.n
    int* matched;            // frm.n
    u32* num_matches;
    int* acc_idx;            // q.n  accepted keypoint index or -1
    int* err;
};

__global__ __launch_bounds__(256) void k_proj_candidates(const proj_problem* __restrict__ problems, proj_params pp) {
    const proj_problem& P = problems[blockIdx.y];
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= P.q.n) {
        return;
    }
    u64 top[kTopK];
    u32 total = 0;
#pragma unroll
    for (int k = 0; k < kTopK; ++k) {
        top[k] = kNoKey;
    }
    query_ctx c;
    if (P.q.valid[qi]) {
        if (make_query_ctx(pp, P.frm, P.q, qi, c)) {
            u32 qd[8];
            load_desc(P.q.desc + 32 * (size_t)qi, qd);
            window_topk<kTopK>(P.frm, c, qd, P.occupied, nullptr, top, total);
        }
        else if (lane == 0) {
            *P.err = OVS_ERR_OUT_OF_RANGE;
        }
    }
    if (lane < kTopK) {
        u64 v = top[0];
#pragma unroll
        for (int k = 1; k < kTopK; ++k) {
            v = lane == k ? top[k] : v;
        }
        P.lists[(size_t)qi * kTopK + lane] = v;
    }
    if (lane == 0) {
        P.totals[qi] = total;
    }
}

__device__ __forceinline__ int angle_bin(float delta_angle) {
    // angle_checker::append_delta_angle, match/angle_checker.h:72-83 (30 bins)
    if (delta_angle < 0.0f) {
        delta_angle = __fadd_rn(delta_angle, 360.0f);
    }
    if (360.0f <= delta_angle) {
        delta_angle = __fsub_rn(delta_angle, 360.0f);
    }
    return __float2int_rn(__fmul_rn(delta_angle, 1.0f / 30));
}

// Sequential resolution, one wavefront per problem (the whole wave executes the same decisions).
__global__ __launch_bounds__(64) void k_proj_resolve(const proj_problem* __restrict__ problems, proj_params pp) {
    extern __shared__ u32 blocked[]; // bitmap over the searched frame's keypoints
    __shared__ u32 hist[32];
    __shared__ u32 order[32];
    const proj_problem& P = problems[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = P.frm.n, nq = P.q.n;
    const int words = (n + 31) >> 5;
    for (int i = lane; i < words; i += 64) {
        blocked[i] = 0;
    }
    if (lane < 32) {
        hist[lane] = 0;
    }
    for (int i = lane; i < n; i += 64) {
        P.matched[i] = -1;
    }
    __syncthreads();

    const bool excl = pp.kind != kOneWay;
    u32 num = 0;
    for (int base = 0; base < nq; base += 64) {
        const int mq = base + lane;
        u64 mykeys[kTopK];
        u32 mytot = 0;
        u32 myvalid = 0;
#pragma unroll
        for (int k = 0; k < kTopK; ++k) {
            mykeys[k] = kNoKey;
        }
        if (mq < nq) {
            myvalid = P.q.valid[mq];
            mytot = P.totals[mq];
#pragma unroll
            for (int k = 0; k < kTopK; ++k) {
                mykeys[k] = P.lists[(size_t)mq * kTopK + k];
            }
        }
        int myacc = -1;
        const int lim = min(64, nq - base);
        for (int j = 0; j < lim; ++j) {
            const int qi = base + j;
            const u32 valid = __shfl(myvalid, j);
            const u32 tot = __shfl(mytot, j);
            if (!valid || tot == 0) {
                continue;
            }
            u64 keys[kTopK];
#pragma unroll
            for (int k = 0; k < kTopK; ++k) {
                const u32 lo = __shfl((u32)mykeys[k], j);
                const u32 hi = __shfl((u32)(mykeys[k] >> 32), j);
                keys[k] = ((u64)hi << 32) | lo;
            }
            const int needed = pp.kind == kFrameLms ? 2 : 1;
            u64 best = kNoKey, second = kNoKey;
            int found = 0;
#pragma unroll
            for (int k = 0; k < kTopK; ++k) {
                const u64 key = keys[k];
                if (key != kNoKey && found < 2) {
                    const u32 idx = key_idx(key);
                    const bool b = excl && ((blocked[idx >> 5] >> (idx & 31)) & 1u);
                    if (!b) {
                        if (found == 0) {
                            best = key;
                        }
                        else {
                            second = key;
                        }
                        ++found;
                    }
                }
            }
            if (found < needed && tot > (u32)kTopK) {
                // every listed candidate was claimed by an earlier query: rescan the whole window
                query_ctx c;
                make_query_ctx(pp, P.frm, P.q, qi, c);
                u32 qd[8];
                load_desc(P.q.desc + 32 * (size_t)qi, qd);
                u64 top2[2];
                u32 t2;
                window_topk<2>(P.frm, c, qd, P.occupied, excl ? blocked : nullptr, top2, t2);
                best = top2[0];
                second = top2[1];
            }
            if (best == kNoKey) {
                continue;
            }
            const u32 bd = key_dist(best);
            const u32 bidx = key_idx(best);
            bool accept;
            switch (pp.kind) {
                case kCurrLast:
                    accept = bd <= kThrHigh;
                    break;
                case kFrameLms: {
                    accept = bd <= kThrHigh;
                    if (accept) {
                        // Lowe's ratio only between candidates of the same octave, projection.cc:69-73
                        const int bl = P.frm.kps[bidx].octave;
                        const int slv = second == kNoKey ? -1 : P.frm.kps[key_idx(second)].octave;
                        const u32 sd = second == kNoKey ? kMaxDist : key_dist(second);
                        if (bl == slv && (float)bd > __fmul_rn(pp.lowe, (float)sd)) {
                            accept = false;
                        }
                    }
                    break;
                }
                case kFrameKeyfrm:
                    accept = bd <= pp.thr;
                    break;
                case kSim3:
                    accept = bd <= kThrLow;
                    break;
                default:
                    accept = bd <= kThrHigh;
                    break;
            }
            if (!accept) {
                continue;
            }
            if (pp.kind == kOneWay) {
                if (lane == 0) {
                    P.acc_idx[qi] = (int)bidx; // best index of the query, no exclusion
                }
                continue;
            }
            ++num;
            const bool blocks = (pp.kind == kFrameKeyfrm || pp.kind == kSim3) ? true : (P.q.has_obs ? P.q.has_obs[qi] != 0 : true);
            if (lane == 0) {
                P.matched[bidx] = qi;
                if (blocks) {
                    blocked[bidx >> 5] |= 1u << (bidx & 31);
                }
            }
            if (lane == j) {
                myacc = (int)bidx;
            }
            __syncthreads();
        }
        if (mq < nq && pp.kind != kOneWay) {
            P.acc_idx[mq] = myacc;
        }
    }
    __syncthreads();

    const bool orient = pp.check_orientation && (pp.kind == kCurrLast || pp.kind == kFrameKeyfrm);
    if (orient) {
        // rotation consistency, match/angle_checker.h: keep the three fullest of 30 bins
        for (int qi = lane; qi < nq; qi += 64) {
            const int idx = P.acc_idx[qi];
            if (0 <= idx) {
                int bin = angle_bin(__fsub_rn(P.q.angle[qi], P.frm.kps[idx].angle));
                if (bin < 0 || 30 <= bin) {
                    *P.err = OVS_ERR_OUT_OF_RANGE; // angle_histogram_.at(bin) would throw
                    bin = 0;
                }
                atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        if (lane == 0) {
            ovs_sort::index_sort_by_size(hist, order, 30);
        }
        __syncthreads();
        const u32 v0 = order[0], v1 = order[1], v2 = order[2];
        u32 removed = 0;
        for (int qi = lane; qi < nq; qi += 64) {
            const int idx = P.acc_idx[qi];
            if (0 <= idx) {
                int bin = angle_bin(__fsub_rn(P.q.angle[qi], P.frm.kps[idx].angle));
                bin = (bin < 0 || 30 <= bin) ? 0 : bin;
                if ((u32)bin != v0 && (u32)bin != v1 && (u32)bin != v2) {
                    P.matched[idx] = -2;
                    ++removed;
                }
            }
        }
        removed = wave_sum_u32(removed);
        num -= removed;
    }
    if (lane == 0 && pp.kind != kOneWay) {
        *P.num_matches = num;
    }
}

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

__device__ __forceinline__ void knn_push(u64 key, u64& b0, u64& b1) {
    if (key < b0) {
        b1 = b0;
        b0 = key;
    }
    else if (key < b1) {
        b1 = key;
    }
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
    u64 b0 = kNoKey, b1 = kNoKey;
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
            knn_push(((u64)d << 48) | (u64)(base + j), b0, b1);
        }
    }
    wb[wave][lane][0] = b0;
    wb[wave][lane][1] = b1;
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
    ovs_knn2 r;
    r.dist[0] = m0 == kNoKey ? kMaxDist : (u32)(m0 >> 48);
    r.row[0] = m0 == kNoKey ? -1 : (long long)(m0 & 0xFFFFFFFFFFFFull) + row_offset;
    r.dist[1] = m1 == kNoKey ? kMaxDist : (u32)(m1 >> 48);
    r.row[1] = m1 == kNoKey ? -1 : (long long)(m1 & 0xFFFFFFFFFFFFull) + row_offset;
    out[qi] = r;
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
    ovs_knn2 r;
    r.dist[0] = m0 == kNoKey ? kMaxDist : (u32)(m0 >> 48);
    r.row[0] = m0 == kNoKey ? -1 : (long long)(m0 & 0xFFFFFFFFFFFFull);
    r.dist[1] = m1 == kNoKey ? kMaxDist : (u32)(m1 >> 48);
    r.row[1] = m1 == kNoKey ? -1 : (long long)(m1 & 0xFFFFFFFFFFFFull);
    out[qi] = r;
}

// ---- host-side plumbing --------------------------------------------------------------------------
struct thread_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    ovs::dev_buffer arena;
};

thread_ctx* ctx_for(int device) {
    static thread_local std::vector<thread_ctx> ctxs;
    for (auto& c : ctxs) {
        if (c.device == device) {
            return &c;
        }
    }
    ctxs.emplace_back();
    ctxs.back().device = device;
    return &ctxs.back();
}

struct carver {
    uint8_t* base = nullptr;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        T* p = reinterpret_cast<T*>(base + off);
        off += (n * sizeof(T) + 255) & ~static_cast<size_t>(255);
        return p;
    }
};

inline size_t al(size_t bytes) { return (bytes + 255) & ~static_cast<size_t>(255); }

int check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        return ovs::fail(OVS_ERR_HIP, "no HIP device available (this library has no CPU path)");
    }
    if (device < 0 || device >= ndev) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    }
    return OVS_OK;
}

int ensure_stream(thread_ctx* c) {
    if (!c->stream) {
        OVS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    return OVS_OK;
}

int grid_lds_bytes(const ovs_grid& g) { return static_cast<int>((2 * g.num_cols * g.num_rows + 1) * sizeof(u32)); }

int validate_frame(const ovs_frame_view* f) {
    if (!f || f->num_keypts < 0 || (f->num_keypts > 0 && (!f->undist_keypts || !f->descriptors)) || !f->scale_factors
        || f->num_scale_levels <= 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid frame view");
    }
    if (f->num_keypts >= (1 << 20)) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 2^20 keypoints per frame are not supported");
    }
    const unsigned long long ncell = static_cast<unsigned long long>(f->grid.num_cols) * f->grid.num_rows;
    if (ncell == 0 || ncell > 16384) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "grids with more than 16384 cells are not supported");
    }
    return OVS_OK;
}

// Runs one projection-matching problem given host views.
int run_projection(const ovs_frame_view* frm, const ovs_proj_queries* q, const uint8_t* occupied, proj_params pp,
                   int32_t* matched_out, int32_t* best_of_query_out, uint32_t* num_matches, int device) {
    int rc = validate_frame(frm);
    if (rc != OVS_OK) {
        return rc;
    }
    if (!q || q->n < 0 || (q->n > 0 && (!q->reproj_xy || !q->scale_level || !q->desc || !q->valid))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid query set");
    }
    if (pp.check_orientation && (pp.kind == kCurrLast || pp.kind == kFrameKeyfrm) && q->n > 0 && !q->angle) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "query angles are required for the orientation check");
    }
    if ((rc = check_device(device)) != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    const size_t n = frm->num_keypts, nq = q->n, L = frm->num_scale_levels;
    const size_t ncell = static_cast<size_t>(frm->grid.num_cols) * frm->grid.num_rows;
    const size_t total = al(n * sizeof(ovs_keypoint)) + al(n * 32) + al(n * 4) + al(L * 4) + al((ncell + 1) * 4) + al(n * 4) + al(n)
                         + al(nq * 8) + al(nq * 4) * 3 + al(nq * 32) + al(nq) * 2
                         + al(nq * kTopK * 8) + al(nq * 4) * 2 + al(n * 4) + al(256) * 3 + al(sizeof(proj_problem));
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    hipStream_t st = ctx->stream;

    auto up = [&](void* dst, const void* src, size_t bytes) -> hipError_t {
        if (!bytes) {
            return hipSuccess;
        }
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    };

    proj_problem P;
    std::memset(&P, 0, sizeof(P));
    auto* d_kps = cv.take<ovs_keypoint>(n);
    auto* d_desc = cv.take<uint8_t>(n * 32);
    auto* d_sxr = cv.take<float>(n);
    auto* d_sf = cv.take<float>(L);
    auto* d_cell_start = cv.take<u32>(ncell + 1);
    auto* d_cell_items = cv.take<u32>(n);
    auto* d_occ = cv.take<uint8_t>(n);
    auto* d_rxy = cv.take<float>(nq * 2);
    auto* d_xr = cv.take<float>(nq);
    auto* d_sl = cv.take<int>(nq);
    auto* d_ang = cv.take<float>(nq);
    auto* d_qdesc = cv.take<uint8_t>(nq * 32);
    auto* d_valid = cv.take<uint8_t>(nq);
    auto* d_hasobs = cv.take<uint8_t>(nq);
    auto* d_lists = cv.take<u64>(nq * kTopK);
    auto* d_totals = cv.take<u32>(nq);
    auto* d_acc = cv.take<int>(nq);
    auto* d_matched = cv.take<int>(n);
    auto* d_num = cv.take<u32>(1);
    auto* d_err = cv.take<int>(1);
    auto* d_prob = cv.take<proj_problem>(1);

    OVS_HIP(up(d_kps, frm->undist_keypts, n * sizeof(ovs_keypoint)));
    OVS_HIP(up(d_desc, frm->descriptors, n * 32));
    if (frm->stereo_x_right) {
        OVS_HIP(up(d_sxr, frm->stereo_x_right, n * 4));
    }
    OVS_HIP(up(d_sf, frm->scale_factors, L * 4));
    if (occupied) {
        OVS_HIP(up(d_occ, occupied, n));
    }
    OVS_HIP(up(d_rxy, q->reproj_xy, nq * 8));
    if (q->x_right) {
        OVS_HIP(up(d_xr, q->x_right, nq * 4));
    }
    OVS_HIP(up(d_sl, q->scale_level, nq * 4));
    if (q->angle) {
        OVS_HIP(up(d_ang, q->angle, nq * 4));
    }
    OVS_HIP(up(d_qdesc, q->desc, nq * 32));
    OVS_HIP(up(d_valid, q->valid, nq));
    if (q->has_observation) {
        OVS_HIP(up(d_hasobs, q->has_observation, nq));
    }
    OVS_HIP(hipMemsetAsync(d_num, 0, 4, st));
    OVS_HIP(hipMemsetAsync(d_err, 0, 4, st));
    if (nq) {
        OVS_HIP(hipMemsetAsync(d_acc, 0xFF, nq * 4, st));
    }

    P.frm.n = static_cast<int>(n);
    P.frm.kps = d_kps;
    P.frm.desc = d_desc;
    P.frm.sxr = frm->stereo_x_right ? d_sxr : nullptr;
    P.frm.scale_factors = d_sf;
    P.frm.num_levels = static_cast<int>(L);
    P.frm.grid = frm->grid;
    P.frm.cell_start = d_cell_start;
    P.frm.cell_items = d_cell_items;
    P.q.n = static_cast<int>(nq);
    P.q.reproj_xy = d_rxy;
    P.q.x_right = q->x_right ? d_xr : nullptr;
    P.q.scale_level = d_sl;
    P.q.angle = q->angle ? d_ang : nullptr;
    P.q.desc = d_qdesc;
    P.q.valid = d_valid;
    P.q.has_obs = q->has_observation ? d_hasobs : nullptr;
    P.occupied = occupied ? d_occ : nullptr;
    P.lists = d_lists;
    P.totals = d_totals;
    P.matched = d_matched;
    P.num_matches = d_num;
    P.acc_idx = d_acc;
    P.err = d_err;
    OVS_HIP(up(d_prob, &P, sizeof(P)));

    OVS_LAUNCH("k_grid_build", k_grid_build, dim3(1), dim3(256), grid_lds_bytes(frm->grid), st,
               d_kps, static_cast<const int*>(nullptr), static_cast<int>(n), static_cast<int>(std::max<size_t>(n, 1)), frm->grid,
               d_cell_start, d_cell_items);
    if (nq) {
        OVS_LAUNCH("k_proj_candidates", k_proj_candidates, dim3((nq + 3) / 4, 1), dim3(256), 0, st, d_prob, pp);
    }
    OVS_LAUNCH("k_proj_resolve", k_proj_resolve, dim3(1), dim3(64), ((n + 31) / 32 + 1) * 4, st, d_prob, pp);
    OVS_HIP(hipGetLastError());

    int err = 0;
    uint32_t num = 0;
    if (matched_out && n) {
        OVS_HIP(hipMemcpyAsync(matched_out, d_matched, n * 4, hipMemcpyDeviceToHost, st));
    }
    if (best_of_query_out && nq) {
        OVS_HIP(hipMemcpyAsync(best_of_query_out, d_acc, nq * 4, hipMemcpyDeviceToHost, st));
    }
    OVS_HIP(hipMemcpyAsync(&num, d_num, 4, hipMemcpyDeviceToHost, st));
    OVS_HIP(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
    OVS_HIP(hipStreamSynchronize(st));
    if (err != 0) {
        return ovs::fail(err, "scale level or rotation bin out of range");
    }
    if (num_matches) {
        *num_matches = num;
    }
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

int ovs_get_cell_indices(const ovs_grid* g, float x, float y, int* cell_x, int* cell_y) {
    // data::get_cell_indices, src/openvslam/data/common.h:59-64 (pure host arithmetic)
    *cell_x = ovs::cv_floor_d(static_cast<double>(x - g->min_x) * g->inv_cell_width);
    *cell_y = ovs::cv_floor_d(static_cast<double>(y - g->min_y) * g->inv_cell_height);
    return (0 <= *cell_x && *cell_x < static_cast<int>(g->num_cols) && 0 <= *cell_y && *cell_y < static_cast<int>(g->num_rows)) ? 1 : 0;
}

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

int ovs_match_current_and_last_frames(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                      const uint8_t* curr_occupied, float margin, int direction,
                                      int last_num_scale_levels, float lowe_ratio, int check_orientation,
                                      int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kCurrLast, margin, direction, last_num_scale_levels, lowe_ratio, kThrHigh, check_orientation};
    return run_projection(curr, q, curr_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_frame_and_landmarks(const ovs_frame_view* frm, const ovs_proj_queries* q,
                                  const uint8_t* frm_occupied, float margin, float lowe_ratio,
                                  int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kFrameLms, margin, OVS_DIR_NONE, 0, lowe_ratio, kThrHigh, 0};
    return run_projection(frm, q, frm_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_frame_and_keyframe(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                 const uint8_t* curr_occupied, float margin, uint32_t hamm_dist_thr,
                                 int check_orientation,
                                 int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kFrameKeyfrm, margin, OVS_DIR_NONE, 0, 0.0f, hamm_dist_thr, check_orientation};
    return run_projection(curr, q, curr_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_by_sim3_transform(const ovs_frame_view* keyfrm, const ovs_proj_queries* q,
                                const uint8_t* keyfrm_occupied, float margin,
                                int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kSim3, margin, OVS_DIR_NONE, 0, 0.0f, kThrLow, 0};
    return run_projection(keyfrm, q, keyfrm_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_keyframe_one_way(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, float margin,
                               int32_t* best_idx_of_query, int device) {
    proj_params pp{kOneWay, margin, OVS_DIR_NONE, 0, 0.0f, kThrHigh, 0};
    return run_projection(keyfrm, q, nullptr, pp, nullptr, best_idx_of_query, nullptr, device);
}

int ovs_cross_check(const int32_t* idx2_of_1, int n1, const int32_t* idx1_of_2, int n2,
                    int32_t* out_idx2_of_1, uint32_t* num_matches) {
    // projection.cc:618-633 -- a linear pass over n1 ints, done on the host
    unsigned int num = 0;
    std::fill(out_idx2_of_1, out_idx2_of_1 + n1, -1);
    for (int i = 0; i < n1; ++i) {
        const int idx_2 = idx2_of_1[i];
        if (idx_2 < 0) {
            continue;
        }
        if (n2 <= idx_2) {
            return ovs::fail(OVS_ERR_OUT_OF_RANGE, "index out of range");
        }
        if (idx1_of_2[idx_2] == i) {
            out_idx2_of_1[i] = idx_2;
            ++num;
        }
    }
    *num_matches = num;
    return OVS_OK;
}

} // extern "C"

namespace {

// Fills the problem descriptors of a batch of consecutive-frame matches from device-side counts and
// splits the query fields out of the previous frame's keypoints.  grid = (ceil(cap / 256), n_frames).
__global__ __launch_bounds__(256) void k_batch_setup(const ovs_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                     const int* __restrict__ counts, int n_frames, int cap,
                                                     const float* __restrict__ reproj_xy, ovs_grid grid,
                                                     const float* __restrict__ scale_factors, int num_levels,
                                                     u32* cell_start, u32* cell_items, int ncell,
                                                     int* sl, float* ang, uint8_t* valid, u64* lists, u32* totals, int* acc,
                                                     int* matched, u32* num_matches, int* err, proj_problem* problems) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (f == 0) {
        if (i < cap) {
            matched[i] = -1;
        }
        if (i == 0) {
            num_matches[0] = 0;
        }
        return;
    }
    const int n_last = max(min(counts[f - 1], cap), 0);
    const size_t qo = (size_t)f * cap;
    if (i < cap) {
        const ovs_keypoint kp = kps[(size_t)(f - 1) * cap + min(i, max(n_last - 1, 0))];
        sl[qo + i] = kp.octave;
        ang[qo + i] = kp.angle;
        valid[qo + i] = (i < n_last && !(reproj_xy[2 * (qo + i)] < -1e30f)) ? 1 : 0;
        acc[qo + i] = -1;
    }
    if (i == 0) {
        proj_problem P;
        P.frm.n = max(min(counts[f], cap), 0);
        P.frm.kps = kps + qo;
        P.frm.desc = desc + qo * 32;
        P.frm.sxr = nullptr;
        P.frm.scale_factors = scale_factors;
        P.frm.num_levels = num_levels;
        P.frm.grid = grid;
        P.frm.cell_start = cell_start + (size_t)f * (ncell + 1);
        P.frm.cell_items = cell_items + qo;
        P.q.n = n_last;
        P.q.reproj_xy = reproj_xy + 2 * qo;
        P.q.x_right = nullptr;
        P.q.scale_level = sl + qo;
        P.q.angle = ang + qo;
        P.q.desc = desc + (size_t)(f - 1) * cap * 32;
        P.q.valid = valid + qo;
        P.q.has_obs = nullptr;
        P.occupied = nullptr;
        P.lists = lists + qo * kTopK;
        P.totals = totals + qo;
        P.matched = matched + qo;
        P.num_matches = num_matches + f;
        P.acc_idx = acc + qo;
        P.err = err;
        problems[f - 1] = P;
    }
}

} // namespace

extern "C" {

int ovs_match_batch_current_and_last_dev(const ovs_keypoint* d_kps, const uint8_t* d_desc,
                                         const int32_t* d_counts, int n_frames, int cap,
                                         const float* d_reproj_xy, const ovs_grid* grid,
                                         const float* scale_factors, int num_scale_levels,
                                         float margin, int check_orientation,
                                         int32_t* d_matched, uint32_t* d_num_matches,
                                         int device, void* stream_) {
    if (!d_kps || !d_desc || !d_counts || !d_reproj_xy || !grid || !scale_factors || !d_matched || !d_num_matches
        || n_frames <= 0 || cap <= 0 || num_scale_levels <= 0 || num_scale_levels > 64) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (cap >= (1 << 20)) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 2^20 keypoints per frame are not supported");
    }
    const size_t ncell = static_cast<size_t>(grid->num_cols) * grid->num_rows;
    if (ncell == 0 || ncell > 16384) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "grids with more than 16384 cells are not supported");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const size_t F = n_frames, fc = F * cap;
    const size_t total = al(64 * 4) + al(F * (ncell + 1) * 4) + al(fc * 4) * 5 + al(fc) + al(fc * kTopK * 8) + al(256) + al(F * sizeof(proj_problem));
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_sf = cv.take<float>(64);
    auto* d_cell_start = cv.take<u32>(F * (ncell + 1));
    auto* d_cell_items = cv.take<u32>(fc);
    auto* d_sl = cv.take<int>(fc);
    auto* d_ang = cv.take<float>(fc);
    auto* d_totals = cv.take<u32>(fc);
    auto* d_acc = cv.take<int>(fc);
    auto* d_valid = cv.take<uint8_t>(fc);
    auto* d_lists = cv.take<u64>(fc * kTopK);
    auto* d_err = cv.take<int>(1);
    auto* d_prob = cv.take<proj_problem>(F);
    OVS_HIP(hipMemcpyAsync(d_sf, scale_factors, static_cast<size_t>(num_scale_levels) * 4, hipMemcpyHostToDevice, st));

    proj_params pp{kCurrLast, margin, OVS_DIR_NONE, num_scale_levels, 0.0f, kThrHigh, check_orientation};
    OVS_LAUNCH("k_batch_setup", k_batch_setup, dim3((cap + 255) / 256, n_frames), dim3(256), 0, st,
               d_kps, d_desc, d_counts, n_frames, cap, d_reproj_xy, *grid, d_sf, num_scale_levels,
               d_cell_start, d_cell_items, static_cast<int>(ncell), d_sl, d_ang, d_valid, d_lists, d_totals, d_acc,
               d_matched, d_num_matches, d_err, d_prob);
    OVS_LAUNCH("k_grid_build", k_grid_build, dim3(n_frames), dim3(256), grid_lds_bytes(*grid), st,
               d_kps, d_counts, 0, cap, *grid, d_cell_start, d_cell_items);
    if (n_frames > 1) {
        OVS_LAUNCH("k_proj_candidates", k_proj_candidates, dim3((cap + 3) / 4, n_frames - 1), dim3(256), 0, st, d_prob, pp);
        OVS_LAUNCH("k_proj_resolve", k_proj_resolve, dim3(n_frames - 1), dim3(64), ((cap + 31) / 32 + 1) * 4, st, d_prob, pp);
    }
    OVS_HIP(hipGetLastError());
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

int ovs_desc_db_search_dev(ovs_desc_db* db, const uint8_t* d_queries, int nq, ovs_knn2* d_out, void* stream_) {
    if (!db || nq < 0 || (nq > 0 && (!d_queries || !d_out))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (nq == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(db->device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const int slices = static_cast<int>(std::max<long long>(1, (db->n_rows + kKnnRowsPerBlock - 1) / kKnnRowsPerBlock));
    int rc = db->partial.reserve(static_cast<size_t>(slices) * nq * 2 * sizeof(u64));
    if (rc != OVS_OK) {
        return rc;
    }
    OVS_LAUNCH("k_knn2_partial", k_knn2_partial, dim3((nq + 63) / 64, slices), dim3(256), 0, st,
               db->d_rows, db->n_rows, d_queries, nq, db->partial.as<u64>());
    OVS_LAUNCH("k_knn2_finish", k_knn2_finish, dim3((nq + 255) / 256), dim3(256), 0, st,
               db->partial.as<u64>(), slices, nq, db->row_offset, d_out);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
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

// ---- not yet on the device -------------------------------------------------------------------------
int ovs_get_keypoints_in_cell(const ovs_frame_view*, int, const float*, const float*, const int32_t*, const int32_t*,
                              uint32_t*, int, uint32_t*, int) {
    return ovs::fail(OVS_ERR_UNSUPPORTED, "not implemented yet");
}
int ovs_stereo_compute(const ovs_image_view*, const ovs_image_view*, int, const ovs_keypoint*, int, const ovs_keypoint*, int,
                       const uint8_t*, const uint8_t*, const float*, const float*, float, float, float*, float*, int) {
    return ovs::fail(OVS_ERR_UNSUPPORTED, "not implemented yet");
}
int ovs_brute_force_match(const uint8_t*, const float*, int, const uint8_t*, const float*, const uint8_t*, int, float, int,
                          int32_t*, uint32_t*, int) {
    return ovs::fail(OVS_ERR_UNSUPPORTED, "not implemented yet");
}
int ovs_bow_match_frame_and_keyframe(const ovs_bow_feat_vec*, const uint8_t*, const float*, const uint8_t*, int,
                                     const ovs_bow_feat_vec*, const uint8_t*, const float*, int, float, int, int32_t*, uint32_t*, int) {
    return ovs::fail(OVS_ERR_UNSUPPORTED, "not implemented yet");
}
int ovs_bow_match_keyframes(const ovs_bow_feat_vec*, const uint8_t*, const float*, const uint8_t*, int,
                            const ovs_bow_feat_vec*, const uint8_t*, const float*, const uint8_t*, int, float, int, int32_t*, uint32_t*, int) {
    return ovs::fail(OVS_ERR_UNSUPPORTED, "not implemented yet");
}
int ovs_match_for_triangulation(const ovs_bow_feat_vec*, const uint8_t*, const ovs_keypoint*, const double*, const float*, const uint8_t*, int,
                                const ovs_bow_feat_vec*, const uint8_t*, const ovs_keypoint*, const double*, const float*, const uint8_t*, int,
                                const double*, const double*, const float*, int, int, uint32_t*, uint32_t*, int) {
    return ovs::fail(OVS_ERR_UNSUPPORTED, "not implemented yet");
}

} // extern "C"
