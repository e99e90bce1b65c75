// ovs_projection.hip -- match::projection (src/openvslam/match/projection.cc) and the camera-grid
// window query it is built on (src/openvslam/data/common.cc:97-186) for gfx950.
//
// Kernels
//   k_grid_build       one workgroup per searched frame: counting sort of its keypoints into the
//                      camera grid.  Cells are flattened x-major (cell = cx * rows + cy) and keypoint
//                      indices ascend inside a cell, so the candidates of a window
//                      [min_cx..max_cx] x [min_cy..max_cy] are, per column cx, ONE contiguous range
//                      of the sorted arrays, visited in exactly the order of the nested loops of
//                      get_keypoints_in_cell (common.cc:156-181).  The sort also writes position,
//                      octave, index and descriptor of every keypoint in grid order, so a window
//                      scan reads contiguous memory.
//   k_proj_candidates  8 lanes per query: prefix over the window's column ranges, lanes stride over
//                      the concatenated ranges, gates that do not depend on earlier queries, popcount
//                      Hamming, and the K smallest (distance, scan position) keys per query.
//   k_proj_resolve     one wavefront per matching problem: reproduces the loop-carried dependency
//                      `frm.landmarks_[best] = lm` (projection.cc:164,190) -- see the kernel comment --
//                      and the rotation-histogram check (match/angle_checker.h).
//   k_window_list      get_keypoints_in_cell itself (index lists in the reference's order).
#include "ovs_match_common.h"
#include "std_sort_port.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

using namespace ovsm;

namespace {

constexpr int kTopK = 8;
constexpr int kGroupLive = 64; // lanes per query when a call has a few problems: a wave per query (16 lanes: 16.2 us for the 2241 queries of a frame, 64: 10.5; OVS_GROUP_LIVE=8 keeps kGroup)
constexpr int kGroup = 8; // lanes per query in k_proj_candidates (16: 0.29 ms, 8: 0.21 ms, 4: 0.46 ms per 255 x 2243 queries)

enum proj_kind {
    kCurrLast = 0,    // projection::match_current_and_last_frames
    kFrameLms = 1,    // projection::match_frame_and_landmarks
    kFrameKeyfrm = 2, // projection::match_frame_and_keyframe
    kSim3 = 3,        // projection::match_by_Sim3_transform
    kOneWay = 4,      // one direction of projection::match_keyframes_mutually
    kRange = 5,       // robust::brute_force_match / bow_tree::match_*: every query names one contiguous
                      // range of the candidate list; best <= 50 and Lowe's ratio against the runner-up
    kFuseDetect = 6,  // fuse::detect_duplication: per query best keypoint, signed level gate, <= 50
    kFuseReplace = 7, // fuse::replace_duplication: unsigned level gate + chi-square reprojection gate
    kArea = 8,        // area::match_in_consistent_area: level-0 window, ratio test, a strictly closer query takes a keypoint over
    kTri = 9          // robust::match_for_triangulation: bucket ranges, epipole / epipolar gates in double, last of equals wins
};

// what is compiled into an instance of the candidate / resolve kernels besides the tracking matchers
enum proj_mode {
    kModePlain = 0,
    kModeFuse = 1, // the double-precision chi-square gate of match::fuse
    kModeTri = 2,  // the double-precision epipolar gates of robust::match_for_triangulation
    kModeArea = 3  // per-keypoint (distance, owner) state instead of the claim bitmap
};

__host__ __device__ inline bool kind_is_one_way(int kind) { return kind == kOneWay || kind == kFuseDetect || kind == kFuseReplace; }

struct frame_dev {
    int n;
    const ovs_keypoint* kps;
    const uint8_t* desc;
    const float* sxr; // stereo_x_right or nullptr
    const float* scale_factors;
    int num_levels;
    ovs_grid grid;
    const u32* cell_start; // num_cols * num_rows + 1
    const uint4* g_entry;  // grid order: {x bits, y bits, octave, keypoint index}
    const uint4* g_desc;   // grid order: two uint4 per keypoint
    int n_entries_plus1;   // ranged problems: number of list entries + 1; 0: a grid, cell_start[cells] holds it
    const float* inv_level_sigma_sq; // kFuseReplace only
    // kTri only
    const double* bearings; // n x 3: bearings_ of the searched keyframe
    const double* tri_E;    // E_12 row-major (9) followed by the epipole in the searched keyframe (3)
    u32* tri_amb;           // [0] = number of borderline pairs seen, [1] = capacity, then (query, entry) pairs
    const u32* tri_ovr;     // [0] = number of pairs, then (query, entry) pairs whose epipolar decision is the
                            // opposite of the device's comparison (established on the host, see run_triangulation)
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
    const u32* range_lo; // kRange: candidates of query i are entries [range_lo[i], range_hi[i])
    const u32* range_hi;
    const double* reproj_xy_d; // kFuseReplace: the double-precision reprojection of the chi-square gate
    const double* bearings;    // kTri: n x 3 bearings of the query keypoints
    const double* sin_thr;     // kTri: n, sine of the epipolar residual threshold of the query's scale level
    const uint8_t* stereo;     // kTri: n, 1 = the query keypoint has a stereo x_right >= 0
};

struct proj_params {
    int kind;
    float margin;
    int direction;
    int last_levels;
    float lowe;
    u32 thr;
    int check_orientation;
    int angle_sign;       // +1: delta = query angle - candidate angle, -1: the reverse
    int per_query_output; // the rotation check clears the query's own result instead of the candidate's slot
};

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

// ---- grid ---------------------------------------------------------------------------------------
// data::get_cell_indices, src/openvslam/data/common.h:59-64
__device__ __forceinline__ bool cell_of(const ovs_grid& g, float x, float y, int& cx, int& cy) {
    cx = cv_floor_dd((double)(x - g.min_x) * g.inv_cell_width);
    cy = cv_floor_dd((double)(y - g.min_y) * g.inv_cell_height);
    return 0 <= cx && cx < (int)g.num_cols && 0 <= cy && cy < (int)g.num_rows;
}

// ITEMS_LDS: the keypoint indices in cell order (the array the sort works on) live in LDS behind the
// counters instead of in global memory: the scatter, the per-cell ordering and the final gather are
// chains of dependent accesses, and the workgroup is the only thing running for its frame.
// T threads: 256 for batches (one workgroup per frame, hundreds of frames), 1024 for a single frame that is waited for
template <bool ITEMS_LDS, int T>
__global__ __launch_bounds__(T) void k_grid_build(const ovs_keypoint* __restrict__ kps_all, const uint8_t* __restrict__ desc_all,
                                                    const int* __restrict__ counts, int fixed_n, int cap, ovs_grid g,
                                                    u32* __restrict__ cell_start_all, u32* __restrict__ items_all,
                                                    uint4* __restrict__ entry_all, uint4* __restrict__ gdesc_all) {
    extern __shared__ u32 sm[];
    __shared__ u32 red[T / 64];
    const int f = blockIdx.x;
    const int n = counts ? max(min(counts[f], cap), 0) : fixed_n;
    const int ncell = (int)(g.num_cols * g.num_rows);
    const ovs_keypoint* kps = kps_all + (size_t)f * cap;
    const uint8_t* desc = desc_all + (size_t)f * cap * 32;
    u32* cell_start = cell_start_all + (size_t)f * (ncell + 1);
    const int ncell_ = (int)(g.num_cols * g.num_rows);
    u32* items = ITEMS_LDS ? sm + 2 * ncell_ + 1 : items_all + (size_t)f * cap;
    uint4* entry = entry_all + (size_t)f * cap;
    uint4* gdesc = gdesc_all + (size_t)f * cap * 2;
    u32* cnt = sm;             // ncell + 1
    u32* cur = sm + ncell + 1; // ncell
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int c = tid; c <= ncell; c += T) {
        cnt[c] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n; i += T) {
        int cx, cy;
        if (cell_of(g, kps[i].x, kps[i].y, cx, cy)) {
            atomicAdd(&cnt[cx * g.num_rows + cy], 1u);
        }
    }
    __syncthreads();
    {
        const int per = (ncell + 1 + T - 1) / T;
        const int b = min(tid * per, ncell + 1), e = min(b + per, ncell + 1);
        u32 sum = 0;
        for (int i = b; i < e; ++i) {
            sum += cnt[i];
        }
        const u32 inc = wave_incl_scan_u32(sum);
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
    for (int c = tid; c <= ncell; c += T) {
        cell_start[c] = cnt[c];
        if (c < ncell) {
            cur[c] = cnt[c];
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += T) {
        int cx, cy;
        if (cell_of(g, kps[i].x, kps[i].y, cx, cy)) {
            const u32 pos = atomicAdd(&cur[cx * g.num_rows + cy], 1u);
            items[pos] = (u32)i;
        }
    }
    __syncthreads();
    // insertion order inside a cell is ascending keypoint index (common.cc:111-116)
    for (int c = tid; c < ncell; c += T) {
        const u32 b = cnt[c], e = cnt[c + 1];
        for (u32 i = b + 1; i < e; ++i) {
            const u32 v = items[i];
            u32 j = i;
            while (j > b && items[j - 1] > v) {
                items[j] = items[j - 1];
                --j;
            }
            items[j] = v;
        }
    }
    __syncthreads();
    const int total = (int)cnt[ncell];
    for (int j = tid; j < total; j += T) {
        const u32 i = items[j];
        const ovs_keypoint kp = kps[i];
        entry[j] = make_uint4(__float_as_uint(kp.x), __float_as_uint(kp.y), (u32)kp.octave, i);
        const u32* d = reinterpret_cast<const u32*>(desc + 32 * (size_t)i);
        gdesc[2 * j] = make_uint4(d[0], d[1], d[2], d[3]);
        gdesc[2 * j + 1] = make_uint4(d[4], d[5], d[6], d[7]);
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
    w.min_cx = w.max_cx = w.min_cy = w.max_cy = 0;
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
    bool ranged; // kRange: no window, no geometric gates
    u32 lo, hi;
    int fuse;       // 0: none, 1: signed level gate (fuse.cc:86), 2: unsigned gate + chi-square gate (fuse.cc:194-222)
    double rxd, ryd;
    // kTri: bearing of the query keypoint, sine of its residual threshold, stereo flag, query number
    double b1x, b1y, b1z, sin_thr;
    bool stereo1;
    int qi;
};

// MODE: what besides the tracking matchers is compiled in (separate kernel instances keep the double-
// precision state and gates of match::fuse and of the triangulation matcher out of their kernels)
template <int MODE>
__device__ __forceinline__ bool make_query_ctx(const proj_params& pp, const frame_dev& f, const query_dev& q, int qi, query_ctx& c) {
    c.ranged = false;
    c.lo = c.hi = 0;
    c.fuse = 0;
    c.rxd = c.ryd = 0.0;
    c.b1x = c.b1y = c.b1z = c.sin_thr = 0.0;
    c.stereo1 = false;
    c.qi = qi;
    if (pp.kind == kRange || pp.kind == kTri) {
        c.ranged = true;
        c.lo = q.range_lo[qi];
        c.hi = q.range_hi[qi];
        c.rx = c.ry = c.margin = c.xr = 0.0f;
        c.min_level = c.max_level = -1;
        c.ugate = false;
        c.sl = 0;
        c.stereo_gate = false;
        if (MODE == kModeTri) {
            c.b1x = q.bearings[3 * (size_t)qi];
            c.b1y = q.bearings[3 * (size_t)qi + 1];
            c.b1z = q.bearings[3 * (size_t)qi + 2];
            c.sin_thr = q.sin_thr[qi];
            c.stereo1 = q.stereo[qi] != 0;
        }
        return true;
    }
    const int sl = q.scale_level[qi];
    if (pp.kind == kArea) {
        // get_keypoints_in_cell(x, y, margin, scale_level_1, scale_level_1), area.cc:29-30: the margin is
        // not scaled, and a negative level switches the level gate off (common.cc:154)
        c.rx = q.reproj_xy[2 * qi];
        c.ry = q.reproj_xy[2 * qi + 1];
        c.margin = pp.margin;
        c.xr = -1.0f;
        c.sl = 0;
        c.ugate = false;
        c.stereo_gate = false;
        c.min_level = c.max_level = sl;
        return true;
    }
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
        case kFuseDetect:
            c.min_level = -1;
            c.max_level = -1;
            if (MODE == kModeFuse) {
                c.fuse = 1;
            }
            break;
        case kFuseReplace:
            c.min_level = -1;
            c.max_level = -1;
            c.ugate = true;
            if (MODE == kModeFuse) {
                c.fuse = 2;
                c.rxd = q.reproj_xy_d[2 * qi];
                c.ryd = q.reproj_xy_d[2 * qi + 1];
                c.rx = (float)c.rxd; // get_keypoints_in_cell takes floats: the doubles narrow at the call
                c.ry = (float)c.ryd;
            }
            break;
        default:
            c.min_level = -1;
            c.max_level = -1;
            c.ugate = true;
            break;
    }
    return true;
}

// list key = distance (9 bits) << 40 | tie-break position (20 bits: the entry index, monotone in the
// reference's scan position) << 20 | keypoint index (20 bits)
__device__ __forceinline__ u64 make_key(u32 dist, u32 pos, u32 idx) { return ((u64)dist << 40) | ((u64)pos << 20) | idx; }
__device__ __forceinline__ u32 key_dist(u64 k) { return (u32)(k >> 40); }
__device__ __forceinline__ u32 key_idx(u64 k) { return (u32)(k & 0xFFFFFu); }

// window / level gates of get_keypoints_in_cell (common.cc:166-178) + the per-candidate gates of
// the projection loops that do not depend on earlier queries
template <int MODE>
__device__ __forceinline__ bool passes_gates(const frame_dev& f, const query_ctx& c, const uint4& en, const uint8_t* occupied, const u32* blocked,
                                             bool matcher_gates) {
    if (c.ranged) {
        const u32 ridx = en.w;
        if (occupied && occupied[ridx]) {
            return false;
        }
        return !(blocked && ((blocked[ridx >> 5] >> (ridx & 31)) & 1u));
    }
    // kModeArea: `blocked` is the per-keypoint (distance, owner) state, tested against the distance in
    // window_topk
    const int oct = (int)en.z;
    const bool check_level = (0 < c.min_level) || (0 <= c.max_level);
    if (check_level) {
        if (oct < c.min_level) {
            return false;
        }
        if (0 <= c.max_level && c.max_level < oct) {
            return false;
        }
    }
    const float dist_x = __fsub_rn(__uint_as_float(en.x), c.rx);
    const float dist_y = __fsub_rn(__uint_as_float(en.y), c.ry);
    if (!(fabsf(dist_x) < c.margin && fabsf(dist_y) < c.margin)) {
        return false;
    }
    if (!matcher_gates) {
        return true;
    }
    const u32 idx = en.w;
    if (occupied && occupied[idx]) {
        return false;
    }
    if (MODE != kModeArea && blocked && ((blocked[idx >> 5] >> (idx & 31)) & 1u)) {
        return false;
    }
    if (c.ugate) {
        const u32 uo = (u32)oct;
        if (uo < c.sl - 1u || c.sl < uo) {
            return false;
        }
    }
    if (c.stereo_gate && f.sxr) {
        const float sx = f.sxr[idx];
        if (sx > 0) {
            const float err = fabsf(__fsub_rn(c.xr, sx));
            if (c.margin < err) {
                return false;
            }
        }
    }
    if (MODE == kModeFuse && c.fuse == 1) {
        // int scale_level, int pred_scale_level (fuse.cc:68,83-88)
        if (oct < (int)c.sl - 1 || (int)c.sl < oct) {
            return false;
        }
    }
    else if (MODE == kModeFuse && c.fuse == 2) {
        // fuse.cc:198-222: reproj is double, keypoint and x_right are float; products and sums in the
        // reference's order without contraction
        const double e_x = __dsub_rn(c.rxd, (double)__uint_as_float(en.x));
        const double e_y = __dsub_rn(c.ryd, (double)__uint_as_float(en.y));
        double err_sq = __dadd_rn(__dmul_rn(e_x, e_x), __dmul_rn(e_y, e_y));
        const float sx = f.sxr ? f.sxr[idx] : -1.0f;
        double chi = (double)5.99146f;
        if (sx >= 0) {
            const float e_r = __fsub_rn(c.xr, sx);
            err_sq = __dadd_rn(err_sq, (double)__fmul_rn(e_r, e_r));
            chi = (double)7.81473f;
        }
        if (chi < __dmul_rn(err_sq, (double)f.inv_level_sigma_sq[oct])) {
            return false;
        }
    }
    return true;
}

// Scan of one query window by a group of G lanes that share the query.  Returns (in every lane of
// the group) the K best candidates among the keypoints that pass all gates, best first, as 32-bit
// selection keys; `total` counts the candidates.
//
// Selection key = distance << 20 | entry index j in the searched frame's CSR order.  The scan walks
// the window's columns in ascending order and each column's entries in ascending j, and the CSR is
// stored column-major, so the scan position the reference breaks distance ties with is monotone in
// j: ordering by (distance, j) is the reference's order.  32-bit keys make the per-lane insertion a
// min / max pair per slot and the group minimum three DPP instructions.
constexpr u32 kNoKey32 = 0xFFFFFFFFu;
__device__ __forceinline__ u32 sel_dist(u32 k) { return k >> 20; }
__device__ __forceinline__ u32 sel_entry(u32 k) { return k & 0xFFFFFu; }
// the list format of the resolve kernel: distance, tie-break position, keypoint index
// (the triangulation matcher keeps the LAST of equally distant candidates, robust.cc:102: its keys carry
// the entry index complemented, so that the smallest key is again the winner)
template <int MODE>
__device__ __forceinline__ u64 list_key(const frame_dev& f, u32 sel) {
    const u32 entry = MODE == kModeTri ? 0xFFFFFu - sel_entry(sel) : sel_entry(sel);
    return sel == kNoKey32 ? kNoKey : make_key(sel_dist(sel), sel_entry(sel), f.g_entry[entry].w);
}

// robust::match_for_triangulation, robust.cc:106-121 and check_epipolar_constraint, robust.cc:317-335,
// for the query of `c` and the keypoint `idx` of the searched keyframe (list entry j).
//
// The reference accepts when pi/2 - |acos(cos_residual)| < thr * scale_factor.  acos takes values in
// [0, pi], so this is asin(cos_residual) < T and, asin being monotone, cos_residual < sin(T) (the
// host passes sin(T) per query); NaN (a zero normal, or |cos_residual| > 1 by rounding) is never an
// inlier.  The two forms can only disagree where rounding decides: the subtraction from pi/2 is exact
// for cos_residual <= 0.7 and acos is good to an ulp of about 2.2e-16, so pairs whose cosine lies
// within 1e-13 of sin(T) are borderline.  Those are reported to the host, which evaluates the
// reference's expression with its own libm; a pair whose host result differs from the comparison
// made here comes back in the override list and the match is redone.
__device__ __forceinline__ bool tri_gate(const frame_dev& f, const query_ctx& c, u32 idx, u32 j, bool record) {
    const double* b2 = f.bearings + 3 * (size_t)idx;
    const double b2x = b2[0], b2y = b2[1], b2z = b2[2];
    const double* E = f.tri_E;
    const bool stereo2 = f.sxr && 0.0f <= f.sxr[idx];
    if (!c.stereo1 && !stereo2) {
        const double cos_dist = __dadd_rn(__dadd_rn(__dmul_rn(E[9], b2x), __dmul_rn(E[10], b2y)), __dmul_rn(E[11], b2z));
        if (0.99862953475 < cos_dist) {
            return false;
        }
    }
    double e[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        e[r] = __dadd_rn(__dadd_rn(__dmul_rn(E[3 * r], b2x), __dmul_rn(E[3 * r + 1], b2y)), __dmul_rn(E[3 * r + 2], b2z));
    }
    const double dot = __dadd_rn(__dadd_rn(__dmul_rn(e[0], c.b1x), __dmul_rn(e[1], c.b1y)), __dmul_rn(e[2], c.b1z));
    const double norm = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(e[0], e[0]), __dmul_rn(e[1], e[1])), __dmul_rn(e[2], e[2])));
    const double cr = __ddiv_rn(dot, norm);
    bool inlier = -1.0 <= cr && cr < c.sin_thr;
    if (fabs(__dsub_rn(cr, c.sin_thr)) < 1e-13) {
        if (record) {
            const u32 slot = atomicAdd(&f.tri_amb[0], 1u);
            if (slot < f.tri_amb[1]) {
                f.tri_amb[2 + 2 * slot] = (u32)c.qi;
                f.tri_amb[3 + 2 * slot] = j;
            }
        }
        const u32 n_ovr = f.tri_ovr ? f.tri_ovr[0] : 0u;
        for (u32 k = 0; k < n_ovr; ++k) {
            if (f.tri_ovr[1 + 2 * k] == (u32)c.qi && f.tri_ovr[2 + 2 * k] == j) {
                inlier = !inlier;
                break;
            }
        }
    }
    return inlier;
}

// `min_key`: candidates whose selection key is smaller are left out (the continuation of a list whose
// entries up to min_key - 1 have been used up).  `posted` / `qid` (the workgroup-wide resolve kernel): entries
// on which a query with an index below qid has posted are left out as well.
template <int G, int K, int MODE>
__device__ void window_topk(const frame_dev& f, const query_ctx& c, const u32* qd, const uint8_t* occupied,
                            const u32* blocked, u32 (&top)[K], u32& total, u32 min_key = 0u, const u32* posted = nullptr, u32 qid = 0u) {
    const int gl = threadIdx.x & (G - 1);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        top[k] = kNoKey32;
    }
    total = 0;
    window w;
    w.min_cx = w.max_cx = w.min_cy = w.max_cy = 0;
    if (!c.ranged) {
        w = window_of(f.grid, c.rx, c.ry, c.margin);
        if (w.empty) {
            return;
        }
    }
    const int R = (int)f.grid.num_rows;
    const int ncx = c.ranged ? 1 : w.max_cx - w.min_cx + 1;
    u32 loc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        loc[k] = kNoKey32;
    }
    u32 cnt = 0;
    for (int cb = 0; cb < ncx; cb += G) {
        u32 s = 0, e = 0;
        if (c.ranged) {
            if (gl == 0) {
                s = c.lo;
                e = c.hi;
            }
        }
        else if (cb + gl < ncx) {
            const int cx = w.min_cx + cb + gl;
            s = f.cell_start[cx * R + w.min_cy];
            e = f.cell_start[cx * R + w.max_cy + 1];
        }
        const u32 len = e - s;
        // inclusive prefix over the group's lanes and its total, without LDS round trips: a whole wave
        // uses the wave scan, a group of up to 16 lanes row shifts masked at the group boundary
        static_assert(G == 64 || G <= 16, "a query's lane group is the wave or lies inside one DPP row");
        u32 inc = len, T;
        if (G == 64) {
            inc = wave_incl_scan_u32(len);
            T = (u32)__builtin_amdgcn_readlane((int)inc, 63);
        }
        else {
            u32 t = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, true); // row_shr:1
            inc += gl >= 1 ? t : 0u;
            if (G > 2) {
                t = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, true); // row_shr:2
                inc += gl >= 2 ? t : 0u;
            }
            if (G > 4) {
                t = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, true); // row_shr:4
                inc += gl >= 4 ? t : 0u;
            }
            if (G > 8) {
                t = (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, true); // row_shr:8
                inc += gl >= 8 ? t : 0u;
            }
            T = group_sum_u32<G>(len);
        }
        const u32 exc = inc - len;
        for (u32 t0 = 0; t0 < T; t0 += G) {
            const u32 t = t0 + gl;
            int lo = 0; // first lane whose inclusive prefix exceeds t = the column holding candidate t
#pragma unroll
            for (int step = G / 2; step >= 1; step >>= 1) {
                const u32 v = __shfl(inc, lo + step - 1, G);
                if (t >= v) {
                    lo += step;
                }
            }
            const u32 cs = __shfl(s, lo, G), cexc = __shfl(exc, lo, G);
            if (t < T) {
                const u32 j = cs + (t - cexc);
                const uint4 en = f.g_entry[j];
                // (the descriptor is fetched with the entry, not after the gates: the scan is a chain of
                // dependent loads and this takes one link out of every round; a candidate that fails the
                // gates costs 32 bytes that the frame's other queries keep in the L2 anyway)
                const uint4 d0 = f.g_desc[2 * j], d1 = f.g_desc[2 * j + 1];
                if (passes_gates<MODE>(f, c, en, occupied, blocked, true)) {
                    const u32 dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w)
                                     + __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
                    bool keep = true;
                    if (MODE == kModeTri) {
                        // robust.cc:102 first, then the gates in double
                        keep = dist <= kThrLow && tri_gate(f, c, en.w, j, blocked == nullptr);
                    }
                    else if (MODE == kModeArea) {
                        // area.cc:46-49: a keypoint an earlier query holds at this distance or closer is skipped
                        keep = !blocked || (blocked[en.w] >> 20) > dist;
                    }
                    u32 key = (dist << 20) | (MODE == kModeTri ? 0xFFFFFu - j : j);
                    keep = keep && key >= min_key && !(posted && posted[j] < qid);
                    if (keep) {
                        ++cnt;
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const u32 small = min(loc[k], key);
                            key = max(loc[k], key);
                            loc[k] = small;
                        }
                    }
                }
            }
        }
    }
    total = group_sum_u32<G>(cnt);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const u32 m = group_min_u32<G>(loc[0]);
        top[k] = m;
        if (loc[0] == m && m != kNoKey32) {
#pragma unroll
            for (int t = 0; t + 1 < K; ++t) {
                loc[t] = loc[t + 1];
            }
            loc[K - 1] = kNoKey32;
        }
    }
}

// GROUP lanes share a query: kGroup for batches (the device is full of queries), kGroupLive for a call with
// a few problems that is waited for (a frame's 2000 queries fill 70 workgroups with eight lanes each; with
// more lanes per query the scan of a window is shorter and more compute units take part).
template <int MODE, int GROUP = kGroup>
__global__ __launch_bounds__(256) void k_proj_candidates(const proj_problem* __restrict__ problems, proj_params pp) {
    const proj_problem& P = problems[blockIdx.y];
    constexpr int QPB = 256 / GROUP;
    const int qi = blockIdx.x * QPB + (threadIdx.x / GROUP);
    const int gl = threadIdx.x & (GROUP - 1);
    if (qi >= P.q.n) {
        return;
    }
    u32 top[kTopK];
    u32 total = 0;
#pragma unroll
    for (int k = 0; k < kTopK; ++k) {
        top[k] = kNoKey32;
    }
    query_ctx c;
    if (P.q.valid[qi]) {
        if (make_query_ctx<MODE>(pp, P.frm, P.q, qi, c)) {
            u32 qd[8];
            load_desc(P.q.desc + 32 * (size_t)qi, qd);
            window_topk<GROUP, kTopK, MODE>(P.frm, c, qd, P.occupied, nullptr, top, total);
        }
        else if (gl == 0) {
            *P.err = OVS_ERR_OUT_OF_RANGE;
        }
    }
#pragma unroll
    for (int part = 0; part < (kTopK + GROUP - 1) / GROUP; ++part) {
        const int slot = part * GROUP + gl;
        if (slot < kTopK) {
            u32 v = top[0];
#pragma unroll
            for (int k = 1; k < kTopK; ++k) {
                v = slot == k ? top[k] : v;
            }
            P.lists[(size_t)qi * kTopK + slot] = list_key<MODE>(P.frm, v);
        }
    }
    if (gl == 0) {
        P.totals[qi] = total;
    }
}

// ---- resolution ---------------------------------------------------------------------------------
__device__ __forceinline__ bool accept_rule(const proj_params& pp, const frame_dev& f, u64 best, u64 second) {
    const u32 bd = key_dist(best);
    switch (pp.kind) {
        case kFrameLms: {
            if (bd > kThrHigh) {
                return false;
            }
            // Lowe's ratio only between candidates of the same octave, projection.cc:69-73
            const int bl = f.kps[key_idx(best)].octave;
            const int slv = second == kNoKey ? -1 : f.kps[key_idx(second)].octave;
            const u32 sd = second == kNoKey ? kMaxDist : key_dist(second);
            return !(bl == slv && (float)bd > __fmul_rn(pp.lowe, (float)sd));
        }
        case kRange:
        case kArea: {
            // robust.cc:270-281, bow_tree.cc:80-87, area.cc:61-68
            if (bd > kThrLow) {
                return false;
            }
            const u32 sd = second == kNoKey ? kMaxDist : key_dist(second);
            return !(__fmul_rn(pp.lowe, (float)sd) < (float)bd);
        }
        case kFrameKeyfrm:
            return bd <= pp.thr;
        case kSim3:
        case kFuseDetect:
        case kFuseReplace:
        case kTri:
            return bd <= kThrLow;
        default:
            return bd <= kThrHigh; // kCurrLast, kOneWay
    }
}

// One wavefront per problem.
//
// The reference walks the queries in order; each takes the best candidate no earlier query has
// claimed.  Here 64 consecutive queries are evaluated at once against the current claim bitmap;
// every accepting lane posts its lane id on its pick with an LDS atomicMin, and a lane whose pick
// (or, for the ratio test, whose runner-up) carries a smaller lane id has been overtaken by an
// earlier query of the same round.  All lanes before the first overtaken one commit together, the
// rest are re-evaluated in the next round, so the result is the serial one while a round usually
// retires most of the chunk.  A query whose list ran dry while an unlisted candidate could still
// change its outcome is handled alone with a full rescan of its window.
// IN_LDS: the per-keypoint `matched` and `claim` arrays fit in LDS next to the bitmap; otherwise
// they stay in global memory and queries are retired one per round.
// The resolve kernel is one wavefront per workgroup.  The LDS unit executes the instructions of one
// wave in the order they were issued, so a read that follows a write of another lane in program
// order sees it: a point where all lanes must see each other's LDS writes only has to keep the
// compiler from moving accesses across it.  No s_barrier (whose preceding full wait would also
// stall on the global loads that are deliberately left in flight) and no counter drain.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// KIND >= 0 compiles the kernel for that one matcher (the kind tests fold away, which matters here:
// the kernel is a single wavefront whose time is its instruction count); -1 = any kind.
template <bool IN_LDS, int MODE, int KIND = -1>
__global__ __launch_bounds__(64) void k_proj_resolve(const proj_problem* __restrict__ problems, proj_params pp_arg, int lds_n) {
    proj_params pp = pp_arg;
    if (KIND >= 0) {
        pp.kind = KIND;
    }
    extern __shared__ u32 sm[];
    __shared__ u32 hist[32];
    __shared__ u32 order[32];
    __shared__ ovs_sort::sort_frame sort_stack[ovs_sort::kSortStackEntries];
    const proj_problem& P = problems[blockIdx.x];
    const int lane = threadIdx.x;
    const int n = P.frm.n, nq = P.q.n;
    const int words = (n + 31) >> 5;
    u32* blocked = sm;
    int* lmatched = reinterpret_cast<int*>(sm + ((lds_n + 31) >> 5) + 1);
    u32* claim = reinterpret_cast<u32*>(lmatched + lds_n);
    for (int i = lane; i < words; i += 64) {
        blocked[i] = 0;
    }
    if (lane < 32) {
        hist[lane] = 0;
    }
    for (int i = lane; i < n; i += 64) {
        if (IN_LDS) {
            lmatched[i] = -1;
            claim[i] = 0x7FFFFFFFu;
        }
        else {
            P.matched[i] = -1;
        }
    }
    wave_sync();

    constexpr bool AREA = MODE == kModeArea;
    // kArea: `state` holds, per keypoint of the searched frame, distance << 20 | query of the match that
    // currently owns it (all ones = free); it takes the place of the claim bitmap.  Frames too large for
    // the LDS plan keep it in the global `matched` array (initialised to -1 above, i.e. free) and resolve
    // one query per round.
    u32* state = IN_LDS ? reinterpret_cast<u32*>(lmatched) : reinterpret_cast<u32*>(P.matched);
    const bool excl = !kind_is_one_way(pp.kind);
    const int needed = (pp.kind == kFrameLms || pp.kind == kRange || pp.kind == kArea) ? 2 : 1;
    const u32 athr = pp.kind == kFrameKeyfrm ? pp.thr
                                             : ((pp.kind == kSim3 || pp.kind == kRange || pp.kind == kFuseDetect || pp.kind == kFuseReplace
                                                 || pp.kind == kArea || pp.kind == kTri)
                                                    ? kThrLow
                                                    : kThrHigh);
    u32 mynum = 0;
    // the candidate list, count and flags of a chunk's query are fetched while the previous chunk is
    // being resolved
    u64 nkeys[kTopK];
    u32 ntot = 0;
    bool ndone = true, nblocks = true;
    auto fetch = [&](int q) {
#pragma unroll
        for (int k = 0; k < kTopK; ++k) {
            nkeys[k] = kNoKey;
        }
        ntot = 0;
        ndone = true;
        nblocks = true;
        if (q < nq) {
            ntot = P.totals[q];
            ndone = !P.q.valid[q] || ntot == 0;
            const ulonglong2* lp = reinterpret_cast<const ulonglong2*>(P.lists + (size_t)q * kTopK);
#pragma unroll
            for (int k = 0; k < kTopK / 2; ++k) {
                const ulonglong2 v = lp[k];
                nkeys[2 * k] = v.x;
                nkeys[2 * k + 1] = v.y;
            }
            nblocks = (pp.kind == kFrameKeyfrm || pp.kind == kSim3 || pp.kind == kRange) ? true : (P.q.has_obs ? P.q.has_obs[q] != 0 : true);
        }
    };
    fetch(lane);
    for (int base = 0; base < nq; base += 64) {
        const int mq = base + lane;
        u64 mykeys[kTopK];
#pragma unroll
        for (int k = 0; k < kTopK; ++k) {
            mykeys[k] = nkeys[k];
        }
        const u32 mytot = ntot;
        bool done = ndone;
        const bool blocks = nblocks;
        fetch(mq + 64);
        int myacc = -1;
        while (true) {
            u64 best = kNoKey, second = kNoKey;
            int found = 0;
            if (!done) {
#pragma unroll
                for (int k = 0; k < kTopK; ++k) {
                    const u64 key = mykeys[k];
                    if (key != kNoKey && found < needed) { // the runner-up only matters to the ratio tests
                        const u32 idx = key_idx(key);
                        const bool b = AREA ? (state[idx] >> 20) <= key_dist(key) : (excl && ((blocked[idx >> 5] >> (idx & 31)) & 1u));
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
            }
            // A rescan can only matter if an unlisted candidate (whose distance is at least the last
            // listed one) could still be accepted or could change the ratio test.
            bool need_rescan = false;
            bool second_irrelevant = false;
            if (!done && found < needed && mytot > (u32)kTopK) {
                const u32 last_d = key_dist(mykeys[kTopK - 1]);
                if (found == 0) {
                    need_rescan = last_d <= athr;
                }
                else {
                    const u32 bd = key_dist(best);
                    if (bd <= athr) {
                        need_rescan = (float)bd > __fmul_rn(pp.lowe, (float)last_d);
                        second_irrelevant = !need_rescan;
                    }
                }
            }
            const bool live = !done && !need_rescan && best != kNoKey;
            const bool accept = live && (second_irrelevant ? true : accept_rule(pp, P.frm, best, second));
            const u32 bidx = key_idx(best);
            const u64 rescan_mask = __ballot(need_rescan);
            const int first_rescan = rescan_mask ? __ffsll((long long)rescan_mask) - 1 : 64;
            int stop;
            if (!excl) {
                stop = 64; // no interaction between queries
            }
            else if (IN_LDS) {
                const bool part = accept && lane < first_rescan;
                if (part) {
                    atomicMin(&claim[bidx], (u32)lane);
                }
                wave_sync();
                bool conflict = false;
                if (live && lane < first_rescan) {
                    conflict = claim[bidx] < (u32)lane;
                    if (needed == 2 && second != kNoKey) {
                        conflict |= claim[key_idx(second)] < (u32)lane;
                    }
                }
                const u64 cmask = __ballot(conflict);
                const int first_conflict = cmask ? __ffsll((long long)cmask) - 1 : 64;
                stop = min(first_rescan, first_conflict);
                wave_sync();
                if (part) {
                    claim[bidx] = 0x7FFFFFFFu;
                }
            }
            else {
                const u64 undone = __ballot(!done);
                const int first = undone ? __ffsll((long long)undone) - 1 : 64;
                stop = min(first_rescan, first == 64 ? 64 : first + 1);
            }
            if (AREA) {
                // area.cc:75-84: the keypoint changes hands; the previous owner's match is void from now
                // on, which the final pass sees from the owner field
                if (accept && lane < stop) {
                    state[bidx] = (key_dist(best) << 20) | (u32)mq;
                    myacc = (int)bidx;
                }
            }
            else if (accept && lane < stop) {
                if (excl) {
                    if (IN_LDS) {
                        lmatched[bidx] = mq;
                    }
                    else {
                        P.matched[bidx] = mq;
                    }
                    if (blocks) {
                        atomicOr(&blocked[bidx >> 5], 1u << (bidx & 31));
                    }
                    ++mynum;
                }
                myacc = (int)bidx;
            }
            if (lane < stop) {
                done = true;
            }
            wave_sync();
            if (stop == 64) {
                break;
            }
            if (stop == first_rescan) {
                const int qi = base + stop;
                query_ctx c;
                make_query_ctx<MODE>(pp, P.frm, P.q, qi, c);
                u32 qd[8];
                load_desc(P.q.desc + 32 * (size_t)qi, qd);
                u32 sel2[2];
                u32 t2;
                window_topk<64, 2, MODE>(P.frm, c, qd, P.occupied, AREA ? state : blocked, sel2, t2);
                const u64 top2[2] = {list_key<MODE>(P.frm, sel2[0]), list_key<MODE>(P.frm, sel2[1])};
                if (top2[0] != kNoKey && accept_rule(pp, P.frm, top2[0], top2[1])) {
                    const u32 ridx = key_idx(top2[0]);
                    const bool rblocks = __shfl((int)blocks, stop) != 0;
                    if (AREA) {
                        if (lane == stop) {
                            state[ridx] = (key_dist(top2[0]) << 20) | (u32)qi;
                            myacc = (int)ridx;
                        }
                    }
                    else if (lane == stop) {
                        if (IN_LDS) {
                            lmatched[ridx] = qi;
                        }
                        else {
                            P.matched[ridx] = qi;
                        }
                        if (rblocks) {
                            atomicOr(&blocked[ridx >> 5], 1u << (ridx & 31));
                        }
                        ++mynum;
                        myacc = (int)ridx;
                    }
                }
                if (lane == stop) {
                    done = true;
                }
                wave_sync();
            }
        }
        if (mq < nq) {
            P.acc_idx[mq] = myacc;
        }
    }
    wave_sync();
    u32 num = wave_sum_u32(mynum);

    const bool orient = pp.check_orientation && (pp.kind == kCurrLast || pp.kind == kFrameKeyfrm || pp.kind == kRange || pp.kind == kTri);
    if (AREA) {
        // area.cc:87-102.  The rotation histogram holds an entry for every match that was ever made,
        // including those whose keypoint a later query took over (append_delta_angle is never undone);
        // a match counts in the end if its query still owns the keypoint and, with the rotation check
        // on, its bin is among the three fullest.  Every lane revisits the queries it wrote acc_idx for.
        if (pp.check_orientation) {
            for (int qi = lane; qi < nq; qi += 64) {
                const int idx = P.acc_idx[qi];
                if (0 <= idx) {
                    int bin = angle_bin(__fsub_rn(P.q.angle[qi], P.frm.kps[idx].angle));
                    if (bin < 0 || 30 <= bin) {
                        *P.err = OVS_ERR_OUT_OF_RANGE;
                        bin = 0;
                    }
                    atomicAdd(&hist[bin], 1u);
                }
            }
            wave_sync();
            if (lane == 0) {
                ovs_sort::index_sort_by_size(hist, order, 30, sort_stack);
            }
            wave_sync();
        }
        const u32 v0 = order[0], v1 = order[1], v2 = order[2];
        u32 kept = 0;
        for (int qi = lane; qi < nq; qi += 64) {
            const int idx = P.acc_idx[qi];
            if (0 <= idx) {
                bool ok = (state[idx] & 0xFFFFFu) == (u32)qi;
                if (ok && pp.check_orientation) {
                    int bin = angle_bin(__fsub_rn(P.q.angle[qi], P.frm.kps[idx].angle));
                    bin = (bin < 0 || 30 <= bin) ? 0 : bin;
                    ok = (u32)bin == v0 || (u32)bin == v1 || (u32)bin == v2;
                }
                if (!ok) {
                    P.acc_idx[qi] = -1;
                }
                kept += ok ? 1u : 0u;
            }
        }
        kept = wave_sum_u32(kept);
        if (lane == 0) {
            *P.num_matches = kept;
        }
        return;
    }
    if (orient) {
        // rotation consistency, match/angle_checker.h: keep the three fullest of 30 bins.  Every lane
        // revisits exactly the queries it wrote acc_idx for.
        // Both passes gather (accepted index -> keypoint angle): eight independent queries per lane
        // are in flight at a time, the loads of a group are issued before any of them is used.
        constexpr int U = 8;
        auto bins_of = [&](int q0, int (&idx)[U], int (&bin)[U]) {
            float qa[U], ca[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int qi = q0 + 64 * u;
                idx[u] = qi < nq ? P.acc_idx[qi] : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int qi = q0 + 64 * u;
                qa[u] = 0 <= idx[u] ? P.q.angle[qi] : 0.0f;
                ca[u] = 0 <= idx[u] ? P.frm.kps[idx[u]].angle : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                bin[u] = angle_bin(pp.angle_sign >= 0 ? __fsub_rn(qa[u], ca[u]) : __fsub_rn(ca[u], qa[u]));
            }
        };
        for (int q0 = lane; q0 < nq; q0 += 64 * U) {
            int idx[U], bin[U];
            bins_of(q0, idx, bin);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (0 <= idx[u]) {
                    if (bin[u] < 0 || 30 <= bin[u]) {
                        *P.err = OVS_ERR_OUT_OF_RANGE; // angle_histogram_.at(bin) would throw
                        bin[u] = 0;
                    }
                    atomicAdd(&hist[bin[u]], 1u);
                }
            }
        }
        wave_sync();
        if (lane == 0) {
            ovs_sort::index_sort_by_size(hist, order, 30, sort_stack);
        }
        wave_sync();
        const u32 v0 = order[0], v1 = order[1], v2 = order[2];
        u32 removed = 0;
        for (int q0 = lane; q0 < nq; q0 += 64 * U) {
            int idx[U], bin[U];
            bins_of(q0, idx, bin);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (0 <= idx[u]) {
                    const int b = (bin[u] < 0 || 30 <= bin[u]) ? 0 : bin[u];
                    if ((u32)b != v0 && (u32)b != v1 && (u32)b != v2) {
                        if (pp.per_query_output) {
                            P.acc_idx[q0 + 64 * u] = -1;
                        }
                        else if (IN_LDS) {
                            lmatched[idx[u]] = -2;
                        }
                        else {
                            P.matched[idx[u]] = -2;
                        }
                        ++removed;
                    }
                }
            }
        }
        removed = wave_sum_u32(removed);
        num -= removed;
        wave_sync();
    }
    if (IN_LDS && excl) {
        for (int i = lane; i < n; i += 64) {
            P.matched[i] = lmatched[i];
        }
    }
    if (lane == 0 && excl) {
        *P.num_matches = num;
    }
}

// ---- resolution by a whole workgroup, for the matchers without a ratio test ---------------------------
// projection::match_current_and_last_frames, match_frame_and_keyframe and match_by_Sim3_transform accept a
// query's best free candidate when its distance is below a threshold, and nothing else about the
// candidates matters (projection.cc:161-186, :273-298, :367-405).  Walking the queries in order, each
// taking the best candidate no earlier (blocking) query holds, is then a serial dictatorship, and its
// outcome is the one assignment in which every query holds the first entry of its list that no query with
// a smaller index holds.  That fixed point is reached without walking the queries in order: all queries
// post their index on their current candidate (LDS atomicMin), a query that finds a smaller index there
// moves on to its next candidate, until nobody moves.  The minimum posted on a keypoint only ever
// decreases, so a rejection is final, and a round costs two barriers however many queries take part
// (the single-wave kernel retires the queries up to the first conflict per round: about twelve).
// Queries of landmarks without observations take a keypoint without holding it (projection.cc:164): they
// are rejected like the others but post nothing; the keypoint's slot ends up with the last taker.
// A query whose eight listed candidates are all rejected while its window holds more gets the next eight
// from a rescan of the window by one wave (selection keys above the last listed one).
constexpr int kWideThreads = 1024;
constexpr int kWideQueries = 4;    // queries per thread
constexpr int kWideQueue = 256;    // list continuations per batch
constexpr int kWideMaxRounds = 1 << 20;
// LDS of the kernel: per entry of the searched frame's grid (an entry is a keypoint) the smallest query index
// posted on it, its keypoint index and its angle; per query its candidate list as eight words
// `distance << 20 | entry` -- the selection keys of window_topk: they order the candidates as the reference's
// scan does, name the keypoint and are the lower bound of a list's continuation.  The rounds touch LDS only.
__host__ __device__ inline size_t wide_lds_bytes(int n, int nq) { return (3 * (size_t)max(n, 1) + (size_t)max(nq, 1) * kTopK) * 4; }

// The rotation bins that angle_checker keeps (angle_checker.h:126-134: the first three of the bin indices
// sorted by size, descending, with std::sort) as a bit mask, computed by the first wave of the workgroup.
// A non-empty bin with three larger bins is out, one with at most three bins of its size or larger is in,
// whatever the sort does with equal sizes; only when equal sizes straddle the third place does the order
// std::sort leaves them in decide, and then the port of that sort runs.
__device__ __forceinline__ u32 kept_rotation_bins(const u32* hist, u32* order, ovs_sort::sort_frame* sort_stack) {
    const int lane = threadIdx.x & 63;
    const u32 s = lane < 30 ? hist[lane] : 0u;
    u32 greater = 0, at_least = 0;
    for (int b = 0; b < 30; ++b) {
        const u32 v = hist[b];
        greater += v > s;
        at_least += v >= s;
    }
    const bool ambiguous = lane < 30 && s > 0 && greater < 3 && at_least > 3;
    if (__ballot(ambiguous)) {
        if (lane == 0) {
            ovs_sort::index_sort_by_size(hist, order, 30, sort_stack);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return (1u << order[0]) | (1u << order[1]) | (1u << order[2]);
    }
    return (u32)__ballot(lane < 30 && s > 0 && greater < 3);
}

template <int KIND>
__global__ __launch_bounds__(kWideThreads) void k_proj_resolve_wide(const proj_problem* __restrict__ problems, proj_params pp_arg, int lds_n) {
    proj_params pp = pp_arg;
    if (KIND >= 0) {
        pp.kind = KIND;
    }
    extern __shared__ u32 wide_sm[];
    __shared__ u32 hist[32];
    __shared__ u32 order[32];
    __shared__ ovs_sort::sort_frame sort_stack[ovs_sort::kSortStackEntries];
    __shared__ u32 queue[kWideQueue];
    __shared__ u32 n_queue;
    __shared__ u32 moved_flag[3];
    __shared__ u32 keep_mask;
    __shared__ u32 red[kWideThreads / 64];
    const proj_problem& P = problems[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.frm.n, nq = P.q.n;
    const bool orient = pp.check_orientation && (pp.kind == kCurrLast || pp.kind == kFrameKeyfrm);
    u32* holder = wide_sm;                                        // lds_n words
    u32* entry_idx = wide_sm + lds_n;                             // keypoint index of an entry
    float* entry_angle = reinterpret_cast<float*>(wide_sm + 2 * lds_n);
    u32* rows = wide_sm + 3 * lds_n;                              // nq x kTopK
    const u32 athr = pp.kind == kFrameKeyfrm ? pp.thr : (pp.kind == kSim3 ? kThrLow : kThrHigh);
    const int n_entries = (int)P.frm.cell_start[P.frm.grid.num_cols * P.frm.grid.num_rows];
    for (int i = tid; i < n; i += kWideThreads) {
        holder[i] = 0xFFFFFFFFu;
        P.matched[i] = -1;
        if (i < n_entries) {
            const u32 idx = P.frm.g_entry[i].w;
            entry_idx[i] = idx;
            entry_angle[i] = orient ? P.frm.kps[idx].angle : 0.0f;
        }
    }
    // the lists as selection keys: list key = distance << 40 | entry << 20 | keypoint
    for (int i = tid; i < nq * kTopK; i += kWideThreads) {
        const u64 key = P.lists[i];
        rows[i] = key == kNoKey ? kNoKey32 : (u32)(key >> 20);
    }
    if (tid < 32) {
        hist[tid] = 0;
    }
    if (tid < 3) {
        moved_flag[tid] = 0;
    }
    if (tid == 0) {
        n_queue = 0;
    }
    // state of this thread's queries: the current candidate (kNoKey32: none left), its position in the list,
    // the number of candidates from the start of the current list on
    u32 cur[kWideQueries], rem[kWideQueries];
    int ptr[kWideQueries];
    float qangle[kWideQueries];
    bool blocks[kWideQueries], live[kWideQueries];
    int waiting[kWideQueries]; // 0: no, 1: wants its list continued, 2: queued for it
#pragma unroll
    for (int k = 0; k < kWideQueries; ++k) {
        const int q = tid + k * kWideThreads;
        cur[k] = kNoKey32;
        rem[k] = 0;
        ptr[k] = 0;
        qangle[k] = 0.0f;
        blocks[k] = false;
        waiting[k] = 0;
        live[k] = false;
        if (q < nq && P.q.valid[q]) {
            rem[k] = P.totals[q];
            live[k] = rem[k] != 0;
            blocks[k] = (pp.kind == kFrameKeyfrm || pp.kind == kSim3) ? true : (P.q.has_obs ? P.q.has_obs[q] != 0 : true);
            qangle[k] = orient ? P.q.angle[q] : 0.0f;
        }
    }
    __syncthreads();
    auto usable = [&](u32 key) { return (key != kNoKey32 && sel_dist(key) <= athr) ? key : kNoKey32; };
#pragma unroll
    for (int k = 0; k < kWideQueries; ++k) {
        const u32 q = (u32)(tid + k * kWideThreads);
        if (live[k]) {
            cur[k] = usable(rows[(size_t)q * kTopK]);
            if (cur[k] != kNoKey32 && blocks[k]) {
                atomicMin(&holder[sel_entry(cur[k])], q);
            }
        }
    }
    __syncthreads();

    // A round: every query looks at the word of its candidate; one that finds a smaller index there moves on
    // (skipping candidates that already carry a smaller index) and posts on its new candidate at once.  A post
    // made during a round may or may not be seen by the checks of that round; it is seen by the next, and a
    // round in which nobody moved made no post, so its checks saw the final words.
    // A query whose list is used up while its window holds more waits, holding nothing, until the other
    // queries have settled; then all waiting queries get the next candidates of their windows at once, one wave
    // per query (a rescan costs about 7 us of dependent loads, whether for one query or for sixteen), and the
    // rounds go on.  The order of these events does not matter: words only decrease, rejections are final.
    for (int round = 0;; ++round) {
        int moved = 0;
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            const u32 q = (u32)(tid + k * kWideThreads);
            bool stepped = false;
            while (waiting[k] == 0 && cur[k] != kNoKey32 && holder[sel_entry(cur[k])] < q) {
                stepped = true;
                ++ptr[k];
                if (ptr[k] == kTopK) {
                    // the list is used up: the window holds more only if it counted more than the list's length
                    if (rem[k] > (u32)kTopK) {
                        waiting[k] = 1;
                    }
                    cur[k] = kNoKey32;
                    break;
                }
                cur[k] = usable(rows[(size_t)q * kTopK + ptr[k]]);
            }
            if (stepped && cur[k] != kNoKey32 && blocks[k]) {
                atomicMin(&holder[sel_entry(cur[k])], q);
            }
            moved |= stepped ? 1 : 0;
            if (waiting[k] == 1) {
                const u32 slot = atomicAdd(&n_queue, 1u);
                if (slot < (u32)kWideQueue) {
                    queue[slot] = q;
                    waiting[k] = 2;
                } // else: after the next batch
            }
        }
        if (moved) {
            moved_flag[round % 3] = 1u;
        }
        if (tid == 0) {
            moved_flag[(round + 1) % 3] = 0u; // last read two rounds ago
        }
        __syncthreads();
        if (moved_flag[round % 3] != 0u) {
            if (round >= kWideMaxRounds) {
                if (tid == 0) {
                    *P.err = OVS_ERR_OUT_OF_RANGE; // cannot happen: every round with a move uses up a candidate
                }
                break;
            }
            continue;
        }
        // everybody who is not waiting has settled
        const int asked = (int)n_queue;
        if (asked == 0) {
            break;
        }
        const int queued = min(asked, kWideQueue);
        for (int e = wave; e < queued; e += kWideThreads / 64) {
            const int q = (int)queue[e];
            const u32 last = rows[(size_t)q * kTopK + kTopK - 1];
            query_ctx c;
            make_query_ctx<kModePlain>(pp, P.frm, P.q, q, c);
            u32 qd[8];
            load_desc(P.q.desc + 32 * (size_t)q, qd);
            u32 sel[kTopK];
            u32 t;
            // behind the last candidate the query has seen, without the entries that already carry a smaller index
            window_topk<64, kTopK, kModePlain>(P.frm, c, qd, P.occupied, nullptr, sel, t, last + 1u, holder, (u32)q);
            if (lane < kTopK) {
                u32 v = sel[0];
#pragma unroll
                for (int k = 1; k < kTopK; ++k) {
                    v = lane == k ? sel[k] : v;
                }
                rows[(size_t)q * kTopK + lane] = v;
            }
        }
        __syncthreads();
        if (tid == 0) {
            n_queue = 0;
        }
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            const u32 q = (u32)(tid + k * kWideThreads);
            if (waiting[k] == 2) {
                waiting[k] = 0;
                rem[k] -= (u32)kTopK;
                ptr[k] = 0;
                cur[k] = usable(rows[(size_t)q * kTopK]);
                if (cur[k] != kNoKey32 && blocks[k]) {
                    atomicMin(&holder[sel_entry(cur[k])], q);
                }
            }
        }
        __syncthreads(); // the cleared counter and the new posts, before the next round
    }

    // every query now holds its final candidate
    u32 mynum = 0;
    int acc[kWideQueries], bin[kWideQueries];
#pragma unroll
    for (int k = 0; k < kWideQueries; ++k) {
        const int q = tid + k * kWideThreads;
        acc[k] = -1;
        bin[k] = 0;
        if (cur[k] != kNoKey32) {
            const u32 e = sel_entry(cur[k]);
            acc[k] = (int)entry_idx[e];
            if ((u32)acc[k] >= (u32)n) {
                *P.err = OVS_ERR_OUT_OF_RANGE; // cannot happen with lists of this frame: never index past it
                acc[k] = -1;
                continue;
            }
            atomicMax(&P.matched[acc[k]], q); // the last taker of a keypoint that nobody held
            ++mynum;
            if (orient) {
                const float ca = entry_angle[e];
                bin[k] = angle_bin(pp.angle_sign >= 0 ? __fsub_rn(qangle[k], ca) : __fsub_rn(ca, qangle[k]));
                if (bin[k] < 0 || 30 <= bin[k]) {
                    *P.err = OVS_ERR_OUT_OF_RANGE; // angle_histogram_.at(bin) would throw
                    bin[k] = 0;
                }
                atomicAdd(&hist[bin[k]], 1u);
            }
        }
    }
    u32 removed = 0;
    if (orient) {
        // rotation consistency, match/angle_checker.h: keep the three fullest of 30 bins
        __syncthreads();
        if (wave == 0) {
            const u32 mask = kept_rotation_bins(hist, order, sort_stack);
            if (lane == 0) {
                keep_mask = mask;
            }
        }
        __syncthreads();
        const u32 keep = keep_mask;
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            if (0 <= acc[k] && !((keep >> bin[k]) & 1u)) {
                if (pp.per_query_output) {
                    acc[k] = -1;
                }
                else {
                    P.matched[acc[k]] = -2; // after every taker has posted (barriers above)
                }
                ++removed;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kWideQueries; ++k) {
        const int q = tid + k * kWideThreads;
        if (q < nq) {
            P.acc_idx[q] = acc[k];
        }
    }
    const u32 kept = wave_sum_u32(mynum - removed);
    if (lane == 0) {
        red[wave] = kept;
    }
    __syncthreads();
    if (tid == 0) {
        u32 num = 0;
        for (int w = 0; w < kWideThreads / 64; ++w) {
            num += red[w];
        }
        *P.num_matches = num;
    }
}

// ---- resolution by a whole workgroup, for the matchers with a ratio test --------------------------------
// projection::match_frame_and_landmarks (projection.cc:11-82) and robust::brute_force_match / bow_tree::match_*
// (robust.cc:250-285, bow_tree.cc:60-95) accept a query's best free candidate only if it also passes a test
// against the second best free one, and a query that fails it holds nothing.  The outcome of the serial
// walk is still the one assignment in which every query's decision is the reference's decision function
// applied to the keypoints that queries with smaller indices hold; but a decision can flip both ways while
// that set grows (same-octave rule of projection.cc:69-73), so the words "smallest index posted" are rebuilt
// from the current decisions in every round instead of only decreasing: every query decides from the words
// of the previous round, and a round that changes no decision is the serial result (the decision of query q
// is final after q rounds at the latest; in practice after as many rounds as the longest chain of queries
// that push each other on, about ten).  Lists are kept whole: a query whose list does not hold two free
// candidates while its window holds more gets eight more behind it (up to kJacobiRowsMax rows, one wave per
// query, all waiting queries at once when the others have settled).  Pool or rows exhausted: the kernel
// reports kResolveOverflow and the host runs the single-wave kernel on the same lists.
constexpr int kJacobiPool = 256;    // continuation rows in LDS
constexpr int kJacobiRowsMax = 3;   // rows of eight candidates per query (the first included)
constexpr int kResolveOverflow = 1000;
__host__ __device__ inline size_t jacobi_lds_bytes(int n, int nq) { return (2 * (size_t)max(n, 1) + (size_t)max(nq, 1) * kTopK) * 4; }

template <int KIND>
__global__ __launch_bounds__(kWideThreads) void k_proj_resolve_jacobi(const proj_problem* __restrict__ problems, proj_params pp_arg, int lds_n) {
    proj_params pp = pp_arg;
    pp.kind = KIND;
    extern __shared__ u32 wide_sm[];
    __shared__ u32 hist[32];
    __shared__ u32 order[32];
    __shared__ ovs_sort::sort_frame sort_stack[ovs_sort::kSortStackEntries];
    __shared__ u32 pool[kJacobiPool * kTopK];
    __shared__ u32 queue[kJacobiPool];      // query | row slot << 20 of a waiting query
    __shared__ u32 n_queue, n_pool, overflow;
    __shared__ u32 moved_flag[3];
    __shared__ u32 keep_mask;
    __shared__ u32 red[kWideThreads / 64];
    const proj_problem& P = problems[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.frm.n, nq = P.q.n;
    const bool orient = pp.check_orientation && KIND == kRange;
    u32* holder = wide_sm;              // lds_n words: smallest index of the blocking queries that hold the entry
    u32* entry_info = wide_sm + lds_n;  // keypoint index | octave << 24
    u32* rows = wide_sm + 2 * lds_n;    // nq x kTopK selection keys
    const u32 athr = KIND == kFrameLms ? kThrHigh : kThrLow;
    const int n_entries = P.frm.n_entries_plus1 ? P.frm.n_entries_plus1 - 1 : (int)P.frm.cell_start[P.frm.grid.num_cols * P.frm.grid.num_rows];
    for (int i = tid; i < n; i += kWideThreads) {
        P.matched[i] = -1;
    }
    for (int i = tid; i < min(n_entries, lds_n); i += kWideThreads) {
        const uint4 en = P.frm.g_entry[i];
        entry_info[i] = (en.w & 0xFFFFFFu) | (en.z << 24);
    }
    for (int i = tid; i < nq * kTopK; i += kWideThreads) {
        const u64 key = P.lists[i];
        rows[i] = key == kNoKey ? kNoKey32 : (u32)(key >> 20);
    }
    if (tid < 32) {
        hist[tid] = 0;
    }
    if (tid < 3) {
        moved_flag[tid] = 0;
    }
    if (tid == 0) {
        n_queue = 0;
        n_pool = 0;
        overflow = 0;
    }
    u32 decision[kWideQueries], total[kWideQueries], last_listed[kWideQueries];
    int ext[kWideQueries][kJacobiRowsMax - 1], nrows[kWideQueries];
    bool blocks[kWideQueries], live[kWideQueries];
    int waiting[kWideQueries]; // 0: no, 1: wants more candidates, 2: queued for them
#pragma unroll
    for (int k = 0; k < kWideQueries; ++k) {
        const int q = tid + k * kWideThreads;
        decision[k] = kNoKey32;
        total[k] = 0;
        last_listed[k] = 0;
        nrows[k] = 1;
#pragma unroll
        for (int r = 0; r < kJacobiRowsMax - 1; ++r) {
            ext[k][r] = -1;
        }
        blocks[k] = false;
        live[k] = false;
        waiting[k] = 0;
        if (q < nq && P.q.valid[q]) {
            total[k] = P.totals[q];
            live[k] = total[k] != 0;
            blocks[k] = KIND == kRange ? true : (P.q.has_obs ? P.q.has_obs[q] != 0 : true);
        }
    }
    auto row_ptr = [&](int k, int q, int r) -> const u32* { return r == 0 ? rows + (size_t)q * kTopK : pool + ext[k][r - 1] * kTopK; };
    auto octave_of = [&](u32 key) { return (int)(entry_info[sel_entry(key)] >> 24); };
    // the reference's decision for a query from its best and second best free candidates (kNoKey32: none)
    auto accepts = [&](u32 best, u32 second) {
        const u32 bd = sel_dist(best);
        if (bd > athr) {
            return false;
        }
        const u32 sd = second == kNoKey32 ? kMaxDist : sel_dist(second);
        if (KIND == kFrameLms) {
            // Lowe's ratio only between candidates of the same octave, projection.cc:69-73
            const int bl = octave_of(best), slv = second == kNoKey32 ? -1 : octave_of(second);
            return !(bl == slv && (float)bd > __fmul_rn(pp.lowe, (float)sd));
        }
        return !(__fmul_rn(pp.lowe, (float)sd) < (float)bd); // robust.cc:270-281, bow_tree.cc:80-87
    };
    __syncthreads();

    for (int round = 0;; ++round) {
        // the words of this round from the decisions of the last one
        for (int i = tid; i < lds_n; i += kWideThreads) {
            holder[i] = 0xFFFFFFFFu;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            if (decision[k] != kNoKey32 && blocks[k]) {
                atomicMin(&holder[sel_entry(decision[k])], (u32)(tid + k * kWideThreads));
            }
        }
        __syncthreads();
        int moved = 0;
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            const u32 q = (u32)(tid + k * kWideThreads);
            if (!live[k] || waiting[k] != 0) {
                continue;
            }
            u32 best = kNoKey32, second = kNoKey32, last = kNoKey32;
            int found = 0;
            for (int r = 0; r < nrows[k]; ++r) {
                const u32* row = row_ptr(k, (int)q, r);
#pragma unroll
                for (int c = 0; c < kTopK; ++c) {
                    const u32 key = row[c];
                    if (key != kNoKey32) {
                        last = key;
                        if (found < 2 && !(holder[sel_entry(key)] < q)) {
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
            }
            // candidates that are not listed lie behind `last`: they matter only while one of them could
            // still be accepted or could still fail the ratio test (as in the single-wave kernel)
            bool more = false;
            if (found < 2 && total[k] > (u32)(nrows[k] * kTopK) && last != kNoKey32) {
                const u32 last_d = sel_dist(last);
                more = found == 0 ? last_d <= athr : (sel_dist(best) <= athr && (float)sel_dist(best) > __fmul_rn(pp.lowe, (float)last_d));
            }
            u32 d = kNoKey32;
            if (more) {
                waiting[k] = 1;
                last_listed[k] = last;
            }
            else if (found > 0 && accepts(best, second)) {
                d = best;
            }
            moved |= d != decision[k] ? 1 : 0;
            decision[k] = d;
        }
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            if (waiting[k] == 1) {
                const u32 q = (u32)(tid + k * kWideThreads);
                const u32 slot = atomicAdd(&n_pool, 1u);
                if (nrows[k] >= kJacobiRowsMax || slot >= (u32)kJacobiPool) {
                    overflow = 1u;
                }
                else {
                    const u32 at = atomicAdd(&n_queue, 1u);
                    queue[at] = q | (slot << 20);
                    pool[slot * kTopK] = last_listed[k]; // for the wave that fills the row: where the list ends so far
                    ext[k][nrows[k] - 1] = (int)slot;
                    waiting[k] = 2;
                }
            }
        }
        if (moved) {
            moved_flag[round % 3] = 1u;
        }
        if (tid == 0) {
            moved_flag[(round + 1) % 3] = 0u; // last read two rounds ago
        }
        __syncthreads();
        if (overflow) {
            if (tid == 0) {
                *P.err = kResolveOverflow;
            }
            return;
        }
        if (moved_flag[round % 3] != 0u) {
            if (round >= kWideMaxRounds) {
                if (tid == 0) {
                    *P.err = kResolveOverflow; // cannot happen: query q is final after q rounds
                }
                return;
            }
            continue;
        }
        // everybody who is not waiting has settled
        const int queued = (int)n_queue;
        if (queued == 0) {
            break;
        }
        for (int e = wave; e < queued; e += kWideThreads / 64) {
            const int q = (int)(queue[e] & 0xFFFFFu), slot = (int)(queue[e] >> 20);
            const u32 last = pool[slot * kTopK]; // the last candidate listed so far, left there by the query's thread
            query_ctx c;
            make_query_ctx<kModePlain>(pp, P.frm, P.q, q, c);
            u32 qd[8];
            load_desc(P.q.desc + 32 * (size_t)q, qd);
            u32 sel[kTopK];
            u32 t;
            window_topk<64, kTopK, kModePlain>(P.frm, c, qd, P.occupied, nullptr, sel, t, last + 1u);
            if (lane < kTopK) {
                u32 v = sel[0];
#pragma unroll
                for (int k = 1; k < kTopK; ++k) {
                    v = lane == k ? sel[k] : v;
                }
                pool[slot * kTopK + lane] = v;
            }
        }
        __syncthreads();
        if (tid == 0) {
            n_queue = 0;
        }
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            if (waiting[k] == 2) {
                waiting[k] = 0;
                ++nrows[k];
            }
        }
        __syncthreads();
    }

    // every query now holds its final decision
    u32 mynum = 0;
    int acc[kWideQueries], bin[kWideQueries];
#pragma unroll
    for (int k = 0; k < kWideQueries; ++k) {
        const int q = tid + k * kWideThreads;
        acc[k] = -1;
        bin[k] = 0;
        if (decision[k] != kNoKey32) {
            acc[k] = (int)(entry_info[sel_entry(decision[k])] & 0xFFFFFFu);
            if ((u32)acc[k] >= (u32)n) {
                *P.err = OVS_ERR_OUT_OF_RANGE; // cannot happen with lists of this frame: never index past it
                acc[k] = -1;
                continue;
            }
            atomicMax(&P.matched[acc[k]], q); // the last taker of a keypoint that nobody held
            ++mynum;
            if (orient) {
                const float qa = P.q.angle[q], ca = P.frm.kps[acc[k]].angle;
                bin[k] = angle_bin(pp.angle_sign >= 0 ? __fsub_rn(qa, ca) : __fsub_rn(ca, qa));
                if (bin[k] < 0 || 30 <= bin[k]) {
                    *P.err = OVS_ERR_OUT_OF_RANGE; // angle_histogram_.at(bin) would throw
                    bin[k] = 0;
                }
                atomicAdd(&hist[bin[k]], 1u);
            }
        }
    }
    u32 removed = 0;
    if (orient) {
        __syncthreads();
        if (wave == 0) {
            const u32 mask = kept_rotation_bins(hist, order, sort_stack);
            if (lane == 0) {
                keep_mask = mask;
            }
        }
        __syncthreads();
        const u32 keep = keep_mask;
#pragma unroll
        for (int k = 0; k < kWideQueries; ++k) {
            if (0 <= acc[k] && !((keep >> bin[k]) & 1u)) {
                if (pp.per_query_output) {
                    acc[k] = -1;
                }
                else {
                    P.matched[acc[k]] = -2;
                }
                ++removed;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kWideQueries; ++k) {
        const int q = tid + k * kWideThreads;
        if (q < nq) {
            P.acc_idx[q] = acc[k];
        }
    }
    const u32 kept = wave_sum_u32(mynum - removed);
    if (lane == 0) {
        red[wave] = kept;
    }
    __syncthreads();
    if (tid == 0) {
        u32 num = 0;
        for (int w = 0; w < kWideThreads / 64; ++w) {
            num += red[w];
        }
        *P.num_matches = num;
    }
}

// ---- get_keypoints_in_cell ------------------------------------------------------------------------
// One wavefront per query.  count_only: out_counts[q] = number of indices; otherwise the indices
// are written in the reference's order at out_indices[offsets[q]..].
__global__ __launch_bounds__(256) void k_window_list(frame_dev f, int nq, const float* __restrict__ ref_xy, const float* __restrict__ margin,
                                                     const int* __restrict__ min_level, const int* __restrict__ max_level,
                                                     int count_only, u32* __restrict__ out_counts, const u32* __restrict__ offsets,
                                                     u32* __restrict__ out_indices) {
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (qi >= nq) {
        return;
    }
    query_ctx c;
    c.rx = ref_xy[2 * qi];
    c.ry = ref_xy[2 * qi + 1];
    c.margin = margin[qi];
    c.min_level = min_level ? min_level[qi] : -1;
    c.max_level = max_level ? max_level[qi] : -1;
    c.ugate = false;
    c.stereo_gate = false;
    c.ranged = false;
    c.lo = c.hi = 0;
    c.xr = 0;
    c.sl = 0;
    u32 written = 0;
    const window w = window_of(f.grid, c.rx, c.ry, c.margin);
    if (!w.empty) {
        const int R = (int)f.grid.num_rows;
        u32* out = count_only ? nullptr : out_indices + offsets[qi];
        for (int cx = w.min_cx; cx <= w.max_cx; ++cx) {
            const u32 s = f.cell_start[cx * R + w.min_cy], e = f.cell_start[cx * R + w.max_cy + 1];
            for (u32 j0 = s; j0 < e; j0 += 64) {
                const u32 j = j0 + lane;
                bool pass = false;
                uint4 en = make_uint4(0, 0, 0, 0);
                if (j < e) {
                    en = f.g_entry[j];
                    pass = passes_gates<false>(f, c, en, nullptr, nullptr, false);
                }
                const u64 bits = __ballot(pass);
                if (pass && out) {
                    out[written + __popcll(bits & ((1ull << lane) - 1ull))] = en.w;
                }
                written += __popcll(bits);
            }
        }
    }
    if (count_only && lane == 0) {
        out_counts[qi] = written;
    }
}

// Fills the problem descriptors of a batch of consecutive-frame matches from device-side counts and
// splits the query fields out of the previous frame's keypoints.  grid = (ceil(cap / 256), n_frames).
struct scale_table {
    float v[64];
};

__global__ __launch_bounds__(256) void k_batch_setup(const ovs_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                     const int* __restrict__ counts, int n_frames, int cap,
                                                     const float* __restrict__ reproj_xy, ovs_grid grid,
                                                     float* __restrict__ scale_factors, scale_table sf_host, int num_levels,
                                                     u32* cell_start, uint4* g_entry, uint4* g_desc, int ncell,
                                                     int* sl, float* ang, uint8_t* valid, u64* lists, u32* totals, int* acc,
                                                     int* matched, u32* num_matches, int* err, proj_problem* problems) {
    const int f = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (f == 0) {
        if (i < num_levels) {
            scale_factors[i] = sf_host.v[i]; // the caller's table travels as a kernel argument: no copy to order
        }
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
        P.frm.g_entry = g_entry + qo;
        P.frm.g_desc = g_desc + 2 * qo;
        P.frm.n_entries_plus1 = 0;
        P.q.n = n_last;
        P.q.reproj_xy = reproj_xy + 2 * qo;
        P.q.x_right = nullptr;
        P.q.scale_level = sl + qo;
        P.q.angle = ang + qo;
        P.q.desc = desc + (size_t)(f - 1) * cap * 32;
        P.q.valid = valid + qo;
        P.q.has_obs = nullptr;
        P.q.range_lo = nullptr;
        P.q.range_hi = nullptr;
        P.q.reproj_xy_d = nullptr;
        P.q.bearings = nullptr;
        P.q.sin_thr = nullptr;
        P.q.stereo = nullptr;
        P.frm.inv_level_sigma_sq = nullptr;
        P.frm.bearings = nullptr;
        P.frm.tri_E = nullptr;
        P.frm.tri_amb = nullptr;
        P.frm.tri_ovr = nullptr;
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

// Candidate list of the range matchers: entry j is candidate entry_idx[j]; copies its descriptor
// next to it so that a range scan reads contiguous memory.
__global__ __launch_bounds__(256) void k_entries_from_list(const u32* __restrict__ entry_idx, int n_entries, const uint8_t* __restrict__ desc,
                                                           uint4* __restrict__ g_entry, uint4* __restrict__ g_desc) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_entries) {
        return;
    }
    const u32 i = entry_idx[j];
    g_entry[j] = make_uint4(0u, 0u, 0u, i);
    const u32* d = reinterpret_cast<const u32*>(desc + 32 * (size_t)i);
    g_desc[2 * j] = make_uint4(d[0], d[1], d[2], d[3]);
    g_desc[2 * j + 1] = make_uint4(d[4], d[5], d[6], d[7]);
}

// match::angle_checker on its own (angle_checker.h:60-134): the rotation histogram of n delta angles
// and, per entry, whether its bin is among the num_bins_thr fullest.  One wavefront; the same
// angle_bin / index_sort_by_size code the resolve kernels run.
__global__ __launch_bounds__(64) void k_angle_partition(const float* __restrict__ delta_angles, int n, int num_bins_thr,
                                                        uint8_t* __restrict__ is_valid, int* __restrict__ err) {
    __shared__ u32 hist[32];
    __shared__ u32 order[32];
    __shared__ ovs_sort::sort_frame sort_stack[ovs_sort::kSortStackEntries];
    const int lane = threadIdx.x;
    if (lane < 32) {
        hist[lane] = 0;
    }
    wave_sync();
    for (int i = lane; i < n; i += 64) {
        const int bin = angle_bin(delta_angles[i]);
        if (bin < 0 || 30 <= bin) {
            *err = OVS_ERR_OUT_OF_RANGE; // angle_histogram_.at(bin) throws
        }
        else {
            atomicAdd(&hist[bin], 1u);
        }
    }
    wave_sync();
    if (lane == 0) {
        ovs_sort::index_sort_by_size(hist, order, 30, sort_stack);
    }
    wave_sync();
    for (int i = lane; i < n; i += 64) {
        const int bin = angle_bin(delta_angles[i]);
        bool ok = false;
        for (int k = 0; k < num_bins_thr; ++k) {
            ok |= (int)order[k] == bin;
        }
        is_valid[i] = ok ? 1 : 0;
    }
}

// ---- host side ------------------------------------------------------------------------------------
// LDS plan of k_proj_resolve: the claim bitmap always lives in LDS; the per-keypoint `matched` and
// `claim` arrays join it when everything fits in 150 KB (a workgroup is one wavefront and a launch
// has one workgroup per problem, so one workgroup per compute unit may as well take its LDS).
int resolve_lds_n(int n) { return (static_cast<size_t>(n) * 8 + (static_cast<size_t>(n) / 32 + 2) * 4 <= 150 * 1024) ? std::max(n, 1) : 0; }
size_t resolve_lds_bytes(int n, int lds_n) {
    if (lds_n > 0) {
        return (static_cast<size_t>((lds_n + 31) / 32) + 1) * 4 + static_cast<size_t>(lds_n) * 8;
    }
    return (static_cast<size_t>((n + 31) / 32) + 1) * 4;
}

// launches one instance of the resolve kernel.  Dynamic LDS beyond the default limit is granted once
// per (device, kernel instance), to the largest size the LDS plan above can ask for, and never lowered.
constexpr size_t kResolveMaxLds = 150 * 1024 + 64;

template <typename Kernel>
int launch_resolve_instance(Kernel kernel, const proj_problem* d_prob, int n_problems, size_t bytes, int lds_n, const proj_params& pp, hipStream_t st) {
    if (bytes > kResolveMaxLds) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "resolve kernel: LDS plan exceeds the limit");
    }
    if (bytes > 48 * 1024) {
        static std::mutex mu;
        static std::set<std::pair<int, const void*>> granted;
        int dev = 0;
        OVS_HIP(hipGetDevice(&dev));
        const void* fn = reinterpret_cast<const void*>(kernel);
        std::lock_guard<std::mutex> lock(mu);
        if (!granted.count({dev, fn})) {
            OVS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kResolveMaxLds)));
            granted.insert({dev, fn});
        }
    }
    OVS_LAUNCH("k_proj_resolve", kernel, dim3(n_problems), dim3(64), bytes, st, d_prob, pp, lds_n);
    return OVS_OK;
}

template <int MODE>
int launch_resolve_t(const proj_problem* d_prob, int n_problems, int max_n, const proj_params& pp, hipStream_t st) {
    const int lds_n = resolve_lds_n(max_n);
    const size_t bytes = resolve_lds_bytes(max_n, lds_n);
    if (lds_n > 0) {
        // the per-frame matchers of the tracking thread get a kernel compiled for their kind
        if (MODE == kModePlain && pp.kind == kCurrLast) {
            return launch_resolve_instance(k_proj_resolve<true, false, kCurrLast>, d_prob, n_problems, bytes, lds_n, pp, st);
        }
        if (MODE == kModePlain && pp.kind == kFrameLms) {
            return launch_resolve_instance(k_proj_resolve<true, false, kFrameLms>, d_prob, n_problems, bytes, lds_n, pp, st);
        }
        if (MODE == kModePlain && pp.kind == kRange) {
            return launch_resolve_instance(k_proj_resolve<true, false, kRange>, d_prob, n_problems, bytes, lds_n, pp, st);
        }
        return launch_resolve_instance(k_proj_resolve<true, MODE>, d_prob, n_problems, bytes, lds_n, pp, st);
    }
    return launch_resolve_instance(k_proj_resolve<false, MODE>, d_prob, n_problems, bytes, 0, pp, st);
}

bool kind_is_fuse(int kind) { return kind == kFuseDetect || kind == kFuseReplace; }

// The workgroup-wide resolve kernel serves the matchers without a ratio test when the per-keypoint words fit
// in LDS and a thread gets at most kWideQueries queries -- for calls with a few problems, where a frame is
// waited for: a batch of hundreds of frames fills the device with one wave per frame and takes 0.24 ms for
// 255 frames, while sixteen mostly waiting waves per frame take 0.37 ms and hold a compute unit each
// (the four-stream step went from 5.96 to 7.85 ms).  OVS_RESOLVE=wave / wide forces one of the two.
constexpr int kWideMaxProblems = 8;
bool wide_resolve_applies(const proj_params& pp, int n_problems, int max_n, int max_nq) {
    static const int forced = [] {
        const char* e = std::getenv("OVS_RESOLVE");
        return !e ? 0 : (std::string(e) == "wave" ? 1 : (std::string(e) == "wide" ? 2 : 0));
    }();
    const bool few = forced == 2 || (forced == 0 && n_problems <= kWideMaxProblems);
    return few && (pp.kind == kCurrLast || pp.kind == kFrameKeyfrm || pp.kind == kSim3) && max_nq <= kWideThreads * kWideQueries
           && wide_lds_bytes(max_n, max_nq) <= 150 * 1024;
}

template <typename Kernel>
int launch_resolve_wide(Kernel kernel, const proj_problem* d_prob, int n_problems, int max_n, int max_nq, const proj_params& pp, hipStream_t st) {
    const size_t bytes = wide_lds_bytes(max_n, max_nq);
    if (bytes > 48 * 1024) {
        static std::mutex mu;
        static std::set<std::pair<int, const void*>> granted;
        int dev = 0;
        OVS_HIP(hipGetDevice(&dev));
        const void* fn = reinterpret_cast<const void*>(kernel);
        std::lock_guard<std::mutex> lock(mu);
        if (!granted.count({dev, fn})) {
            OVS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            granted.insert({dev, fn});
        }
    }
    OVS_LAUNCH("k_proj_resolve_wide", kernel, dim3(n_problems), dim3(kWideThreads), bytes, st, d_prob, pp, std::max(max_n, 1));
    return OVS_OK;
}

// The ratio-test matchers on the workgroup-wide kernel that rebuilds its words every round: calls with a few
// problems, as for k_proj_resolve_wide (one 2000-query frame: 0.032 ms against 0.115 ms on a single wave).
// OVS_RESOLVE=wave / wide forces one of the two.
bool jacobi_resolve_applies(const proj_params& pp, int n_problems, int max_n, int max_nq) {
    static const int forced = [] {
        const char* e = std::getenv("OVS_RESOLVE");
        return !e ? 0 : (std::string(e) == "wave" ? 1 : (std::string(e) == "wide" ? 2 : 0));
    }();
    const bool few = forced == 2 || (forced == 0 && n_problems <= kWideMaxProblems);
    return few && (pp.kind == kFrameLms || pp.kind == kRange) && max_nq <= kWideThreads * kWideQueries && jacobi_lds_bytes(max_n, max_nq) <= 140 * 1024;
}

template <typename Kernel>
int launch_resolve_jacobi(Kernel kernel, const proj_problem* d_prob, int n_problems, int max_n, int max_nq, const proj_params& pp, hipStream_t st) {
    const size_t bytes = jacobi_lds_bytes(max_n, max_nq);
    if (bytes > 48 * 1024) {
        static std::mutex mu;
        static std::set<std::pair<int, const void*>> granted;
        int dev = 0;
        OVS_HIP(hipGetDevice(&dev));
        const void* fn = reinterpret_cast<const void*>(kernel);
        std::lock_guard<std::mutex> lock(mu);
        if (!granted.count({dev, fn})) {
            OVS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
            granted.insert({dev, fn});
        }
    }
    OVS_LAUNCH("k_proj_resolve_jacobi", kernel, dim3(n_problems), dim3(kWideThreads), bytes, st, d_prob, pp, std::max(max_n, 1));
    return OVS_OK;
}

// serial_only: the single-wave kernels (the repetition after a workgroup-wide kernel reported kResolveOverflow)
int launch_resolve(const proj_problem* d_prob, int n_problems, int max_n, int max_nq, const proj_params& pp, hipStream_t st, bool serial_only = false) {
    if (!serial_only && jacobi_resolve_applies(pp, n_problems, max_n, max_nq)) {
        if (pp.kind == kFrameLms) {
            return launch_resolve_jacobi(k_proj_resolve_jacobi<kFrameLms>, d_prob, n_problems, max_n, max_nq, pp, st);
        }
        return launch_resolve_jacobi(k_proj_resolve_jacobi<kRange>, d_prob, n_problems, max_n, max_nq, pp, st);
    }
    if (!serial_only && wide_resolve_applies(pp, n_problems, max_n, max_nq)) {
        if (pp.kind == kCurrLast) {
            return launch_resolve_wide(k_proj_resolve_wide<kCurrLast>, d_prob, n_problems, max_n, max_nq, pp, st);
        }
        if (pp.kind == kFrameKeyfrm) {
            return launch_resolve_wide(k_proj_resolve_wide<kFrameKeyfrm>, d_prob, n_problems, max_n, max_nq, pp, st);
        }
        return launch_resolve_wide(k_proj_resolve_wide<kSim3>, d_prob, n_problems, max_n, max_nq, pp, st);
    }
    if (kind_is_fuse(pp.kind)) {
        return launch_resolve_t<kModeFuse>(d_prob, n_problems, max_n, pp, st);
    }
    if (pp.kind == kTri) {
        return launch_resolve_t<kModeTri>(d_prob, n_problems, max_n, pp, st);
    }
    if (pp.kind == kArea) {
        // the (distance, owner) state of every keypoint of the searched frame lives in LDS when it fits
        // (about 18 000 keypoints), else in global memory with one query resolved per round
        const int lds_n = resolve_lds_n(max_n);
        if (lds_n == 0) {
            return launch_resolve_instance(k_proj_resolve<false, kModeArea, kArea>, d_prob, n_problems, resolve_lds_bytes(max_n, 0), 0, pp, st);
        }
        return launch_resolve_instance(k_proj_resolve<true, kModeArea, kArea>, d_prob, n_problems, resolve_lds_bytes(max_n, lds_n), lds_n, pp, st);
    }
    return launch_resolve_t<kModePlain>(d_prob, n_problems, max_n, pp, st);
}

// candidate lists of n_problems problems with at most max_nq queries each
void launch_candidates(int max_nq, int n_problems, const proj_problem* d_prob, const proj_params& pp, hipStream_t st) {
    static const bool live_off = [] {
        const char* e = std::getenv("OVS_GROUP_LIVE");
        return e && std::string(e) == "8";
    }();
    const bool live = !live_off && n_problems <= kWideMaxProblems;
    const int qpb = 256 / (live ? kGroupLive : kGroup);
    const dim3 grid((max_nq + qpb - 1) / qpb, n_problems);
    if (kind_is_fuse(pp.kind)) {
        if (live) {
            OVS_LAUNCH("k_proj_candidates", (k_proj_candidates<kModeFuse, kGroupLive>), grid, dim3(256), 0, st, d_prob, pp);
        }
        else {
            OVS_LAUNCH("k_proj_candidates", k_proj_candidates<kModeFuse>, grid, dim3(256), 0, st, d_prob, pp);
        }
    }
    else if (pp.kind == kTri) {
        if (live) {
            OVS_LAUNCH("k_proj_candidates", (k_proj_candidates<kModeTri, kGroupLive>), grid, dim3(256), 0, st, d_prob, pp);
        }
        else {
            OVS_LAUNCH("k_proj_candidates", k_proj_candidates<kModeTri>, grid, dim3(256), 0, st, d_prob, pp);
        }
    }
    else {
        // kArea included: the take-over rule only acts in the resolve kernel
        if (live) {
            OVS_LAUNCH("k_proj_candidates", (k_proj_candidates<kModePlain, kGroupLive>), grid, dim3(256), 0, st, d_prob, pp);
        }
        else {
            OVS_LAUNCH("k_proj_candidates", k_proj_candidates<kModePlain>, grid, dim3(256), 0, st, d_prob, pp);
        }
    }
}

int grid_lds_bytes(const ovs_grid& g) { return static_cast<int>((2 * g.num_cols * g.num_rows + 1) * sizeof(u32)); }
// the index array fits in LDS next to the counters (64 KB of dynamic LDS need no attribute)
bool grid_items_in_lds(const ovs_grid& g, int cap) { return grid_lds_bytes(g) + 4ll * cap <= 64 * 1024; }

// Launches k_grid_build for `frames` frames of at most `cap` keypoints.  Dynamic LDS beyond the default
// limit (a grid of up to 16384 cells needs 128 KB of counters) is granted once per (device, kernel
// instance), to the most the kernel can ask for, and never lowered.
constexpr int kGridMaxLds = (2 * 16384 + 1) * 4 + 64;
int launch_grid_build_kernel(int frames, const ovs_keypoint* d_kps, const uint8_t* d_desc, const int* d_counts, int fixed_n, int cap, const ovs_grid& g,
                             u32* d_cell_start, u32* d_items, uint4* d_entry, uint4* d_gdesc, hipStream_t st) {
    const bool in_lds = grid_items_in_lds(g, cap);
    const int bytes = grid_lds_bytes(g) + (in_lds ? 4 * cap : 0);
    if (bytes > kGridMaxLds) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "grid build: LDS plan exceeds the limit");
    }
    // a few frames that are waited for: four times the threads per frame
    const bool wide = frames <= 8;
    auto kernel = in_lds ? (wide ? k_grid_build<true, 1024> : k_grid_build<true, 256>) : (wide ? k_grid_build<false, 1024> : k_grid_build<false, 256>);
    if (bytes > 48 * 1024) {
        static std::mutex mu;
        static std::set<std::pair<int, const void*>> granted;
        int dev = 0;
        OVS_HIP(hipGetDevice(&dev));
        const void* fn = reinterpret_cast<const void*>(kernel);
        std::lock_guard<std::mutex> lock(mu);
        if (!granted.count({dev, fn})) {
            OVS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kGridMaxLds));
            granted.insert({dev, fn});
        }
    }
    OVS_LAUNCH("k_grid_build", kernel, dim3(frames), dim3(wide ? 1024 : 256), bytes, st, d_kps, d_desc, d_counts, fixed_n, cap, g, d_cell_start, d_items, d_entry, d_gdesc);
    return OVS_OK;
}

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


size_t frame_bytes(const ovs_frame_view* f) {
    const size_t n = f->num_keypts, L = f->num_scale_levels;
    const size_t ncell = static_cast<size_t>(f->grid.num_cols) * f->grid.num_rows;
    return al(n * sizeof(ovs_keypoint)) + al(n * 32) + al(n * 4) + al(L * 4) + al((ncell + 1) * 4) + al(n * 4) + al(n * 16) + al(n * 32);
}

// Carves the device buffers of a host frame view and stages the caller's arrays for the upload.
struct staged_frame {
    frame_dev dev{}; // every field the staging does not set is zero (n_entries_plus1 = 0: a grid)
    u32* d_items = nullptr;
};

void stage_frame(const ovs_frame_view* f, carver& cv, host_stager& hs, staged_frame& sf_out) {
    frame_dev& out = sf_out.dev;
    const size_t n = f->num_keypts, L = f->num_scale_levels;
    const size_t ncell = static_cast<size_t>(f->grid.num_cols) * f->grid.num_rows;
    auto* d_kps = cv.take<ovs_keypoint>(n);
    auto* d_desc = cv.take<uint8_t>(n * 32);
    auto* d_sxr = cv.take<float>(n);
    auto* d_sf = cv.take<float>(L);
    auto* d_cell_start = cv.take<u32>(ncell + 1);
    auto* d_items = cv.take<u32>(n);
    auto* d_entry = cv.take<uint4>(n);
    auto* d_gdesc = cv.take<uint4>(2 * n);
    hs.put(d_kps, f->undist_keypts, n * sizeof(ovs_keypoint));
    hs.put(d_desc, f->descriptors, n * 32);
    if (f->stereo_x_right) {
        hs.put(d_sxr, f->stereo_x_right, n * 4);
    }
    hs.put(d_sf, f->scale_factors, L * 4);
    out.n = static_cast<int>(n);
    out.kps = d_kps;
    out.desc = d_desc;
    out.sxr = f->stereo_x_right ? d_sxr : nullptr;
    out.scale_factors = d_sf;
    out.num_levels = static_cast<int>(L);
    out.grid = f->grid;
    out.cell_start = d_cell_start;
    out.g_entry = d_entry;
    out.g_desc = d_gdesc;
    out.n_entries_plus1 = 0;
    out.inv_level_sigma_sq = nullptr;
    sf_out.d_items = d_items;
}

// Builds the grid of a staged frame (after the stager's upload has been enqueued on `st`).
int launch_grid_build(const staged_frame& sf, hipStream_t st) {
    const frame_dev& d = sf.dev;
    return launch_grid_build_kernel(1, d.kps, d.desc, nullptr, d.n, std::max(d.n, 1), d.grid, const_cast<u32*>(d.cell_start), sf.d_items,
                                    const_cast<uint4*>(d.g_entry), const_cast<uint4*>(d.g_desc), st);
}

// Runs one projection-matching problem given host views.
int run_projection(const ovs_frame_view* frm, const ovs_proj_queries* q, const uint8_t* occupied, proj_params pp,
                   int32_t* matched_out, int32_t* best_of_query_out, uint32_t* num_matches, int device,
                   const double* reproj_xy_d = nullptr, const float* inv_level_sigma_sq = nullptr) {
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
    const size_t n = frm->num_keypts, nq = q->n;
    const size_t total = frame_bytes(frm) + al(n) + al(nq * 8) + al(nq * 4) * 3 + al(nq * 32) + al(nq) * 2
                         + al(nq * kTopK * 8) + al(nq * 4) * 2 + al(n * 4) + al(256) * 3 + al(sizeof(proj_problem))
                         + al(nq * 16) + al(static_cast<size_t>(frm->num_scale_levels) * 4);
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    hipStream_t st = ctx->stream;

    host_stager hs;
    if ((rc = hs.init(ctx, st)) != OVS_OK) {
        return rc;
    }
    proj_problem P;
    std::memset(&P, 0, sizeof(P));
    staged_frame sfrm;
    stage_frame(frm, cv, hs, sfrm);
    P.frm = sfrm.dev;
    auto* d_occ = cv.take<uint8_t>(n);
    auto* d_rxy = cv.take<float>(nq * 2);
    auto* d_xr = cv.take<float>(nq);
    auto* d_sl = cv.take<int>(nq);
    auto* d_ang = cv.take<float>(nq);
    auto* d_qdesc = cv.take<uint8_t>(nq * 32);
    auto* d_valid = cv.take<uint8_t>(nq);
    auto* d_hasobs = cv.take<uint8_t>(nq);
    auto* d_rxy_d = cv.take<double>(nq * 2);
    auto* d_isig = cv.take<float>(static_cast<size_t>(frm->num_scale_levels));
    auto* d_prob = cv.take<proj_problem>(1);
    // results (one span for the way back) and kernel-to-kernel buffers
    auto* d_acc = cv.take<int>(nq);
    auto* d_matched = cv.take<int>(n);
    auto* d_num = cv.take<u32>(1);
    auto* d_err = cv.take<int>(1);
    auto* d_lists = cv.take<u64>(nq * kTopK);
    auto* d_totals = cv.take<u32>(nq);
    if (reproj_xy_d) {
        hs.put(d_rxy_d, reproj_xy_d, nq * 16);
        P.q.reproj_xy_d = d_rxy_d;
    }
    if (inv_level_sigma_sq) {
        hs.put(d_isig, inv_level_sigma_sq, static_cast<size_t>(frm->num_scale_levels) * 4);
        P.frm.inv_level_sigma_sq = d_isig;
    }
    if (occupied) {
        hs.put(d_occ, occupied, n);
    }
    hs.put(d_rxy, q->reproj_xy, nq * 8);
    if (q->x_right) {
        hs.put(d_xr, q->x_right, nq * 4);
    }
    hs.put(d_sl, q->scale_level, nq * 4);
    if (q->angle) {
        hs.put(d_ang, q->angle, nq * 4);
    }
    hs.put(d_qdesc, q->desc, nq * 32);
    hs.put(d_valid, q->valid, nq);
    if (q->has_observation) {
        hs.put(d_hasobs, q->has_observation, nq);
    }
    hs.fill(d_acc, 0xFF, nq * 4);
    hs.fill(d_num, 0, 4);
    hs.fill(d_err, 0, 4);

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
    hs.put(d_prob, &P, sizeof(P));
    OVS_HIP(hs.upload());

    {
        const int grc = launch_grid_build(sfrm, st);
        if (grc != OVS_OK) {
            return grc;
        }
    }
    if (nq) {
        launch_candidates(static_cast<int>(nq), 1, d_prob, pp, st);
    }
    if ((rc = launch_resolve(d_prob, 1, static_cast<int>(n), static_cast<int>(nq), pp, st)) != OVS_OK) {
        return rc;
    }
    OVS_HIP(hipGetLastError());

    int err = 0;
    uint32_t num = 0;
    if (matched_out) {
        hs.get(matched_out, d_matched, n * 4);
    }
    if (best_of_query_out) {
        hs.get(best_of_query_out, d_acc, nq * 4);
    }
    hs.get(&num, d_num, 4);
    hs.get(&err, d_err, 4);
    OVS_HIP(hs.download());
    if (err == kResolveOverflow) {
        // a query needed more candidate rows than the workgroup-wide kernel keeps: the single-wave kernel
        // resolves the same lists
        OVS_HIP(hipMemsetAsync(d_err, 0, 4, st));
        OVS_HIP(hipMemsetAsync(d_num, 0, 4, st));
        OVS_HIP(hipMemsetAsync(d_acc, 0xFF, nq * 4, st));
        if ((rc = launch_resolve(d_prob, 1, static_cast<int>(n), static_cast<int>(nq), pp, st, true)) != OVS_OK) {
            return rc;
        }
        OVS_HIP(hs.download());
    }
    if (err != 0) {
        return ovs::fail(err, "scale level or rotation bin out of range");
    }
    if (num_matches) {
        *num_matches = num;
    }
    return OVS_OK;
}

// A matching problem whose queries each scan one contiguous range of a candidate list
// (robust::brute_force_match: the whole list; bow_tree::match_*: the bucket of the query's node).
// The extra inputs of robust::match_for_triangulation on top of a ranged problem.
struct tri_extra {
    const double* cand_bearings = nullptr; // n_cand x 3
    const float* cand_sxr = nullptr;       // n_cand or nullptr (monocular)
    const double* q_bearings = nullptr;    // nq x 3, query order
    const double* q_sin_thr = nullptr;     // nq
    const double* q_thr = nullptr;         // nq: residual threshold in radians (host re-check)
    const uint8_t* q_stereo = nullptr;     // nq
    double E[12];                          // E_12 row-major, then the epipole
};

struct ranged_problem {
    const uint8_t* cand_descs = nullptr; // n_cand x 32
    const float* cand_angles = nullptr;  // n_cand or nullptr
    int n_cand = 0;
    const uint8_t* cand_occupied = nullptr; // static exclusion or nullptr
    const u32* entry_idx = nullptr;         // candidate index of every list entry
    int n_entries = 0;
    const uint8_t* q_descs = nullptr; // nq x 32 (already gathered in query order)
    const float* q_angles = nullptr;
    const u32* q_lo = nullptr;
    const u32* q_hi = nullptr;
    int nq = 0;
    float lowe = 0;
    int check_orientation = 0, angle_sign = 1, per_query_output = 0;
    const tri_extra* tri = nullptr; // non-null: kTri instead of kRange
};

// cos_residual of check_epipolar_constraint (robust.cc:320-323) in the operation order of tri_gate
// (IEEE double operations, no contraction: the host and the device get the same bits)
double tri_cos_residual(const double* E, const double* b1, const double* b2) {
    double e[3];
    for (int r = 0; r < 3; ++r) {
        e[r] = (E[3 * r] * b2[0] + E[3 * r + 1] * b2[1]) + E[3 * r + 2] * b2[2];
    }
    const double dot = (e[0] * b1[0] + e[1] * b1[1]) + e[2] * b1[2];
    const double norm = std::sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
    return dot / norm;
}

int run_ranged(const ranged_problem& rp, int32_t* matched_out, int32_t* acc_out, uint32_t* num_matches, int device) {
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    if (rp.n_cand >= (1 << 20) || rp.n_entries >= (1 << 20)) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 2^20 keypoints per frame are not supported");
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    const size_t n = rp.n_cand, ne = rp.n_entries, nq = rp.nq;
    const tri_extra* tri = rp.tri;
    // candidate keypoints only carry the angle (read by the rotation check)
    std::vector<ovs_keypoint> kps(n);
    for (size_t i = 0; i < n; ++i) {
        kps[i] = ovs_keypoint{0.f, 0.f, 0.f, rp.cand_angles ? rp.cand_angles[i] : 0.f, 0.f, 0, -1};
    }
    // kTri: borderline epipolar decisions go through the host (tri_gate); `ovr` lists the pairs whose
    // exact result is the opposite of the device's comparison.  The set of borderline pairs depends on
    // the inputs alone, so one repetition with the overrides in place settles the match.
    size_t amb_cap = 1024;
    std::vector<u32> ovr, amb;
    for (int attempt = 0;; ++attempt) {
        const size_t n_ovr = ovr.size() / 2;
        const size_t total = al(n * sizeof(ovs_keypoint)) + al(n * 32) + al(n) + al(ne * 4) + al(ne * 16) + al(ne * 32) + al(nq * 32) + al(nq * 4) * 5
                             + al(nq) + al(nq * kTopK * 8) + al(n * 4) + al(256) * 3 + al(sizeof(proj_problem))
                             + (tri ? al(n * 24) + al(n * 4) + al(nq * 24) + al(nq * 8) + al(nq) + al(12 * 8) + al((2 + 2 * amb_cap) * 4) + al((1 + 2 * n_ovr) * 4) : 0);
        if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
            return rc;
        }
        carver cv;
        cv.base = ctx->arena.as<uint8_t>();
        hipStream_t st = ctx->stream;
        host_stager hs;
        if ((rc = hs.init(ctx, st)) != OVS_OK) {
            return rc;
        }

        auto* d_kps = cv.take<ovs_keypoint>(n);
        auto* d_desc = cv.take<uint8_t>(n * 32);
        auto* d_occ = cv.take<uint8_t>(n);
        auto* d_eidx = cv.take<u32>(ne);
        auto* d_qdesc = cv.take<uint8_t>(nq * 32);
        auto* d_ang = cv.take<float>(nq);
        auto* d_lo = cv.take<u32>(nq);
        auto* d_hi = cv.take<u32>(nq);
        auto* d_valid = cv.take<uint8_t>(nq);
        double *d_cb = nullptr, *d_qb = nullptr, *d_qs = nullptr, *d_E = nullptr;
        float* d_csxr = nullptr;
        uint8_t* d_qst = nullptr;
        u32* d_ovr = nullptr;
        if (tri) {
            d_cb = cv.take<double>(n * 3);
            d_csxr = cv.take<float>(n);
            d_qb = cv.take<double>(nq * 3);
            d_qs = cv.take<double>(nq);
            d_qst = cv.take<uint8_t>(nq);
            d_E = cv.take<double>(12);
            d_ovr = cv.take<u32>(1 + 2 * n_ovr);
        }
        auto* d_prob = cv.take<proj_problem>(1);
        // results (one span for the way back) and kernel-to-kernel buffers
        auto* d_acc = cv.take<int>(nq);
        auto* d_matched = cv.take<int>(n);
        auto* d_num = cv.take<u32>(1);
        auto* d_err = cv.take<int>(1);
        u32* d_amb = tri ? cv.take<u32>(2 + 2 * amb_cap) : nullptr;
        auto* d_totals = cv.take<u32>(nq);
        auto* d_lists = cv.take<u64>(nq * kTopK);
        auto* d_entry = cv.take<uint4>(ne);
        auto* d_gdesc = cv.take<uint4>(2 * ne);

        hs.put(d_kps, kps.data(), n * sizeof(ovs_keypoint));
        hs.put(d_desc, rp.cand_descs, n * 32);
        if (rp.cand_occupied) {
            hs.put(d_occ, rp.cand_occupied, n);
        }
        hs.put(d_eidx, rp.entry_idx, ne * 4);
        hs.put(d_qdesc, rp.q_descs, nq * 32);
        if (rp.q_angles) {
            hs.put(d_ang, rp.q_angles, nq * 4);
        }
        hs.put(d_lo, rp.q_lo, nq * 4);
        hs.put(d_hi, rp.q_hi, nq * 4);
        hs.fill(d_valid, 1, nq);
        hs.fill(d_acc, 0xFF, nq * 4);
        hs.fill(d_num, 0, 4);
        hs.fill(d_err, 0, 4);

        proj_problem P;
        std::memset(&P, 0, sizeof(P));
        if (tri) {
            hs.put(d_cb, tri->cand_bearings, n * 24);
            if (tri->cand_sxr) {
                hs.put(d_csxr, tri->cand_sxr, n * 4);
            }
            hs.put(d_qb, tri->q_bearings, nq * 24);
            hs.put(d_qs, tri->q_sin_thr, nq * 8);
            hs.put(d_qst, tri->q_stereo, nq);
            hs.put(d_E, tri->E, 12 * 8);
            const u32 amb_head[2] = {0u, static_cast<u32>(amb_cap)};
            hs.put(d_amb, amb_head, 8);
            const u32 ovr_head = static_cast<u32>(n_ovr);
            hs.put(d_ovr, &ovr_head, 4);
            hs.put(d_ovr + 1, ovr.data(), n_ovr * 8);
            P.frm.bearings = d_cb;
            P.frm.sxr = tri->cand_sxr ? d_csxr : nullptr;
            P.frm.tri_E = d_E;
            P.frm.tri_amb = d_amb;
            P.frm.tri_ovr = d_ovr;
            P.q.bearings = d_qb;
            P.q.sin_thr = d_qs;
            P.q.stereo = d_qst;
        }
        P.frm.n = static_cast<int>(n);
        P.frm.kps = d_kps;
        P.frm.desc = d_desc;
        P.frm.num_levels = 1;
        P.frm.grid.num_cols = P.frm.grid.num_rows = 1;
        P.frm.g_entry = d_entry;
        P.frm.g_desc = d_gdesc;
        P.frm.n_entries_plus1 = static_cast<int>(ne) + 1;
        P.q.n = static_cast<int>(nq);
        P.q.angle = d_ang;
        P.q.desc = d_qdesc;
        P.q.valid = d_valid;
        P.q.range_lo = d_lo;
        P.q.range_hi = d_hi;
        P.occupied = rp.cand_occupied ? d_occ : nullptr;
        P.lists = d_lists;
        P.totals = d_totals;
        P.matched = d_matched;
        P.num_matches = d_num;
        P.acc_idx = d_acc;
        P.err = d_err;
        hs.put(d_prob, &P, sizeof(P));
        OVS_HIP(hs.upload());

        proj_params pp{tri ? kTri : kRange, 0.0f, OVS_DIR_NONE, 0, rp.lowe, kThrLow, rp.check_orientation, rp.angle_sign, rp.per_query_output};
        if (ne) {
            OVS_LAUNCH("k_entries_from_list", k_entries_from_list, dim3((ne + 255) / 256), dim3(256), 0, st, d_eidx, static_cast<int>(ne), d_desc, d_entry, d_gdesc);
        }
        if (nq) {
            launch_candidates(static_cast<int>(nq), 1, d_prob, pp, st);
        }
        if ((rc = launch_resolve(d_prob, 1, static_cast<int>(std::max(n, ne)), static_cast<int>(nq), pp, st)) != OVS_OK) {
            return rc;
        }
        OVS_HIP(hipGetLastError());
        int err = 0;
        uint32_t num = 0;
        if (matched_out) {
            hs.get(matched_out, d_matched, n * 4);
        }
        if (acc_out) {
            hs.get(acc_out, d_acc, nq * 4);
        }
        hs.get(&num, d_num, 4);
        hs.get(&err, d_err, 4);
        if (tri) {
            amb.assign(2 + 2 * amb_cap, 0u);
            hs.get(amb.data(), d_amb, amb.size() * 4);
        }
        OVS_HIP(hs.download());
        if (err == kResolveOverflow) {
            OVS_HIP(hipMemsetAsync(d_err, 0, 4, st));
            OVS_HIP(hipMemsetAsync(d_num, 0, 4, st));
            OVS_HIP(hipMemsetAsync(d_acc, 0xFF, nq * 4, st));
            if ((rc = launch_resolve(d_prob, 1, static_cast<int>(std::max(n, ne)), static_cast<int>(nq), pp, st, true)) != OVS_OK) {
                return rc;
            }
            OVS_HIP(hs.download());
        }
        if (err != 0) {
            return ovs::fail(err, "rotation bin out of range");
        }
        *num_matches = num;
        if (!tri || amb[0] == 0) {
            return OVS_OK;
        }
        if (amb[0] > amb_cap) {
            amb_cap = amb[0]; // room for every borderline pair, then once more
            continue;
        }
        std::vector<std::pair<u32, u32>> flips;
        for (u32 k = 0; k < amb[0]; ++k) {
            const u32 qi = amb[2 + 2 * k], j = amb[3 + 2 * k];
            const u32 idx_2 = rp.entry_idx[j];
            const double cr = tri_cos_residual(tri->E, tri->q_bearings + 3 * static_cast<size_t>(qi), tri->cand_bearings + 3 * static_cast<size_t>(idx_2));
            const bool dev = -1.0 <= cr && cr < tri->q_sin_thr[qi];
            // the reference's expression (robust.cc:324-334) with the host's libm
            const double residual_rad = M_PI / 2.0 - std::abs(std::acos(cr));
            const bool exact = residual_rad < tri->q_thr[qi];
            if (dev != exact) {
                flips.emplace_back(qi, j);
            }
        }
        std::sort(flips.begin(), flips.end());
        std::vector<u32> want;
        for (const auto& f : flips) {
            want.push_back(f.first);
            want.push_back(f.second);
        }
        if (want == ovr) {
            return OVS_OK; // this run already used the exact decisions
        }
        if (attempt >= 3) {
            return ovs::fail(OVS_ERR_HIP, "triangulation matcher: borderline epipolar decisions did not settle");
        }
        ovr.swap(want);
    }
}

// Queries of the BoW matchers in the reference's iteration order: the merge-join over the two
// feature vectors (bow_tree.cc:37-111), then the keypoints of the first vector's node.
struct bow_queries {
    std::vector<u32> q_src; // keypoint index in the first (query) set
    std::vector<u32> lo, hi;
    std::vector<uint8_t> descs;
    std::vector<float> angles;
};

int build_bow_queries(const ovs_bow_feat_vec* qv, const uint8_t* q_descs, const float* q_angles, const uint8_t* q_valid, int nq_kp,
                      const ovs_bow_feat_vec* cv_, int n_cand, bow_queries& out) {
    int a = 0, b = 0;
    while (a != qv->num_nodes && b != cv_->num_nodes) {
        if (qv->node_ids[a] == cv_->node_ids[b]) {
            for (u32 k = qv->offsets[a]; k < qv->offsets[a + 1]; ++k) {
                const u32 idx = qv->indices[k];
                if (idx >= static_cast<u32>(nq_kp)) {
                    return ovs::fail(OVS_ERR_OUT_OF_RANGE, "keypoint index out of range");
                }
                if (q_valid && !q_valid[idx]) {
                    continue;
                }
                out.q_src.push_back(idx);
                out.lo.push_back(cv_->offsets[b]);
                out.hi.push_back(cv_->offsets[b + 1]);
                out.descs.insert(out.descs.end(), q_descs + 32 * static_cast<size_t>(idx), q_descs + 32 * static_cast<size_t>(idx) + 32);
                out.angles.push_back(q_angles ? q_angles[idx] : 0.f);
            }
            ++a;
            ++b;
        }
        else if (qv->node_ids[a] < cv_->node_ids[b]) {
            a = static_cast<int>(std::lower_bound(qv->node_ids, qv->node_ids + qv->num_nodes, cv_->node_ids[b]) - qv->node_ids);
        }
        else {
            b = static_cast<int>(std::lower_bound(cv_->node_ids, cv_->node_ids + cv_->num_nodes, qv->node_ids[a]) - cv_->node_ids);
        }
    }
    const u32 total = cv_->num_nodes ? cv_->offsets[cv_->num_nodes] : 0;
    for (u32 k = 0; k < total; ++k) {
        if (cv_->indices[k] >= static_cast<u32>(n_cand)) {
            return ovs::fail(OVS_ERR_OUT_OF_RANGE, "keypoint index out of range");
        }
    }
    return OVS_OK;
}

int validate_bow(const ovs_bow_feat_vec* v) {
    if (!v || v->num_nodes < 0 || (v->num_nodes > 0 && (!v->node_ids || !v->offsets || !v->indices))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid feature vector");
    }
    return OVS_OK;
}

} // namespace

extern "C" {

int ovs_get_cell_indices(const ovs_grid* g, float x, float y, int* cell_x, int* cell_y) {
    // data::get_cell_indices, src/openvslam/data/common.h:59-64 (pure host arithmetic)
    *cell_x = ovs::cv_floor_d(static_cast<double>(x - g->min_x) * g->inv_cell_width);
    *cell_y = ovs::cv_floor_d(static_cast<double>(y - g->min_y) * g->inv_cell_height);
    return (0 <= *cell_x && *cell_x < static_cast<int>(g->num_cols) && 0 <= *cell_y && *cell_y < static_cast<int>(g->num_rows)) ? 1 : 0;
}

int ovs_get_keypoints_in_cell(const ovs_frame_view* frm, int n_queries,
                              const float* ref_xy, const float* margin,
                              const int32_t* min_level, const int32_t* max_level,
                              uint32_t* out_indices, int cap, uint32_t* out_offsets, int device) {
    int rc = validate_frame(frm);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n_queries < 0 || !out_offsets || (n_queries > 0 && (!ref_xy || !margin)) || cap < 0 || (cap > 0 && !out_indices)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if ((rc = check_device(device)) != OVS_OK) {
        return rc;
    }
    out_offsets[0] = 0;
    if (n_queries == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    const size_t nq = n_queries;
    const size_t total = frame_bytes(frm) + al(nq * 8) + al(nq * 4) * 5 + al(static_cast<size_t>(cap) * 4 + 4);
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    hipStream_t st = ctx->stream;
    host_stager hs;
    if ((rc = hs.init(ctx, st)) != OVS_OK) {
        return rc;
    }
    staged_frame sfrm;
    stage_frame(frm, cv, hs, sfrm);
    const frame_dev& fd = sfrm.dev;
    auto* d_xy = cv.take<float>(nq * 2);
    auto* d_m = cv.take<float>(nq);
    auto* d_lo = cv.take<int>(nq);
    auto* d_hi = cv.take<int>(nq);
    auto* d_cnt = cv.take<u32>(nq);
    auto* d_off = cv.take<u32>(nq);
    auto* d_idx = cv.take<u32>(static_cast<size_t>(cap) + 1);
    hs.put(d_xy, ref_xy, nq * 8);
    hs.put(d_m, margin, nq * 4);
    if (min_level) {
        hs.put(d_lo, min_level, nq * 4);
    }
    if (max_level) {
        hs.put(d_hi, max_level, nq * 4);
    }
    OVS_HIP(hs.upload());
    {
        const int grc = launch_grid_build(sfrm, st);
        if (grc != OVS_OK) {
            return grc;
        }
    }
    OVS_LAUNCH("k_window_list", k_window_list, dim3((n_queries + 3) / 4), dim3(256), 0, st, fd, n_queries, d_xy, d_m,
               min_level ? d_lo : nullptr, max_level ? d_hi : nullptr, 1, d_cnt, static_cast<const u32*>(nullptr), static_cast<u32*>(nullptr));
    std::vector<u32> counts(nq);
    OVS_HIP(hipMemcpyAsync(counts.data(), d_cnt, nq * 4, hipMemcpyDeviceToHost, st));
    OVS_HIP(hipStreamSynchronize(st));
    unsigned long long run = 0;
    std::vector<u32> offs(nq);
    for (size_t i = 0; i < nq; ++i) {
        offs[i] = static_cast<u32>(run);
        run += counts[i];
        out_offsets[i + 1] = static_cast<u32>(run);
    }
    if (run > static_cast<unsigned long long>(cap)) {
        return ovs::fail(OVS_ERR_CAPACITY, "index buffer too small");
    }
    if (run == 0) {
        return OVS_OK;
    }
    OVS_HIP(hipMemcpyAsync(d_off, offs.data(), nq * 4, hipMemcpyHostToDevice, st));
    OVS_LAUNCH("k_window_list", k_window_list, dim3((n_queries + 3) / 4), dim3(256), 0, st, fd, n_queries, d_xy, d_m,
               min_level ? d_lo : nullptr, max_level ? d_hi : nullptr, 0, d_cnt, d_off, d_idx);
    OVS_HIP(hipMemcpyAsync(out_indices, d_idx, run * 4, hipMemcpyDeviceToHost, st));
    OVS_HIP(hipStreamSynchronize(st));
    return OVS_OK;
}

int ovs_match_current_and_last_frames(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                      const uint8_t* curr_occupied, float margin, int direction,
                                      int last_num_scale_levels, float lowe_ratio, int check_orientation,
                                      int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kCurrLast, margin, direction, last_num_scale_levels, lowe_ratio, kThrHigh, check_orientation, 1, 0};
    return run_projection(curr, q, curr_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_frame_and_landmarks(const ovs_frame_view* frm, const ovs_proj_queries* q,
                                  const uint8_t* frm_occupied, float margin, float lowe_ratio,
                                  int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kFrameLms, margin, OVS_DIR_NONE, 0, lowe_ratio, kThrHigh, 0, 1, 0};
    return run_projection(frm, q, frm_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_frame_and_keyframe(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                 const uint8_t* curr_occupied, float margin, uint32_t hamm_dist_thr,
                                 int check_orientation,
                                 int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kFrameKeyfrm, margin, OVS_DIR_NONE, 0, 0.0f, hamm_dist_thr, check_orientation, 1, 0};
    return run_projection(curr, q, curr_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_by_sim3_transform(const ovs_frame_view* keyfrm, const ovs_proj_queries* q,
                                const uint8_t* keyfrm_occupied, float margin,
                                int32_t* matched_query, uint32_t* num_matches, int device) {
    proj_params pp{kSim3, margin, OVS_DIR_NONE, 0, 0.0f, kThrLow, 0, 1, 0};
    return run_projection(keyfrm, q, keyfrm_occupied, pp, matched_query, nullptr, num_matches, device);
}

int ovs_match_keyframe_one_way(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, float margin,
                               int32_t* best_idx_of_query, int device) {
    proj_params pp{kOneWay, margin, OVS_DIR_NONE, 0, 0.0f, kThrHigh, 0, 1, 0};
    return run_projection(keyfrm, q, nullptr, pp, nullptr, best_idx_of_query, nullptr, device);
}

int ovs_fuse_best_candidates(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, const double* reproj_xy_d,
                             const float* inv_level_sigma_sq, float margin, int replace_mode,
                             int32_t* best_idx_of_query, int device) {
    if (replace_mode && (!inv_level_sigma_sq || (q && q->n > 0 && (!reproj_xy_d || !q->x_right)))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "replace mode needs the double reprojections, x_right and inv_level_sigma_sq");
    }
    proj_params pp{replace_mode ? kFuseReplace : kFuseDetect, margin, OVS_DIR_NONE, 0, 0.0f, kThrLow, 0, 1, 0};
    return run_projection(keyfrm, q, nullptr, pp, nullptr, best_idx_of_query, nullptr, device, replace_mode ? reproj_xy_d : nullptr,
                          replace_mode ? inv_level_sigma_sq : nullptr);
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

int ovs_brute_force_match(const uint8_t* descs_1, const float* angles_1, int n1,
                          const uint8_t* descs_2, const float* angles_2, const uint8_t* valid_2, int n2,
                          float lowe_ratio, int check_orientation,
                          int32_t* matches, uint32_t* num_matches, int device) {
    // robust::brute_force_match (robust.cc:209-315): queries = keyframe keypoints with a valid
    // landmark in index order, candidates = all frame keypoints in index order
    if (n1 < 0 || n2 < 0 || (n1 > 0 && !descs_1) || (n2 > 0 && !descs_2) || !matches || !num_matches
        || (check_orientation && ((n1 > 0 && !angles_1) || (n2 > 0 && !angles_2)))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    std::vector<u32> entries(n1), q_src, lo, hi;
    for (int i = 0; i < n1; ++i) {
        entries[i] = static_cast<u32>(i);
    }
    std::vector<uint8_t> qd;
    std::vector<float> qa;
    for (int i = 0; i < n2; ++i) {
        if (valid_2 && !valid_2[i]) {
            continue;
        }
        q_src.push_back(static_cast<u32>(i));
        lo.push_back(0);
        hi.push_back(static_cast<u32>(n1));
        qd.insert(qd.end(), descs_2 + 32 * static_cast<size_t>(i), descs_2 + 32 * static_cast<size_t>(i) + 32);
        qa.push_back(angles_2 ? angles_2[i] : 0.f);
    }
    ranged_problem rp;
    rp.cand_descs = descs_1;
    rp.cand_angles = angles_1;
    rp.n_cand = n1;
    rp.entry_idx = entries.data();
    rp.n_entries = n1;
    rp.q_descs = qd.data();
    rp.q_angles = qa.data();
    rp.q_lo = lo.data();
    rp.q_hi = hi.data();
    rp.nq = static_cast<int>(q_src.size());
    rp.lowe = lowe_ratio;
    rp.check_orientation = check_orientation;
    rp.angle_sign = -1; // keypts_1[best].angle - keypts_2[idx_2].angle, robust.cc:288-289
    std::vector<int32_t> matched(n1, -1);
    int rc = run_ranged(rp, matched.data(), nullptr, num_matches, device);
    if (rc != OVS_OK) {
        return rc;
    }
    unsigned k = 0;
    for (int i = 0; i < n1; ++i) {
        if (0 <= matched[i]) {
            matches[2 * k] = i;
            matches[2 * k + 1] = static_cast<int32_t>(q_src[matched[i]]);
            ++k;
        }
    }
    return OVS_OK;
}

int ovs_bow_match_frame_and_keyframe(const ovs_bow_feat_vec* kf_vec, const uint8_t* kf_descs,
                                     const float* kf_angles, const uint8_t* valid_kf, int n_kf,
                                     const ovs_bow_feat_vec* frm_vec, const uint8_t* frm_descs,
                                     const float* frm_angles, int n_frm,
                                     float lowe_ratio, int check_orientation,
                                     int32_t* matched_kf_idx_in_frm, uint32_t* num_matches, int device) {
    int rc;
    if ((rc = validate_bow(kf_vec)) != OVS_OK || (rc = validate_bow(frm_vec)) != OVS_OK) {
        return rc;
    }
    if (n_kf < 0 || n_frm < 0 || !matched_kf_idx_in_frm || !num_matches || (n_kf > 0 && !kf_descs) || (n_frm > 0 && !frm_descs)
        || (check_orientation && ((n_kf > 0 && !kf_angles) || (n_frm > 0 && !frm_angles)))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    bow_queries bq;
    if ((rc = build_bow_queries(kf_vec, kf_descs, kf_angles, valid_kf, n_kf, frm_vec, n_frm, bq)) != OVS_OK) {
        return rc;
    }
    ranged_problem rp;
    rp.cand_descs = frm_descs;
    rp.cand_angles = frm_angles;
    rp.n_cand = n_frm;
    rp.entry_idx = frm_vec->indices;
    rp.n_entries = frm_vec->num_nodes ? static_cast<int>(frm_vec->offsets[frm_vec->num_nodes]) : 0;
    rp.q_descs = bq.descs.data();
    rp.q_angles = bq.angles.data();
    rp.q_lo = bq.lo.data();
    rp.q_hi = bq.hi.data();
    rp.nq = static_cast<int>(bq.q_src.size());
    rp.lowe = lowe_ratio;
    rp.check_orientation = check_orientation;
    rp.angle_sign = 1; // keyfrm angle - frame angle, bow_tree.cc:92-93
    std::vector<int32_t> matched(n_frm, -1);
    if ((rc = run_ranged(rp, matched.data(), nullptr, num_matches, device)) != OVS_OK) {
        return rc;
    }
    for (int i = 0; i < n_frm; ++i) {
        matched_kf_idx_in_frm[i] = 0 <= matched[i] ? static_cast<int32_t>(bq.q_src[matched[i]]) : -1;
    }
    return OVS_OK;
}

int ovs_bow_match_keyframes(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1, const float* angles_1,
                            const uint8_t* valid_1, int n1,
                            const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2, const float* angles_2,
                            const uint8_t* valid_2, int n2,
                            float lowe_ratio, int check_orientation,
                            int32_t* matched_idx2_in_kf1, uint32_t* num_matches, int device) {
    int rc;
    if ((rc = validate_bow(vec_1)) != OVS_OK || (rc = validate_bow(vec_2)) != OVS_OK) {
        return rc;
    }
    if (n1 < 0 || n2 < 0 || !matched_idx2_in_kf1 || !num_matches || (n1 > 0 && !descs_1) || (n2 > 0 && !descs_2)
        || (check_orientation && ((n1 > 0 && !angles_1) || (n2 > 0 && !angles_2)))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    bow_queries bq;
    if ((rc = build_bow_queries(vec_1, descs_1, angles_1, valid_1, n1, vec_2, n2, bq)) != OVS_OK) {
        return rc;
    }
    std::vector<uint8_t> occupied;
    if (valid_2) {
        occupied.resize(n2);
        for (int i = 0; i < n2; ++i) {
            occupied[i] = valid_2[i] ? 0 : 1; // keyframe-2 keypoints without a usable landmark are skipped
        }
    }
    ranged_problem rp;
    rp.cand_descs = descs_2;
    rp.cand_angles = angles_2;
    rp.n_cand = n2;
    rp.cand_occupied = valid_2 ? occupied.data() : nullptr;
    rp.entry_idx = vec_2->indices;
    rp.n_entries = vec_2->num_nodes ? static_cast<int>(vec_2->offsets[vec_2->num_nodes]) : 0;
    rp.q_descs = bq.descs.data();
    rp.q_angles = bq.angles.data();
    rp.q_lo = bq.lo.data();
    rp.q_hi = bq.hi.data();
    rp.nq = static_cast<int>(bq.q_src.size());
    rp.lowe = lowe_ratio;
    rp.check_orientation = check_orientation;
    rp.angle_sign = 1;       // keyframe-1 angle - keyframe-2 angle, bow_tree.cc:220-221
    rp.per_query_output = 1; // the rotation check clears matched_lms_in_keyfrm_1[idx_1]
    std::vector<int32_t> acc(bq.q_src.size(), -1);
    if ((rc = run_ranged(rp, nullptr, acc.data(), num_matches, device)) != OVS_OK) {
        return rc;
    }
    std::fill(matched_idx2_in_kf1, matched_idx2_in_kf1 + n1, -1);
    for (size_t q = 0; q < bq.q_src.size(); ++q) {
        if (0 <= acc[q]) {
            matched_idx2_in_kf1[bq.q_src[q]] = acc[q];
        }
    }
    return OVS_OK;
}

int ovs_angle_checker_partition(const float* delta_angles, int n, int num_bins_thr, uint8_t* is_valid, int device) {
    // match::angle_checker<T>(30, num_bins_thr): append_delta_angle for every entry, then
    // get_valid_matches / get_invalid_matches (angle_checker.h:60-134)
    if (n < 0 || num_bins_thr < 0 || 30 < num_bins_thr || (n > 0 && (!delta_angles || !is_valid))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument"); // assert(num_bins_thr_ <= histogram_length_), angle_checker.h:64
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
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    const size_t nn = n;
    if ((rc = ctx->arena.reserve(al(nn * 4) + al(nn) + al(4))) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    host_stager hs;
    if ((rc = hs.init(ctx, ctx->stream)) != OVS_OK) {
        return rc;
    }
    auto* d_ang = cv.take<float>(nn);
    auto* d_valid = cv.take<uint8_t>(nn);
    auto* d_err = cv.take<int>(1);
    hs.put(d_ang, delta_angles, nn * 4);
    hs.fill(d_err, 0, 4);
    OVS_HIP(hs.upload());
    OVS_LAUNCH("k_angle_partition", k_angle_partition, dim3(1), dim3(64), 0, ctx->stream, d_ang, n, num_bins_thr, d_valid, d_err);
    OVS_HIP(hipGetLastError());
    int err = 0;
    hs.get(is_valid, d_valid, nn);
    hs.get(&err, d_err, 4);
    OVS_HIP(hs.download());
    if (err != 0) {
        return ovs::fail(err, "rotation bin out of range");
    }
    return OVS_OK;
}

int ovs_match_for_triangulation(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1,
                                const ovs_keypoint* undist_keypts_1, const double* bearings_1,
                                const float* stereo_x_right_1, const uint8_t* has_lm_1, int n1,
                                const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2,
                                const ovs_keypoint* undist_keypts_2, const double* bearings_2,
                                const float* stereo_x_right_2, const uint8_t* has_lm_2, int n2,
                                const double* E_12, const double* epipole_in_2,
                                const float* scale_factors_1, int num_scale_levels_1,
                                int check_orientation,
                                uint32_t* pairs, uint32_t* num_matches, int device) {
    // robust::match_for_triangulation (robust.cc:18-172).  Queries = the keypoints of keyframe 1
    // without a landmark, in the order of the merge-join over the two feature vectors; the candidates
    // of a query are the keypoints of keyframe 2 in the same vocabulary node.  Distances, both
    // epipolar gates and the sequential "already matched" rule run on the device.
    int rc;
    if ((rc = validate_bow(vec_1)) != OVS_OK || (rc = validate_bow(vec_2)) != OVS_OK) {
        return rc;
    }
    if (n1 < 0 || n2 < 0 || !E_12 || !epipole_in_2 || !scale_factors_1 || !pairs || !num_matches
        || (n1 > 0 && (!descs_1 || !undist_keypts_1 || !bearings_1)) || (n2 > 0 && (!descs_2 || !undist_keypts_2 || !bearings_2))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    std::vector<u32> q_src, lo, hi;
    std::vector<uint8_t> qd, q_stereo;
    std::vector<float> qa;
    std::vector<double> qb, q_thr, q_sin;
    {
        int a = 0, b = 0;
        while (a != vec_1->num_nodes && b != vec_2->num_nodes) {
            if (vec_1->node_ids[a] == vec_2->node_ids[b]) {
                for (u32 k = vec_1->offsets[a]; k < vec_1->offsets[a + 1]; ++k) {
                    const u32 idx_1 = vec_1->indices[k];
                    if (idx_1 >= static_cast<u32>(n1)) {
                        return ovs::fail(OVS_ERR_OUT_OF_RANGE, "keyframe 1 index out of range");
                    }
                    if (has_lm_1 && has_lm_1[idx_1]) {
                        continue; // robust.cc:62-66
                    }
                    const ovs_keypoint& keypt_1 = undist_keypts_1[idx_1];
                    if (keypt_1.octave < 0 || num_scale_levels_1 <= keypt_1.octave) {
                        return ovs::fail(OVS_ERR_OUT_OF_RANGE, "octave out of range"); // scale_factors_.at(), robust.cc:120
                    }
                    q_src.push_back(idx_1);
                    lo.push_back(vec_2->offsets[b]);
                    hi.push_back(vec_2->offsets[b + 1]);
                    qd.insert(qd.end(), descs_1 + 32 * static_cast<size_t>(idx_1), descs_1 + 32 * static_cast<size_t>(idx_1) + 32);
                    qa.push_back(keypt_1.angle);
                    qb.insert(qb.end(), bearings_1 + 3 * static_cast<size_t>(idx_1), bearings_1 + 3 * static_cast<size_t>(idx_1) + 3);
                    q_stereo.push_back(0 <= (stereo_x_right_1 ? stereo_x_right_1[idx_1] : -1.0f) ? 1 : 0);
                    // residual_rad_thr * bearing_1_scale_factor, robust.cc:329-334 (the float factor widens)
                    constexpr double residual_rad_thr = 0.2 * M_PI / 180.0;
                    const double thr = residual_rad_thr * scale_factors_1[keypt_1.octave];
                    q_thr.push_back(thr);
                    q_sin.push_back(std::sin(thr));
                }
                ++a;
                ++b;
            }
            else if (vec_1->node_ids[a] < vec_2->node_ids[b]) {
                a = static_cast<int>(std::lower_bound(vec_1->node_ids, vec_1->node_ids + vec_1->num_nodes, vec_2->node_ids[b]) - vec_1->node_ids);
            }
            else {
                b = static_cast<int>(std::lower_bound(vec_2->node_ids, vec_2->node_ids + vec_2->num_nodes, vec_1->node_ids[a]) - vec_2->node_ids);
            }
        }
    }
    const u32 n_entries = vec_2->num_nodes ? vec_2->offsets[vec_2->num_nodes] : 0;
    for (u32 k = 0; k < n_entries; ++k) {
        if (vec_2->indices[k] >= static_cast<u32>(n2)) {
            return ovs::fail(OVS_ERR_OUT_OF_RANGE, "keyframe 2 index out of range");
        }
    }
    std::vector<float> angles_2(n2);
    for (int i = 0; i < n2; ++i) {
        angles_2[i] = undist_keypts_2[i].angle;
    }
    tri_extra tx;
    tx.cand_bearings = bearings_2;
    tx.cand_sxr = stereo_x_right_2;
    tx.q_bearings = qb.data();
    tx.q_sin_thr = q_sin.data();
    tx.q_thr = q_thr.data();
    tx.q_stereo = q_stereo.data();
    std::memcpy(tx.E, E_12, 9 * sizeof(double));
    std::memcpy(tx.E + 9, epipole_in_2, 3 * sizeof(double));
    ranged_problem rp;
    rp.cand_descs = descs_2;
    rp.cand_angles = angles_2.data();
    rp.n_cand = n2;
    rp.cand_occupied = has_lm_2; // robust.cc:81-85
    rp.entry_idx = vec_2->indices;
    rp.n_entries = static_cast<int>(n_entries);
    rp.q_descs = qd.data();
    rp.q_angles = qa.data();
    rp.q_lo = lo.data();
    rp.q_hi = hi.data();
    rp.nq = static_cast<int>(q_src.size());
    rp.check_orientation = check_orientation;
    rp.angle_sign = 1;       // keypt_1.angle - keyframe-2 angle, robust.cc:136-137
    rp.per_query_output = 1; // the rotation check clears matched_indices_2_in_keyfrm_1[idx_1]
    rp.tri = &tx;
    std::vector<int32_t> acc(q_src.size(), -1);
    if ((rc = run_ranged(rp, nullptr, acc.data(), num_matches, device)) != OVS_OK) {
        return rc;
    }
    std::vector<int32_t> matched_indices_2_in_keyfrm_1(n1, -1);
    for (size_t q = 0; q < q_src.size(); ++q) {
        if (0 <= acc[q]) {
            matched_indices_2_in_keyfrm_1[q_src[q]] = acc[q];
        }
    }
    unsigned k = 0;
    for (int idx_1 = 0; idx_1 < n1; ++idx_1) {
        if (0 <= matched_indices_2_in_keyfrm_1[idx_1]) {
            pairs[2 * k] = static_cast<uint32_t>(idx_1);
            pairs[2 * k + 1] = static_cast<uint32_t>(matched_indices_2_in_keyfrm_1[idx_1]);
            ++k;
        }
    }
    return OVS_OK;
}

int ovs_match_in_consistent_area(const ovs_keypoint* undist_keypts_1, const uint8_t* descs_1, int n1,
                                 const ovs_frame_view* frm_2, float* prev_matched_pts, int margin,
                                 float lowe_ratio, int check_orientation,
                                 int32_t* matched_indices_2_in_frm_1, uint32_t* num_matches, int device) {
    // area::match_in_consistent_area (area.cc:8-112): every keypoint of frame 1 is a query (only those
    // at scale level 0 search, area.cc:24-26) around its previously matched point; the window scan,
    // the distances and the sequential take-over rule run on the device.
    if (n1 < 0 || !frm_2 || !num_matches || (n1 > 0 && (!undist_keypts_1 || !descs_1 || !prev_matched_pts || !matched_indices_2_in_frm_1))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (n1 >= (1 << 20)) {
        // the take-over state packs distance << 20 | index of the owning keypoint of frame 1
        return ovs::fail(OVS_ERR_UNSUPPORTED, "match::area: more than 2^20 keypoints in frame 1 are not supported");
    }
    std::vector<int32_t> sl(n1);
    std::vector<float> ang(n1);
    std::vector<uint8_t> valid(n1);
    for (int i = 0; i < n1; ++i) {
        sl[i] = undist_keypts_1[i].octave;
        ang[i] = undist_keypts_1[i].angle;
        valid[i] = 0 < undist_keypts_1[i].octave ? 0 : 1;
    }
    ovs_proj_queries q;
    std::memset(&q, 0, sizeof(q));
    q.n = n1;
    q.reproj_xy = prev_matched_pts;
    q.scale_level = sl.data();
    q.angle = ang.data();
    q.desc = descs_1;
    q.valid = valid.data();
    uint32_t num = 0;
    proj_params pp{kArea, static_cast<float>(margin), OVS_DIR_NONE, 0, lowe_ratio, kThrLow, check_orientation, 1, 1};
    int rc = run_projection(frm_2, &q, nullptr, pp, nullptr, matched_indices_2_in_frm_1, &num, device);
    if (rc != OVS_OK) {
        return rc;
    }
    for (int idx_1 = 0; idx_1 < n1; ++idx_1) {
        if (0 <= matched_indices_2_in_frm_1[idx_1]) { // area.cc:105-109
            const ovs_keypoint& kp = frm_2->undist_keypts[matched_indices_2_in_frm_1[idx_1]];
            prev_matched_pts[2 * idx_1] = kp.x;
            prev_matched_pts[2 * idx_1 + 1] = kp.y;
        }
    }
    *num_matches = num;
    return OVS_OK;
}

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
    if (reinterpret_cast<uintptr_t>(d_desc) & 15) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "descriptors must be 16-byte aligned");
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
    hipStream_t st = static_cast<hipStream_t>(stream_);
    thread_ctx* ctx = ctx_for_stream(device, st);
    const size_t F = n_frames, fc = F * cap;
    const size_t total = al(64 * 4) + al(F * (ncell + 1) * 4) + al(fc * 4) * 5 + al(fc * 16) + al(fc * 32) + al(fc) + al(fc * kTopK * 8)
                         + al(256) + al(F * sizeof(proj_problem));
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_sf = cv.take<float>(64);
    auto* d_cell_start = cv.take<u32>(F * (ncell + 1));
    auto* d_items = cv.take<u32>(fc);
    auto* d_sl = cv.take<int>(fc);
    auto* d_ang = cv.take<float>(fc);
    auto* d_totals = cv.take<u32>(fc);
    auto* d_acc = cv.take<int>(fc);
    auto* d_entry = cv.take<uint4>(fc);
    auto* d_gdesc = cv.take<uint4>(2 * fc);
    auto* d_valid = cv.take<uint8_t>(fc);
    auto* d_lists = cv.take<u64>(fc * kTopK);
    auto* d_err = cv.take<int>(1);
    auto* d_prob = cv.take<proj_problem>(F);
    scale_table sf_arg;
    std::memset(&sf_arg, 0, sizeof(sf_arg));
    std::memcpy(sf_arg.v, scale_factors, static_cast<size_t>(num_scale_levels) * sizeof(float));

    proj_params pp{kCurrLast, margin, OVS_DIR_NONE, num_scale_levels, 0.0f, kThrHigh, check_orientation, 1, 0};
    OVS_LAUNCH("k_batch_setup", k_batch_setup, dim3((cap + 255) / 256, n_frames), dim3(256), 0, st,
               d_kps, d_desc, d_counts, n_frames, cap, d_reproj_xy, *grid, d_sf, sf_arg, num_scale_levels,
               d_cell_start, d_entry, d_gdesc, static_cast<int>(ncell), d_sl, d_ang, d_valid, d_lists, d_totals, d_acc,
               d_matched, d_num_matches, d_err, d_prob);
    if ((rc = launch_grid_build_kernel(n_frames, d_kps, d_desc, d_counts, 0, cap, *grid, d_cell_start, d_items, d_entry, d_gdesc, st)) != OVS_OK) {
        return rc;
    }
    if (n_frames > 1) {
        launch_candidates(cap, n_frames - 1, d_prob, pp, st);
        if ((rc = launch_resolve(d_prob, n_frames - 1, cap, cap, pp, st)) != OVS_OK) {
            return rc;
        }
    }
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

} // extern "C"
