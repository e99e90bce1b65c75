// ovs_stereo.hip -- match::stereo::compute (src/openvslam/match/stereo.cc:18-113) for gfx950:
// row-bucketed Hamming search of the right image's keypoints for every left keypoint, 11 x 11
// patch SAD over 11 horizontal offsets with a parabola fit for the sub-pixel disparity, and the
// median-of-SAD outlier rejection.
//
// Kernels
//   k_stereo_rows    one workgroup: CSR of right keypoint indices per image row, keypoint i being
//                    listed in rows floor(y - r) .. ceil(y + r), r = 2 * scale[octave], ascending
//                    index inside a row (get_right_keypoint_indices_in_each_row, stereo.cc:115-142)
//   k_stereo_match   16 lanes per left keypoint: candidates of row (size_t)y_left, octave and
//                    disparity gates, strict-< Hamming argmin below 75 (stereo.cc:144-177), then the
//                    patch correlation (integer sums, exact in any order) and the fp32 / fp64
//                    parabola arithmetic of stereo.cc:179-250 in the reference's operation order
//   k_stereo_median  one workgroup: k-th smallest SAD by a two-level radix histogram, then
//                    invalidation above 2 x median (stereo.cc:96-112)
#include "ovs_match_common.h"

#include <algorithm>
#include <cmath>
#include <cstring>

using namespace ovsm;

namespace {

constexpr int kMaxLevels = 16;
constexpr int kSG = 16; // lanes per left keypoint
constexpr u32 kStereoThr = (kThrHigh + kThrLow) / 2; // hamm_dist_thr_, stereo.h:99

struct stereo_images {
    const uint8_t* l[kMaxLevels];
    const uint8_t* r[kMaxLevels];
    long long ls[kMaxLevels], rs[kMaxLevels];
    int lw[kMaxLevels], lh[kMaxLevels], rw[kMaxLevels], rh[kMaxLevels];
    float scale[kMaxLevels], inv_scale[kMaxLevels];
    int num_levels;
    // batches of pairs (blockIdx.y = pair): bytes between consecutive frames of a level, and
    // elements between consecutive pairs in the keypoint / output arrays, row tables and item lists
    long long lfs[kMaxLevels], rfs[kMaxLevels];
    long long kp_stride, rows_stride, items_stride;
};

// T threads per pair: 256 in batches (many pairs fill the device), 1024 for a call with a few pairs, where the chains of
// dependent LDS reads of one workgroup (entry -> row -> row bounds -> the row's entries) are what the caller waits for.
template <int T>
__global__ __launch_bounds__(T) void k_stereo_rows(const ovs_keypoint* __restrict__ kr, const int* __restrict__ n_ptr, int n_fixed,
                                                     stereo_images im, u32* __restrict__ row_start, u32* __restrict__ row_items, int items_cap,
                                                     int lds_items, int* __restrict__ err) {
    extern __shared__ u32 sm[];
    __shared__ u32 red[T / 64];
    const int pair = blockIdx.y;
    const int n = n_ptr ? min(max(n_ptr[pair], 0), n_fixed) : n_fixed;
    const int rows = im.lh[0];
    kr += pair * im.kp_stride;
    row_start += pair * im.rows_stride;
    row_items += pair * im.items_stride;
    u32* cnt = sm;            // rows + 1
    u32* cur = sm + rows + 1; // rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i <= rows; i += T) {
        cnt[i] = 0;
    }
    __syncthreads();
    for (int i = tid; i < n; i += T) {
        const ovs_keypoint kp = kr[i];
        if (kp.octave < 0 || im.num_levels <= kp.octave) {
            *err = OVS_ERR_OUT_OF_RANGE;
            continue;
        }
        const float r = __fmul_rn(2.0f, im.scale[kp.octave]);
        const int max_r = cv_ceil_ff(__fadd_rn(kp.y, r));
        const int min_r = cv_floor_ff(__fsub_rn(kp.y, r));
        if (min_r < 0 || rows <= max_r) {
            *err = OVS_ERR_OUT_OF_RANGE; // indices_right_in_row.at(row_right) throws
        }
        for (int row = max(min_r, 0); row <= min(max_r, rows - 1); ++row) {
            atomicAdd(&cnt[row], 1u);
        }
    }
    __syncthreads();
    {
        const int per = (rows + 1 + T - 1) / T;
        const int b = min(tid * per, rows + 1), e = min(b + per, rows + 1);
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
    const bool fits = cnt[rows] <= (u32)items_cap;
    for (int i = tid; i <= rows; i += T) {
        row_start[i] = fits ? cnt[i] : 0u;
        if (i < rows) {
            cur[i] = cnt[i];
        }
    }
    if (!fits) {
        if (tid == 0) {
            *err = OVS_ERR_CAPACITY;
        }
        return;
    }
    __syncthreads();
    // scatter the keypoint indices into their rows, then order every row's list (the reference
    // appends in keypoint order).  In global memory (lists that do not fit in LDS behind the row tables,
    // `lds_items` words): one insertion sort per row.
    auto fill_and_sort = [&](u32* items) {
        for (int i = tid; i < n; i += T) {
            const ovs_keypoint kp = kr[i];
            if (kp.octave < 0 || im.num_levels <= kp.octave) {
                continue;
            }
            const float r = __fmul_rn(2.0f, im.scale[kp.octave]);
            const int max_r = cv_ceil_ff(__fadd_rn(kp.y, r));
            const int min_r = cv_floor_ff(__fsub_rn(kp.y, r));
            for (int row = max(min_r, 0); row <= min(max_r, rows - 1); ++row) {
                items[atomicAdd(&cur[row], 1u)] = (u32)i;
            }
        }
        __syncthreads();
        for (int row = tid; row < rows; row += T) {
            const u32 b = cnt[row], e = cnt[row + 1];
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
    };
    const u32 total_items = cnt[rows];
    constexpr int kIdxBits = 19;
    if (total_items <= (u32)lds_items && n < (1 << kIdxBits)) {
        // The lists in LDS, unordered, every entry tagged with its row (row << 19 | keypoint index: rows are below
        // 8191).  An entry's place in its row is the number of entries of that row with a smaller keypoint index:
        // one thread per entry counts them and writes the index straight to its final place in global memory.
        // All reads are independent of each other (the insertion sorts of the rows were chains of dependent LDS
        // accesses: 47 of the kernel's 60 us for 1000 keypoints).
        u32* litems = sm + 2 * rows + 1;
        for (int i = tid; i < n; i += T) {
            const ovs_keypoint kp = kr[i];
            if (kp.octave < 0 || im.num_levels <= kp.octave) {
                continue;
            }
            const float r = __fmul_rn(2.0f, im.scale[kp.octave]);
            const int max_r = cv_ceil_ff(__fadd_rn(kp.y, r));
            const int min_r = cv_floor_ff(__fsub_rn(kp.y, r));
            for (int row = max(min_r, 0); row <= min(max_r, rows - 1); ++row) {
                litems[atomicAdd(&cur[row], 1u)] = ((u32)row << kIdxBits) | (u32)i;
            }
        }
        __syncthreads();
        for (u32 p = tid; p < total_items; p += T) {
            const u32 w = litems[p];
            const u32 row = w >> kIdxBits;
            const u32 b = cnt[row], e = cnt[row + 1];
            u32 rank = 0;
            for (u32 q = b; q < e; ++q) {
                rank += litems[q] < w;
            }
            row_items[b + rank] = w & ((1u << kIdxBits) - 1u);
        }
    }
    else {
        fill_and_sort(row_items);
    }
}

template <int W>
__device__ __forceinline__ int group_sum_int(int v) {
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d, W);
    }
    return v;
}

__global__ __launch_bounds__(256) void k_stereo_match(const ovs_keypoint* __restrict__ kl, const uint8_t* __restrict__ dl, const int* __restrict__ nl_ptr, int nl_fixed,
                                                      const ovs_keypoint* __restrict__ kr, const uint8_t* __restrict__ dr,
                                                      stereo_images im, const u32* __restrict__ row_start, const u32* __restrict__ row_items,
                                                      float focal_x_baseline, float max_disp,
                                                      float* __restrict__ out_x_right, float* __restrict__ out_depth, int* __restrict__ out_corr,
                                                      int* __restrict__ err) {
    const int pair = blockIdx.y;
    const int nl = nl_ptr ? min(max(nl_ptr[pair], 0), nl_fixed) : nl_fixed;
    const int i = blockIdx.x * (256 / kSG) + threadIdx.x / kSG;
    const int gl = threadIdx.x & (kSG - 1);
    if (i >= nl) {
        return;
    }
    kl += pair * im.kp_stride;
    kr += pair * im.kp_stride;
    dl += pair * im.kp_stride * 32;
    dr += pair * im.kp_stride * 32;
    row_start += pair * im.rows_stride;
    row_items += pair * im.items_stride;
    out_x_right += pair * im.kp_stride;
    out_depth += pair * im.kp_stride;
    out_corr += pair * im.kp_stride;
    if (gl == 0) {
        out_x_right[i] = -1.0f;
        out_depth[i] = -1.0f;
        out_corr[i] = -1;
    }
    const ovs_keypoint kp = kl[i];
    const int oct = kp.octave;
    if (oct < 0 || im.num_levels <= oct) {
        if (gl == 0) {
            *err = OVS_ERR_OUT_OF_RANGE;
        }
        return;
    }
    const float y_left = kp.y, x_left = kp.x;
    // indices_right_in_row.at(y_left): float -> size_t truncation, stereo.cc:38
    const long long row = (long long)y_left;
    if (!(y_left >= 0.0f) || row >= im.lh[0]) {
        if (gl == 0) {
            *err = OVS_ERR_OUT_OF_RANGE;
        }
        return;
    }
    const u32 b = row_start[row], e = row_start[row + 1];
    if (b == e) {
        return;
    }
    const float min_x_right = __fsub_rn(x_left, max_disp);
    const float max_x_right = x_left; // min_disp_ = 0
    if (max_x_right < 0) {
        return;
    }
    // ---- find_closest_keypoints_in_stereo ----
    u32 qd[8];
    load_desc4(dl + 32 * (size_t)i, qd);
    u64 best = kNoKey;
    for (u32 j = b + gl; j < e; j += kSG) {
        const u32 idx = row_items[j];
        const ovs_keypoint rk = kr[idx];
        if (rk.octave < oct - 1 || rk.octave > oct + 1) {
            continue;
        }
        if (rk.x < min_x_right || max_x_right < rk.x) {
            continue;
        }
        u32 cd[8];
        load_desc4(dr + 32 * (size_t)idx, cd);
        const u64 key = ((u64)hamming8(qd, cd) << 40) | ((u64)(j - b) << 20) | idx;
        best = key < best ? key : best;
    }
    best = group_min_u64<kSG>(best);
    if (best == kNoKey || kStereoThr <= (u32)(best >> 40)) {
        return;
    }
    const u32 best_idx_right = (u32)(best & 0xFFFFFu);

    // ---- compute_subpixel_disparity ----
    const float x_right = kr[best_idx_right].x;
    const float inv_scale_factor = im.inv_scale[oct];
    const int scaled_x_left = __float2int_rn(__fmul_rn(kp.x, inv_scale_factor));
    const int scaled_y_left = __float2int_rn(__fmul_rn(kp.y, inv_scale_factor));
    const int scaled_x_right = __float2int_rn(__fmul_rn(x_right, inv_scale_factor));
    constexpr int win_size = 5, slide_width = 5;
    const int ini_x = scaled_x_right + slide_width - win_size;
    const int end_x = scaled_x_right + slide_width + win_size + 1;
    if (ini_x < 0 || im.rw[oct] <= end_x) {
        return;
    }
    if (scaled_y_left - win_size < 0 || im.lh[oct] < scaled_y_left + win_size + 1 || scaled_x_left - win_size < 0
        || im.lw[oct] < scaled_x_left + win_size + 1 || im.rh[oct] < scaled_y_left + win_size + 1) {
        if (gl == 0) {
            *err = OVS_ERR_OUT_OF_RANGE; // rowRange / colRange would assert
        }
        return;
    }
    const uint8_t* L = im.l[oct] + pair * im.lfs[oct] + (long long)(scaled_y_left - win_size) * im.ls[oct] + (scaled_x_left - win_size);
    const uint8_t* R = im.r[oct] + pair * im.rfs[oct] + (long long)(scaled_y_left - win_size) * im.rs[oct] + (scaled_x_right - win_size);
    const int lc = L[win_size * im.ls[oct] + win_size];
    // every lane owns up to 8 of the 121 patch pixels
    int lv[8], pr[8], pc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int t = gl + kSG * k;
        pr[k] = t / 11;
        pc[k] = t - pr[k] * 11;
        lv[k] = t < 121 ? (int)L[pr[k] * im.ls[oct] + pc[k]] - lc : 0;
    }
    int corr[11];
#pragma unroll
    for (int o = 0; o < 11; ++o) {
        const int offset = o - slide_width;
        const int rc = R[win_size * im.rs[oct] + win_size + offset];
        int s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = gl + kSG * k;
            if (t < 121) {
                const int rv = (int)R[pr[k] * im.rs[oct] + pc[k] + offset] - rc;
                const int d = lv[k] - rv;
                s += d < 0 ? -d : d;
            }
        }
        corr[o] = group_sum_int<kSG>(s);
    }
    float best_correlation = 4294967296.0f; // (float)UINT_MAX
    int best_offset = 0;
#pragma unroll
    for (int o = 0; o < 11; ++o) {
        const float c = (float)corr[o];
        if (c < best_correlation) {
            best_correlation = c;
            best_offset = o - slide_width;
        }
    }
    if (best_offset == -slide_width || best_offset == slide_width) {
        return;
    }
    float c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (int o = 1; o < 10; ++o) {
        if (o - slide_width == best_offset) {
            c1 = (float)corr[o - 1];
            c2 = (float)corr[o];
            c3 = (float)corr[o + 1];
        }
    }
    const double num = (double)__fsub_rn(c1, c3);
    const double den = __dsub_rn(__dmul_rn(2.0, (double)__fadd_rn(c1, c3)), __dmul_rn(4.0, (double)c2));
    const float x_delta = (float)__ddiv_rn(num, den);
    if ((double)x_delta < -1.0 || 1.0 < (double)x_delta) {
        return;
    }
    float best_x_right = __fmul_rn(im.scale[oct], __fadd_rn((float)(scaled_x_right + best_offset), x_delta));
    float best_disp = __fsub_rn(kp.x, best_x_right);
    // stereo.cc:71-80
    if (best_disp < 0.0f || max_disp <= best_disp) {
        return;
    }
    if (best_disp <= 0.0f) {
        best_disp = 0.01f;
        best_x_right = __fsub_rn(x_left, best_disp);
    }
    if (gl == 0) {
        out_depth[i] = __fdiv_rn(focal_x_baseline, best_disp);
        out_x_right[i] = best_x_right;
        out_corr[i] = (int)best_correlation;
    }
}

__global__ __launch_bounds__(256) void k_stereo_median(const int* __restrict__ nl_ptr, int nl_fixed, long long kp_stride, const int* __restrict__ corr,
                                                       float* __restrict__ x_right, float* __restrict__ depth) {
    __shared__ u32 hist[256];
    __shared__ u32 sel[4];
    const int pair = blockIdx.y;
    const int nl = nl_ptr ? min(max(nl_ptr[pair], 0), nl_fixed) : nl_fixed;
    const int tid = threadIdx.x;
    corr += pair * kp_stride;
    x_right += pair * kp_stride;
    depth += pair * kp_stride;
    // number of accepted matches
    hist[tid] = 0;
    __syncthreads();
    u32 mycount = 0;
    for (int i = tid; i < nl; i += 256) {
        const int c = corr[i];
        if (0 <= c) {
            ++mycount;
            atomicAdd(&hist[(c >> 16) & 0xFF], 1u);
        }
    }
    if (tid == 0) {
        sel[0] = 0;
    }
    __syncthreads();
    atomicAdd(&sel[0], mycount);
    __syncthreads();
    const u32 count = sel[0];
    if (count == 0) {
        return; // median_correlation = 0, nothing to reject
    }
    // k-th smallest (k = count / 2, zero based) by three byte-wise radix passes (SAD < 2^24)
    u32 k = count / 2, prefix = 0;
#pragma unroll 1
    for (int shift = 16; shift >= 0; shift -= 8) {
        if (shift != 16) {
            hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < nl; i += 256) {
                const int c = corr[i];
                if (0 <= c && ((u32)c >> (shift + 8)) == prefix) {
                    atomicAdd(&hist[((u32)c >> shift) & 0xFF], 1u);
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            u32 run = 0, b = 0;
            for (; b < 256; ++b) {
                if (k < run + hist[b]) {
                    break;
                }
                run += hist[b];
            }
            sel[1] = b;
            sel[2] = k - run;
        }
        __syncthreads();
        prefix = (prefix << 8) | sel[1];
        k = sel[2];
        __syncthreads();
    }
    const float median_correlation = (float)(int)prefix;
    const float correlation_thr = (float)(2.0 * (double)median_correlation);
    for (int i = tid; i < nl; i += 256) {
        const int c = corr[i];
        if (0 <= c && correlation_thr < (float)c) {
            x_right[i] = -1.0f;
            depth[i] = -1.0f;
        }
    }
}

int launch_stereo(const stereo_images& im, const ovs_keypoint* d_kl, const uint8_t* d_dl, const int* d_nl, int nl,
                  const ovs_keypoint* d_kr, const uint8_t* d_dr, const int* d_nr, int nr, int cap_l,
                  float focal_x_baseline, float true_baseline,
                  u32* d_row_start, u32* d_row_items, int items_cap, int* d_corr, int* d_err,
                  float* d_x_right, float* d_depth, hipStream_t st, int n_pairs = 1) {
    const int rows = im.lh[0];
    const size_t lds_tables = (2 * static_cast<size_t>(rows) + 1) * sizeof(u32);
    if (lds_tables > 64 * 1024) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "images taller than 8190 rows are not supported by the stereo matcher");
    }
    // the rest of 60 KB holds the row lists while they are built
    const int lds_items = lds_tables < 60 * 1024 ? static_cast<int>(std::min<size_t>((60 * 1024 - lds_tables) / sizeof(u32), static_cast<size_t>(std::max(items_cap, 0)))) : 0;
    const size_t lds = lds_tables + static_cast<size_t>(lds_items) * sizeof(u32);
    const float max_disp = focal_x_baseline / true_baseline; // stereo.cc:16
    if (n_pairs <= 8) {
        OVS_LAUNCH("k_stereo_rows", k_stereo_rows<1024>, dim3(1, n_pairs), dim3(1024), lds, st, d_kr, d_nr, nr, im, d_row_start, d_row_items, items_cap, lds_items, d_err);
    }
    else {
        OVS_LAUNCH("k_stereo_rows", k_stereo_rows<256>, dim3(1, n_pairs), dim3(256), lds, st, d_kr, d_nr, nr, im, d_row_start, d_row_items, items_cap, lds_items, d_err);
    }
    if (cap_l > 0) {
        OVS_LAUNCH("k_stereo_match", k_stereo_match, dim3((cap_l + 256 / kSG - 1) / (256 / kSG), n_pairs), dim3(256), 0, st,
                   d_kl, d_dl, d_nl, nl, d_kr, d_dr, im, d_row_start, d_row_items, focal_x_baseline, max_disp, d_x_right, d_depth, d_corr, d_err);
    }
    OVS_LAUNCH("k_stereo_median", k_stereo_median, dim3(1, n_pairs), dim3(256), 0, st, d_nl, nl, im.kp_stride, d_corr, d_x_right, d_depth);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

// upper bound on the number of row-list entries: every right keypoint spans at most
// ceil(2 * 2 * max_scale) + 2 rows
int row_items_bound(int n_right, const float* scale_factors, int num_levels) {
    float smax = 1.0f;
    for (int l = 0; l < num_levels; ++l) {
        smax = std::max(smax, scale_factors[l]);
    }
    return static_cast<int>(std::min<long long>(static_cast<long long>(n_right) * (static_cast<long long>(std::ceil(4.0 * smax)) + 3), 1ll << 30));
}

} // namespace

extern "C" {

int ovs_stereo_compute(const ovs_image_view* left_pyr, const ovs_image_view* right_pyr, int num_levels,
                       const ovs_keypoint* keypts_left, int n_left,
                       const ovs_keypoint* keypts_right, int n_right,
                       const uint8_t* descs_left, const uint8_t* descs_right,
                       const float* scale_factors, const float* inv_scale_factors,
                       float focal_x_baseline, float true_baseline,
                       float* stereo_x_right, float* depths, int device) {
    if (!left_pyr || !right_pyr || num_levels <= 0 || num_levels > kMaxLevels || n_left < 0 || n_right < 0 || !scale_factors || !inv_scale_factors
        || (n_left > 0 && (!keypts_left || !descs_left || !stereo_x_right || !depths)) || (n_right > 0 && (!keypts_right || !descs_right))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    for (int l = 0; l < num_levels; ++l) {
        if (!left_pyr[l].data || !right_pyr[l].data || left_pyr[l].width <= 0 || left_pyr[l].height <= 0 || right_pyr[l].width <= 0
            || right_pyr[l].height <= 0) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "empty pyramid level");
        }
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    hipStream_t st = ctx->stream;
    const size_t nl = n_left, nr = n_right;
    const int items_cap = row_items_bound(n_right, scale_factors, num_levels);
    const int rows = left_pyr[0].height;
    size_t total = al(nl * sizeof(ovs_keypoint)) + al(nr * sizeof(ovs_keypoint)) + al(nl * 32) + al(nr * 32)
                   + al((static_cast<size_t>(rows) + 1) * 4) + al(static_cast<size_t>(items_cap) * 4) + al(nl * 4) * 3 + al(256);
    for (int l = 0; l < num_levels; ++l) {
        total += al(static_cast<size_t>(left_pyr[l].width) * left_pyr[l].height) + al(static_cast<size_t>(right_pyr[l].width) * right_pyr[l].height);
    }
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    if ((rc = ctx->mirror.reserve(ctx->arena.bytes)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    stereo_images im;
    std::memset(&im, 0, sizeof(im));
    im.num_levels = num_levels;
    for (int l = 0; l < num_levels; ++l) {
        const ovs_image_view* views[2] = {&left_pyr[l], &right_pyr[l]};
        for (int side = 0; side < 2; ++side) {
            const ovs_image_view& v = *views[side];
            auto* d = cv.take<uint8_t>(static_cast<size_t>(v.width) * v.height);
            OVS_HIP(image_up(ctx, st, d, v.data, v.stride, static_cast<size_t>(v.width), v.height));
            if (side == 0) {
                im.l[l] = d;
                im.ls[l] = v.width;
                im.lw[l] = v.width;
                im.lh[l] = v.height;
            }
            else {
                im.r[l] = d;
                im.rs[l] = v.width;
                im.rw[l] = v.width;
                im.rh[l] = v.height;
            }
        }
        im.scale[l] = scale_factors[l];
        im.inv_scale[l] = inv_scale_factors[l];
    }
    auto* d_kl = cv.take<ovs_keypoint>(nl);
    auto* d_kr = cv.take<ovs_keypoint>(nr);
    auto* d_dl = cv.take<uint8_t>(nl * 32);
    auto* d_dr = cv.take<uint8_t>(nr * 32);
    auto* d_row_start = cv.take<u32>(static_cast<size_t>(rows) + 1);
    auto* d_row_items = cv.take<u32>(items_cap);
    auto* d_corr = cv.take<int>(nl);
    auto* d_xr = cv.take<float>(nl);
    auto* d_depth = cv.take<float>(nl);
    auto* d_err = cv.take<int>(1);
    if (nl) {
        OVS_HIP(hipMemcpyAsync(d_kl, keypts_left, nl * sizeof(ovs_keypoint), hipMemcpyHostToDevice, st));
        OVS_HIP(hipMemcpyAsync(d_dl, descs_left, nl * 32, hipMemcpyHostToDevice, st));
    }
    if (nr) {
        OVS_HIP(hipMemcpyAsync(d_kr, keypts_right, nr * sizeof(ovs_keypoint), hipMemcpyHostToDevice, st));
        OVS_HIP(hipMemcpyAsync(d_dr, descs_right, nr * 32, hipMemcpyHostToDevice, st));
    }
    OVS_HIP(hipMemsetAsync(d_err, 0, 4, st));
    rc = launch_stereo(im, d_kl, d_dl, nullptr, n_left, d_kr, d_dr, nullptr, n_right, n_left, focal_x_baseline, true_baseline,
                       d_row_start, d_row_items, items_cap, d_corr, d_err, d_xr, d_depth, st);
    if (rc != OVS_OK) {
        return rc;
    }
    int err = 0;
    if (nl) {
        OVS_HIP(hipMemcpyAsync(stereo_x_right, d_xr, nl * 4, hipMemcpyDeviceToHost, st));
        OVS_HIP(hipMemcpyAsync(depths, d_depth, nl * 4, hipMemcpyDeviceToHost, st));
    }
    OVS_HIP(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, st));
    OVS_HIP(hipStreamSynchronize(st));
    if (err != 0) {
        return ovs::fail(err, "keypoint row, octave or patch outside of the image");
    }
    return OVS_OK;
}

int ovs_stereo_compute_dev(ovs_orb* left, int frame_left, ovs_orb* right, int frame_right,
                           const ovs_keypoint* d_kps_left, const uint8_t* d_desc_left, const int32_t* d_n_left,
                           const ovs_keypoint* d_kps_right, const uint8_t* d_desc_right, const int32_t* d_n_right, int cap,
                           float focal_x_baseline, float true_baseline,
                           float* d_stereo_x_right, float* d_depths, int32_t* d_status, int device, void* stream_) {
    if (!left || !right || !d_kps_left || !d_desc_left || !d_n_left || !d_kps_right || !d_desc_right || !d_n_right || cap <= 0
        || !d_stereo_x_right || !d_depths) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    const int num_levels = static_cast<int>(ovs_orb_get_num_scale_levels(left));
    if (num_levels > kMaxLevels || static_cast<int>(ovs_orb_get_num_scale_levels(right)) != num_levels) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "extractors disagree on the number of levels");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    stereo_images im;
    std::memset(&im, 0, sizeof(im));
    im.num_levels = num_levels;
    float sf[kMaxLevels], isf[kMaxLevels];
    ovs_orb_get_scale_factors(left, sf);
    ovs_orb_get_inv_scale_factors(left, isf);
    for (int l = 0; l < num_levels; ++l) {
        size_t stride = 0;
        if ((rc = ovs_orb_pyramid_level_dev(left, frame_left, l, &im.l[l], &im.lw[l], &im.lh[l], &stride)) != OVS_OK) {
            return rc;
        }
        im.ls[l] = static_cast<long long>(stride);
        if ((rc = ovs_orb_pyramid_level_dev(right, frame_right, l, &im.r[l], &im.rw[l], &im.rh[l], &stride)) != OVS_OK) {
            return rc;
        }
        im.rs[l] = static_cast<long long>(stride);
        im.scale[l] = sf[l];
        im.inv_scale[l] = isf[l];
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    thread_ctx* ctx = ctx_for_stream(device, st);
    const int items_cap = row_items_bound(cap, sf, num_levels);
    const int rows = im.lh[0];
    const size_t total = al((static_cast<size_t>(rows) + 1) * 4) + al(static_cast<size_t>(items_cap) * 4) + al(static_cast<size_t>(cap) * 4) + al(256);
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_row_start = cv.take<u32>(static_cast<size_t>(rows) + 1);
    auto* d_row_items = cv.take<u32>(items_cap);
    auto* d_corr = cv.take<int>(cap);
    // the kernels' error word: the caller's, or (d_status == NULL: the caller does not want to know) a scratch word
    int* d_err = d_status ? d_status : cv.take<int>(1);
    OVS_HIP(hipMemsetAsync(d_err, 0, 4, st));
    return launch_stereo(im, d_kps_left, d_desc_left, d_n_left, cap, d_kps_right, d_desc_right, d_n_right, cap, cap, focal_x_baseline, true_baseline,
                         d_row_start, d_row_items, items_cap, d_corr, d_err, d_stereo_x_right, d_depths, st);
}

int ovs_stereo_compute_resident(ovs_orb* left, ovs_orb* right,
                                const ovs_keypoint* keypts_left, int n_left, const ovs_keypoint* keypts_right, int n_right,
                                const uint8_t* descs_left, const uint8_t* descs_right,
                                float focal_x_baseline, float true_baseline,
                                float* stereo_x_right, float* depths, int device) {
    if (!left || !right || n_left < 0 || n_right < 0 || (n_left > 0 && (!keypts_left || !descs_left || !stereo_x_right || !depths))
        || (n_right > 0 && (!keypts_right || !descs_right))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    hipStream_t st = ctx->stream;
    // keypoints, descriptors and results are staged in this thread's arena; the device entry point
    // below takes its scratch from the arena that belongs to (device, st), a different one
    const size_t nl = n_left, nr = n_right;
    const int cap = std::max(std::max(n_left, n_right), 1);
    const size_t total = al(nl * sizeof(ovs_keypoint)) + al(nr * sizeof(ovs_keypoint)) + al(nl * 32) + al(nr * 32) + al(static_cast<size_t>(cap) * 4) * 2
                         + al(8) + al(4);
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_kl = cv.take<ovs_keypoint>(nl);
    auto* d_kr = cv.take<ovs_keypoint>(nr);
    auto* d_dl = cv.take<uint8_t>(nl * 32);
    auto* d_dr = cv.take<uint8_t>(nr * 32);
    auto* d_xr = cv.take<float>(cap);
    auto* d_depth = cv.take<float>(cap);
    auto* d_n = cv.take<int32_t>(2);
    auto* d_status = cv.take<int32_t>(1);
    const int32_t counts[2] = {n_left, n_right};
    host_stager hs;
    if ((rc = hs.init(ctx, st)) != OVS_OK) {
        return rc;
    }
    hs.put(d_kl, keypts_left, nl * sizeof(ovs_keypoint));
    hs.put(d_kr, keypts_right, nr * sizeof(ovs_keypoint));
    hs.put(d_dl, descs_left, nl * 32);
    hs.put(d_dr, descs_right, nr * 32);
    hs.put(d_n, counts, sizeof(counts));
    OVS_HIP(hs.upload());
    // the pyramids were produced by ovs_orb_extract, which is synchronous: they are complete
    if ((rc = ovs_stereo_compute_batch_dev(left, right, 1, d_kl, d_dl, d_n, d_kr, d_dr, d_n + 1, cap, focal_x_baseline, true_baseline, d_xr, d_depth,
                                           d_status, device, st)) != OVS_OK) {
        return rc;
    }
    int32_t status = 0;
    hs.get(stereo_x_right, d_xr, nl * 4);
    hs.get(depths, d_depth, nl * 4);
    hs.get(&status, d_status, 4);
    OVS_HIP(hs.download());
    if (status != 0) {
        return ovs::fail(status, "keypoint row, octave or patch outside of the image");
    }
    return OVS_OK;
}

int ovs_stereo_compute_batch_dev(ovs_orb* left, ovs_orb* right, int n_pairs,
                                 const ovs_keypoint* d_kps_left, const uint8_t* d_desc_left, const int32_t* d_counts_left,
                                 const ovs_keypoint* d_kps_right, const uint8_t* d_desc_right, const int32_t* d_counts_right, int cap,
                                 float focal_x_baseline, float true_baseline,
                                 float* d_stereo_x_right, float* d_depths, int32_t* d_status, int device, void* stream_) {
    if (!left || !right || n_pairs <= 0 || n_pairs > 65535 || !d_kps_left || !d_desc_left || !d_counts_left || !d_kps_right || !d_desc_right
        || !d_counts_right || cap <= 0 || !d_stereo_x_right || !d_depths || !d_status) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    const int num_levels = static_cast<int>(ovs_orb_get_num_scale_levels(left));
    if (num_levels > kMaxLevels || static_cast<int>(ovs_orb_get_num_scale_levels(right)) != num_levels) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "extractors disagree on the number of levels");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    stereo_images im;
    std::memset(&im, 0, sizeof(im));
    im.num_levels = num_levels;
    float sf[kMaxLevels], isf[kMaxLevels];
    ovs_orb_get_scale_factors(left, sf);
    ovs_orb_get_inv_scale_factors(left, isf);
    for (int l = 0; l < num_levels; ++l) {
        // frames of one level lie a constant number of bytes apart: take it from the first and the
        // last pair (which also checks that both handles hold at least n_pairs frames)
        const uint8_t* last = nullptr;
        size_t stride = 0;
        int w = 0, h = 0;
        if ((rc = ovs_orb_pyramid_level_dev(left, 0, l, &im.l[l], &im.lw[l], &im.lh[l], &stride)) != OVS_OK
            || (rc = ovs_orb_pyramid_level_dev(left, n_pairs - 1, l, &last, &w, &h, &stride)) != OVS_OK) {
            return rc;
        }
        im.ls[l] = static_cast<long long>(stride);
        im.lfs[l] = n_pairs > 1 ? (last - im.l[l]) / (n_pairs - 1) : 0;
        if ((rc = ovs_orb_pyramid_level_dev(right, 0, l, &im.r[l], &im.rw[l], &im.rh[l], &stride)) != OVS_OK
            || (rc = ovs_orb_pyramid_level_dev(right, n_pairs - 1, l, &last, &w, &h, &stride)) != OVS_OK) {
            return rc;
        }
        im.rs[l] = static_cast<long long>(stride);
        im.rfs[l] = n_pairs > 1 ? (last - im.r[l]) / (n_pairs - 1) : 0;
        im.scale[l] = sf[l];
        im.inv_scale[l] = isf[l];
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    thread_ctx* ctx = ctx_for_stream(device, st);
    const int items_cap = row_items_bound(cap, sf, num_levels);
    const int rows = im.lh[0];
    im.kp_stride = cap;
    im.rows_stride = static_cast<long long>(al((static_cast<size_t>(rows) + 1) * 4) / 4);
    im.items_stride = static_cast<long long>(al(static_cast<size_t>(items_cap) * 4) / 4);
    const size_t np = static_cast<size_t>(n_pairs);
    const size_t total = np * (static_cast<size_t>(im.rows_stride) * 4 + static_cast<size_t>(im.items_stride) * 4) + al(np * cap * 4) + al(256);
    if ((rc = ctx->arena.reserve(total)) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_row_start = cv.take<u32>(np * im.rows_stride);
    auto* d_row_items = cv.take<u32>(np * im.items_stride);
    auto* d_corr = cv.take<int>(np * cap);
    OVS_HIP(hipMemsetAsync(d_status, 0, 4, st));
    return launch_stereo(im, d_kps_left, d_desc_left, d_counts_left, cap, d_kps_right, d_desc_right, d_counts_right, cap, cap, focal_x_baseline,
                         true_baseline, d_row_start, d_row_items, items_cap, d_corr, d_status, d_stereo_x_right, d_depths, st, n_pairs);
}

} // extern "C"
