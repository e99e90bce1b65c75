// ovs_orb.hip -- ORB extractor for gfx950: image pyramid, per-cell FAST-9/16 + NMS, quadtree
// keypoint distribution, intensity-centroid orientation, 7x7 Gaussian blur and rotated-BRIEF
// descriptors, batched over frames.  Replaces feature::orb_extractor::extract()
// (src/openvslam/feature/orb_extractor.cc:68-144) and everything it calls.
//
// Pipeline per batch (no host synchronisation in between):
//   k_resize_linear   x (levels-1)   level k from level k-1        (orb_extractor.cc:263-272)
//   k_fast_cells      x 1            one workgroup per 64-px cell  (orb_extractor.cc:274-369)
//   k_gauss7          x 1            all levels, all frames, beside FAST + quadtree on the
//                                    handle's second stream        (orb_extractor.cc:132-133)
//   k_quadtree        x 1            one workgroup per (frame, level) (orb_extractor.cc:398-590)
//   k_orient_desc     x 1            16 lanes per keypoint         (orb_extractor.cc:592-702)
//
// Data layout in HBM (per handle, sized for the largest batch seen):
//   pyramid   levels 1..L-1, [level][frame][h_l][pitch_l] u8, pitch_l = w_l rounded up to 64
//             (level 0 is the caller's image and is read in place, as orb_extractor.cc:264 does)
//   blurred   levels 0..L-1, same layout
//   slots     [frame][cell][1024] u32  FAST survivors of one cell in raster order, packed
//             x | y << 12 | score << 24 in border-relative level coordinates; 1024 is the exact
//             maximum 3x3 non-maximum suppression leaves in a 64 x 64 window
//   cand      [2][frame][level][cells_l * 1024] dense candidates in the reference's order (positions, node ids)
//             (cells row-major, raster inside a cell): xy and (response << 24 | quadtree node)
//   lvl_*     [frame][sum ncap_l] per-level keypoints in quadtree list order + [frame][L] counts
#include "ovs_internal.h"
#include "orb_pattern.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <vector>

namespace {

constexpr int kMaxLevels = 16;
constexpr int kBorder = 19;     // orb_patch_radius_, orb_extractor.h:133
constexpr int kCell = 64;       // cell_size, orb_extractor.cc:283
constexpr int kOverlap = 6;     // overlap, orb_extractor.cc:282
constexpr int kSlotsPerCell = 1024;
constexpr int kHalfPatch = 15;  // fast_half_patch_size_, orb_extractor.h:130
constexpr int kQtThreads = 512;      // threads per quadtree workgroup in batches
constexpr int kQtThreadsLive = 1024; // for calls with a few frames (a level-0 workgroup: 61.6 -> 55.3 us)

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// Workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive block ids
// land on different L2s.  The streaming kernels share cache lines between neighbouring tiles /
// cells; they are launched as a 1-D grid of 8 * ceil(total / 8) blocks and every XCD walks one
// contiguous eighth of the logical (frame, tile) space.  Returns -1 for the padding blocks.
__device__ __forceinline__ int xcd_contiguous_block(int total) {
    const int per = (total + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
    return logical < total ? logical : -1;
}

struct level_geom {
    int w, h;
    int pitch;            // row pitch of the library-owned planes of this level
    long long pyr_off;    // byte offset of the level inside one frame's pyramid block (levels >= 1)
    long long blur_off;   // byte offset inside one frame's blurred block
    int cells_x, cells_y; // num_cols, num_rows of orb_extractor.cc:299-300
    int cell_base;        // first flattened cell index of the level
    int quota;            // num_keypts_per_level_
    int ncap;             // quadtree node capacity
    int kp_base;          // first slot of the level in the per-frame level-keypoint arrays
    long long cand_base;  // first dense candidate slot of the level inside a frame
    int nx, ny;           // initial node grid, orb_extractor.cc:486-499
    double delta_x, delta_y;
    float scale;          // scale_factors_[level]
    float patch_size;     // (float)(unsigned)(31 * scale), orb_extractor.cc:380,388
    int tab_off;          // offset of the resize tables of this level in the table buffer
    int area2x;           // cv::resize switches to INTER_AREA for an exact 2x reduction
    int resize_rows;      // tables fit the row-streaming resize kernel
    int blur_tile_base, blur_tiles_x, blur_tiles_y;
};

struct plan_dev {
    int num_levels;
    int total_cells;
    int total_blur_tiles;
    int kp_total_cap;
    int max_cells;
    int max_ncap;
    long long frame_pyr_bytes;
    long long frame_blur_bytes;
    long long frame_cand_cap;
    int ini_thr, min_thr;
    level_geom lv[kMaxLevels];
};

struct image_src {
    const uint8_t* img0;  // level 0, caller-owned
    long long row_stride0, frame_stride0;
    uint8_t* pyr;         // levels >= 1
    const uint8_t* mask;  // level-0 mask or nullptr
    long long mask_stride;
    int mask_w, mask_h;
};

__device__ __forceinline__ const uint8_t* level_ptr(const plan_dev& p, const image_src& s, int frame, int level, long long& pitch) {
    if (level == 0) {
        pitch = s.row_stride0;
        return s.img0 + frame * s.frame_stride0;
    }
    pitch = p.lv[level].pitch;
    return s.pyr + frame * p.frame_pyr_bytes + p.lv[level].pyr_off;
}

// -------------------------------------------------------------------------------------------------
// Pyramid: cv::resize(INTER_LINEAR) for 8-bit single channel, fixed point with 11 coefficient bits.
// Horizontal pass in int32 (S[sx] * a0 + S[sx + 1] * a1), vertical pass
// (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  Tables (source offsets and
// coefficient pairs) are built on the host exactly as OpenCV builds xofs / ialpha / yofs / ibeta.
// One thread produces four horizontally adjacent pixels and stores them as one dword.
// -------------------------------------------------------------------------------------------------
// Four bytes from an arbitrarily aligned address using aligned dword loads only: the dword at the
// address rounded down to 4 and, when the address is not aligned, the next one (which then holds
// at least the last requested byte, so it is backed by memory whenever the four bytes are).
__device__ __forceinline__ u32 load_u32_unaligned(const uint8_t* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const u32* q = reinterpret_cast<const u32*>(a & ~static_cast<uintptr_t>(3));
    const u32 sh = (u32)(a & 3);
    const u32 x0 = q[0];
    const u32 x1 = sh ? q[1] : 0u;
    return __builtin_amdgcn_alignbyte(x1, x0, sh);
}

// 8 bytes from an arbitrarily aligned address using aligned dword loads only (reads the 12 bytes
// starting at the address rounded down to 4)
__device__ __forceinline__ u64 load_8_bytes(const uint8_t* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const u32* q = reinterpret_cast<const u32*>(a & ~static_cast<uintptr_t>(3));
    const u32 sh = (u32)(a & 3);
    const u32 x0 = q[0], x1 = q[1], x2 = q[2];
    const u32 lo = __builtin_amdgcn_alignbyte(x1, x0, sh);
    const u32 hi = __builtin_amdgcn_alignbyte(x2, x1, sh);
    return ((u64)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void k_resize_linear(plan_dev p, image_src s, const int* __restrict__ tabs, int level) {
    const level_geom& g = p.lv[level];
    const level_geom& gs = p.lv[level - 1];
    const int frame = blockIdx.z;
    const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    if (dx0 >= g.w || dy >= g.h) {
        return;
    }
    long long spitch;
    const uint8_t* src = level_ptr(p, s, frame, level - 1, spitch);
    uint8_t* dst = s.pyr + frame * p.frame_pyr_bytes + g.pyr_off;

    u32 packed = 0;
    if (g.area2x) {
        const uint8_t* r0 = src + (long long)(2 * dy) * spitch;
        const uint8_t* r1 = r0 + spitch;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dx = dx0 + k;
            if (dx < g.w) {
                const int v = (r0[2 * dx] + r0[2 * dx + 1] + r1[2 * dx] + r1[2 * dx + 1] + 2) >> 2;
                packed |= (u32)v << (8 * k);
            }
        }
    }
    else {
        // tables: xofs[w4], xab[w4], yofs[h], yab[h] with w4 = w rounded up to 4 (16-byte aligned rows)
        const int w4 = (g.w + 3) & ~3;
        const int* xofs = tabs + g.tab_off;
        const int* xab = xofs + w4;
        const int* yofs = xab + w4;
        const int* yab = yofs + g.h;
        const int sy = yofs[dy];
        const int sy0 = min(max(sy, 0), gs.h - 1);
        const int sy1 = min(max(sy + 1, 0), gs.h - 1);
        const int bb = yab[dy];
        const int b0 = (short)(bb & 0xFFFF), b1 = (short)(bb >> 16);
        const uint8_t* r0 = src + (long long)sy0 * spitch;
        const uint8_t* r1 = src + (long long)sy1 * spitch;
        const int4 so = *reinterpret_cast<const int4*>(xofs + dx0);
        const int4 sa = *reinterpret_cast<const int4*>(xab + dx0);
        const int sx[4] = {so.x, so.y, so.z, so.w};
        const int aa[4] = {sa.x, sa.y, sa.z, sa.w};
        const int nvalid = min(g.w - dx0, 4);
        const int last = sx[nvalid - 1];
        if (last + 1 - sx[0] < 8 && sx[0] + 12 <= gs.w) {
            // the (at most) 8 source bytes of both rows: three aligned dwords per row, byte-aligned
            // into a 64-bit window with v_alignbyte
            const u64 w0 = load_8_bytes(r0 + sx[0]);
            const u64 w1 = load_8_bytes(r1 + sx[0]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < nvalid) {
                    const int o = sx[k] - sx[0];
                    const int o1 = min(sx[k] + 1, gs.w - 1) - sx[0];
                    const int a0 = (short)(aa[k] & 0xFFFF), a1 = (short)(aa[k] >> 16);
                    const int h0 = (int)((w0 >> (8 * o)) & 0xFF) * a0 + (int)((w0 >> (8 * o1)) & 0xFF) * a1;
                    const int h1 = (int)((w1 >> (8 * o)) & 0xFF) * a0 + (int)((w1 >> (8 * o1)) & 0xFF) * a1;
                    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    packed |= (u32)(v & 0xFF) << (8 * k);
                }
            }
        }
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k < nvalid) {
                    const int sx1 = min(sx[k] + 1, gs.w - 1);
                    const int a0 = (short)(aa[k] & 0xFFFF), a1 = (short)(aa[k] >> 16);
                    const int h0 = r0[sx[k]] * a0 + r0[sx1] * a1;
                    const int h1 = r1[sx[k]] * a0 + r1[sx1] * a1;
                    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
                    packed |= (u32)(v & 0xFF) << (8 * k);
                }
            }
        }
    }
    *reinterpret_cast<u32*>(dst + (long long)dy * g.pitch + dx0) = packed;
}

// The same arithmetic organised for instruction count: one wavefront owns 256 output columns
// (four per lane, one dword store) and walks down kResizeRows output rows.  It streams the source
// rows in order; each source row gets ONE row pass per lane -- an 8-byte window (one 12-byte
// buffer load, two v_alignbyte), then per pixel a v_perm that drops S[sx] and S[sx + 1] into the
// two halves of a register and a v_dot2_u32_u16 with the (a0, a1) coefficient pair -- and is used
// by the (at most two) output rows that reference it.  Selectors and coefficients are per-lane
// constants read once from the tables; row offsets and the vertical coefficients are scalars.
// Used when the tables are monotone with 1 <= sy(dy + 1) - sy(dy) <= 2 and every group of four
// output columns reads at most 8 consecutive source bytes (any scale factor in (1, 2)); other
// geometries take k_resize_linear.
#ifndef OVS_RESIZE_ROWS
#define OVS_RESIZE_ROWS 16
#endif
constexpr int kResizeRows = OVS_RESIZE_ROWS;
constexpr int kResizeRowsLive = 4; // for calls with at most kLiveFrames frames
constexpr int kLiveFrames = 8;

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

struct resize_lane {
    u32 sel[4]; // v_perm selectors: byte 0 <- S[sx], byte 2 <- S[min(sx + 1, ws - 1)], bytes 1 and 3 <- 0
    u32 ab[4];  // a0 | a1 << 16
    u32 sx0;    // first source column of the lane's window
};

template <bool UNALIGNED>
__device__ __forceinline__ u32x3 resize_row_issue(__amdgpu_buffer_rsrc_t plane, u32 row_off, const resize_lane& ln) {
    if (UNALIGNED) {
        const u32 t = (row_off & 3u) + ln.sx0;
        return __builtin_amdgcn_raw_buffer_load_b96(plane, t & ~3u, row_off & ~3u, 0);
    }
    return __builtin_amdgcn_raw_buffer_load_b96(plane, ln.sx0 & ~3u, row_off, 0);
}

// row pass: g[k] = (S[sx_k] * a0_k + S[sx_k + 1] * a1_k) >> 4
template <bool UNALIGNED>
__device__ __forceinline__ void resize_row_finish(const u32x3& d, u32 row_off, const resize_lane& ln, u32 (&g)[4]) {
    const u32 shift = UNALIGNED ? (((row_off & 3u) + ln.sx0) & 3u) : (ln.sx0 & 3u);
    const u32 lo = __builtin_amdgcn_alignbyte(d.y, d.x, shift);
    const u32 hi = __builtin_amdgcn_alignbyte(d.z, d.y, shift);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 pair = __builtin_amdgcn_perm(hi, lo, ln.sel[k]);
        g[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pair), __builtin_bit_cast(u16x2, ln.ab[k]), 0u, false) >> 4;
    }
}

// column pass and store of one output row: (((b0 * g0) >> 16) + ((b1 * g1) >> 16) + 2) >> 2
__device__ __forceinline__ void resize_emit(const u32 (&g0)[4], const u32 (&g1)[4], u32 bb, __amdgpu_buffer_rsrc_t out, u32 out_voff, u32 out_row_off, bool store_lane) {
    const u32 b0 = bb & 0xFFFFu, b1 = bb >> 16;
    u32 packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 p0 = __umul24(g0[k], b0), p1 = __umul24(g1[k], b1);
        const u32 halves = __builtin_amdgcn_perm(p1, p0, 0x07060302u); // (p0 >> 16) | (p1 >> 16) << 16
        const u32 v = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, halves), (u16x2){1, 1}, 2u, false) >> 2;
        packed |= v << (8 * k);
    }
    if (store_lane) {
        __builtin_amdgcn_raw_buffer_store_b32(packed, out, out_voff, out_row_off, 0);
    }
}

template <bool UNALIGNED>
__device__ __forceinline__ void resize_tile(const level_geom& g, const level_geom& gs, const uint8_t* src, u32 spitch, uint8_t* dst,
                                            const int* __restrict__ tabs, int y0, int rows, int cb, int lane) {
    const int w4 = (g.w + 3) & ~3;
    const int* xofs = tabs + g.tab_off;
    const int* xab = xofs + w4;
    const int* yofs = xab + w4;
    const int* yab = yofs + g.h;
    const u32 delta = UNALIGNED ? (u32)(reinterpret_cast<uintptr_t>(src) & 3) : 0u;
    const __amdgpu_buffer_rsrc_t plane = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src) - delta, 0,
                                                                           (int)((delta + spitch * (u32)(gs.h - 1) + (u32)gs.w + 3u) & ~3u), 0x00020000);
    const __amdgpu_buffer_rsrc_t out = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)((u32)g.pitch * (u32)g.h), 0x00020000);
    const int dx0 = cb + 4 * lane;
    const bool store_lane = dx0 < w4;
    const int tx = min(dx0, w4 - 4);
    const int4 so = *reinterpret_cast<const int4*>(xofs + tx);
    const int4 sa = *reinterpret_cast<const int4*>(xab + tx);
    const int sx[4] = {so.x, so.y, so.z, so.w};
    resize_lane ln;
    ln.ab[0] = (u32)sa.x; ln.ab[1] = (u32)sa.y; ln.ab[2] = (u32)sa.z; ln.ab[3] = (u32)sa.w;
    ln.sx0 = (u32)sx[0];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // table entries past the last column are zero: keep their offsets inside the window
        const int o = min(max(sx[k] - sx[0], 0), 6);
        const int o1 = min(max(min(sx[k] + 1, gs.w - 1) - sx[0], 0), 7);
        ln.sel[k] = 0x0c000c00u | (u32)o | ((u32)o1 << 16);
    }
    const int y_end = min(y0 + rows, g.h);
    const int last_src = gs.h - 1;
    int dy = y0;
    int sy_next = yofs[dy];
    int r = sy_next; // source rows r, r + 1, ... are streamed; row index clamped when fetched
    u32 ro_a = delta + (u32)min(r, last_src) * spitch, ro_b = delta + (u32)min(r + 1, last_src) * spitch;
    u32x3 raw_a = resize_row_issue<UNALIGNED>(plane, ro_a, ln);
    u32x3 raw_b = resize_row_issue<UNALIGNED>(plane, ro_b, ln);
    u32 ga[4], gb[4] = {0, 0, 0, 0};
    const u32 out_voff = (u32)dx0;
#pragma unroll 1
    for (;; r += 2) {
        resize_row_finish<UNALIGNED>(raw_a, ro_a, ln, ga);
        ro_a = delta + (u32)min(r + 2, last_src) * spitch;
        raw_a = resize_row_issue<UNALIGNED>(plane, ro_a, ln);
        if (sy_next + 1 == r) {
            resize_emit(gb, ga, (u32)yab[dy], out, out_voff, (u32)dy * (u32)g.pitch, store_lane);
            if (++dy == y_end) {
                break;
            }
            sy_next = yofs[dy];
        }
        resize_row_finish<UNALIGNED>(raw_b, ro_b, ln, gb);
        ro_b = delta + (u32)min(r + 3, last_src) * spitch;
        raw_b = resize_row_issue<UNALIGNED>(plane, ro_b, ln);
        if (sy_next == r) {
            resize_emit(ga, gb, (u32)yab[dy], out, out_voff, (u32)dy * (u32)g.pitch, store_lane);
            if (++dy == y_end) {
                break;
            }
            sy_next = yofs[dy];
        }
    }
}

// `rows`: output rows per wave tile (kResizeRows in batches; fewer for a call with a few frames, where a level is a
// handful of tiles and the walk down a tile is what the next level's launch waits for).
__global__ __launch_bounds__(256) void k_resize_rows(plan_dev p, image_src s, const int* __restrict__ tabs, int level, int tiles_x, int tiles, int n_frames,
                                                     int rows) {
    const level_geom& g = p.lv[level];
    const level_geom& gs = p.lv[level - 1];
    const int groups = (tiles + 3) >> 2; // four wave tiles per workgroup
    const int blk = xcd_contiguous_block(groups * n_frames);
    if (blk < 0) {
        return;
    }
    const int frame = blk / groups;
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane((blk - frame * groups) * 4 + (threadIdx.x >> 6));
    if (t >= tiles) {
        return;
    }
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    long long spitch;
    const uint8_t* src = level_ptr(p, s, frame, level - 1, spitch);
    uint8_t* dst = s.pyr + frame * p.frame_pyr_bytes + g.pyr_off;
    if (level == 1 && (((reinterpret_cast<uintptr_t>(src) | (uintptr_t)spitch) & 3) != 0)) {
        resize_tile<true>(g, gs, src, (u32)spitch, dst, tabs, ty * rows, rows, tx * 256, lane);
    }
    else {
        resize_tile<false>(g, gs, src, (u32)spitch, dst, tabs, ty * rows, rows, tx * 256, lane);
    }
}

// -------------------------------------------------------------------------------------------------
// FAST-9/16 with non-maximum suppression, one workgroup per 64-px cell (window <= 70 x 70 with the
// 6-px overlap, so the 64 x 64 interiors tile the bordered level exactly).
//
// For a ring x[0..15] around centre v, let lo = min over the 16 arcs of 9 contiguous pixels of the
// arc maximum and hi = max over the arcs of the arc minimum.  A = max(v - lo, hi - v) is the
// largest t + 1 for which the pixel is still a FAST-9 corner at threshold t, hence
//   corner at threshold t  <=>  A - 1 >= t,   and OpenCV's cornerScore of a corner is A - 1
// independent of the threshold it was detected with.  A pixel is kept iff S = A - 1 >= t and
// S > S' for all 8 neighbours with S' >= t.  The kernel runs this for the initial threshold and,
// when the cell stays empty, once more for the minimum one (orb_extractor.cc:340-347), in stages:
//   A  a cheap necessary test on every pixel (four opposite ring pairs), survivors -> work list
//   B  the exact score of the survivors, the ones that reach the threshold -> second list
//   C  suppression over the second list, the kept ones -> third list and a 64 x 64 bitmap
//   D  mask test and output of the kept ones in raster order (row prefix over the bitmap)
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ bool mask_is_zero(const image_src& s, unsigned y, unsigned x, float scale) {
    // is_in_mask of orb_extractor.cc:278-280: mask.at<uchar>(y * scale, x * scale) == 0
    int my = (int)((float)y * scale);
    int mx = (int)((float)x * scale);
    my = min(max(my, 0), s.mask_h - 1);
    mx = min(max(mx, 0), s.mask_w - 1);
    return s.mask[(long long)my * s.mask_stride + mx] == 0;
}

constexpr int kTileStride = 72; // bytes per LDS tile row; byte b of a row is image column min_x - 1 + b

// Stage A of the FAST kernel for one strip of four horizontally adjacent pixels: a rejection test on
// the four opposite ring pairs (0,8) (2,10) (4,12) (6,14).  A pixel that is a corner at threshold t
// has, in every opposite pair, at least one pixel darker than v - t, or in every pair at least one
// brighter than v + t (any arc of 9 covers one pixel of each pair).  The test runs on the pixels'
// upper six bits, four pixels per register, with plain 32-bit integer adds and bit operations
// (the packed 16-bit min / max / permute instructions issue at about half their rate):
//   x < v - t  =>  (x >> 2) <= (v >> 2) - T      x > v + t  =>  (x >> 2) >= (v >> 2) + T      T = (t + 1) >> 2
// so the six-bit test never rejects a corner; it lets a few more pixels through than the exact one
// would (stage B computes the exact score of whatever passes).
// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts.
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2, 3
    return v;
}

// `lo`, `mid`, `hi`: three consecutive dwords of the six-bit tile (bytes = columns c - 4 .. c + 7) of the
// strip's row (c*), of the rows two above (m*) and two below (p*); up3 / dn3: the strip's own dword of the
// rows three above and below.  `kq` holds 128 - T in every byte.  With six-bit values in eight-bit
// fields neither v + (128 - T) - x nor (128 - T) - v + x leaves its byte, and bit 7 of the byte says
// whether x <= v - T, respectively x >= v + T.  `xd` / `xb`: byte j = 1 iff pixel c + j passed the dark /
// bright test (the caller turns four such bytes into four mask bits with one v_dot4).
__device__ __forceinline__ void fast_quick_flags(u32 c0, u32 c1, u32 c2, u32 m0, u32 m1, u32 m2, u32 p0, u32 p1, u32 p2, u32 up3, u32 dn3, u32 kq,
                                                 u32& xd, u32& xb) {
    const u32 x4 = __builtin_amdgcn_alignbyte(c2, c1, 3), x12 = __builtin_amdgcn_alignbyte(c1, c0, 1);  // columns + 3 / - 3
    const u32 x2 = __builtin_amdgcn_alignbyte(p2, p1, 2), x14 = __builtin_amdgcn_alignbyte(p1, p0, 2);  // row + 2, columns + 2 / - 2
    const u32 x6 = __builtin_amdgcn_alignbyte(m2, m1, 2), x10 = __builtin_amdgcn_alignbyte(m1, m0, 2);  // row - 2
    const u32 vd = c1 + kq, vb = kq - c1;
    const u32 dark = ((vd - dn3) | (vd - up3)) & ((vd - x2) | (vd - x10)) & ((vd - x4) | (vd - x12)) & ((vd - x6) | (vd - x14));
    const u32 bright = ((vb + dn3) | (vb + up3)) & ((vb + x2) | (vb + x10)) & ((vb + x4) | (vb + x12)) & ((vb + x6) | (vb + x14));
    xd = (dark >> 7) & 0x01010101u;
    xb = (bright >> 7) & 0x01010101u;
}

// The four dwords 2 * pair .. 2 * pair + 3 of a tile row (columns 8 * pair - 4 .. 8 * pair + 11 of the
// cell interior) as two aligned 8-byte LDS reads at byte offsets `lo` + 72 * trow and `hi` + 72 * trow,
// lo = 8 * pair, hi = lo + 8.  The caller passes `hi` through an empty asm once, so that the compiler
// cannot tell that the two reads are neighbours and keeps two ds_read_b64 (which the LDS serves
// conflict free for the lane layout of stage A) instead of pairing them into one ds_read2_b64 (32
// banks, twice the cycles).  (`tile` must be the __shared__ array itself so that both reads stay LDS
// instructions; the opaque value is a byte offset, not a pointer.)
template <typename Tile>
__device__ __forceinline__ void tile_row_quad(Tile& tile, int trow, int lo, int hi, u32 (&d)[4]) {
    const uint2 a = *reinterpret_cast<const uint2*>(&tile[trow * kTileStride + lo]);
    const uint2 b = *reinterpret_cast<const uint2*>(&tile[trow * kTileStride + hi]);
    d[0] = a.x;
    d[1] = a.y;
    d[2] = b.x;
    d[3] = b.y;
}

// The exact score of a pixel on one side.  Dark side: v - min over the 16 arcs of 9 ring pixels of the
// arc maximum = the largest t + 1 at which every pixel of some arc is darker than v - t.  Bright side:
// max over the arcs of the arc minimum - v (the dark score of the inverted image).  Two separate
// functions: with the side as a run-time argument the compiler evaluates both sides for every pixel.
__device__ __forceinline__ void load_ring(const uint8_t* c, int (&x)[16]) {
    constexpr int S = kTileStride;
    x[0] = c[3 * S]; x[1] = c[3 * S + 1]; x[2] = c[2 * S + 2]; x[3] = c[S + 3];
    x[4] = c[3]; x[5] = c[-S + 3]; x[6] = c[-2 * S + 2]; x[7] = c[-3 * S + 1];
    x[8] = c[-3 * S]; x[9] = c[-3 * S - 1]; x[10] = c[-2 * S - 2]; x[11] = c[-S - 3];
    x[12] = c[-3]; x[13] = c[S - 3]; x[14] = c[2 * S - 2]; x[15] = c[3 * S - 1];
}
__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }

__device__ __forceinline__ int fast_dark_score(const uint8_t* c) {
    int x[16], mx3[16];
    load_ring(c, x);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mx3[k] = max3i(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
    }
    int lo = 255;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        lo = min(lo, max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
    }
    return (int)c[0] - lo;
}

__device__ __forceinline__ int fast_bright_score(const uint8_t* c) {
    int x[16], mn3[16];
    load_ring(c, x);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn3[k] = min3i(x[k], x[(k + 1) & 15], x[(k + 2) & 15]);
    }
    int hi = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        hi = max(hi, min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
    }
    return hi - (int)c[0];
}

// Workgroup barrier that orders LDS accesses only: no wait for outstanding global loads and stores
// (the cell kernel only exchanges data through LDS; its global stores are final outputs).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// What one cell needs to know about itself (the same in every thread of the workgroup).
struct fast_item {
    int ci, cj;          // cell row / column inside its level
    int min_x, min_y;    // window origin inside the level image
    int th;              // window rows
    int iw, ih;          // interior (scanned) columns / rows
    int skip;            // the reference skips the cell, or nothing is left to scan
    int dwords_inside;   // every tile dword lies inside its image row
    int gw;              // level width
    float scale;         // scale_factors_[level]
    long long out_cell;  // frame * total_cells + flattened cell
    const uint8_t* img;  // level image of the frame
    long long pitch;
};

// One record per flattened cell of a frame, built on the host with the plan: everything about the cell
// that does not depend on the frame or on the mask, so that a workgroup gets its geometry from a
// single 32-byte scalar load.
struct fast_cell_rec {
    u32 origin;   // min_x | min_y << 16: window origin inside the level image
    u32 extent;   // window rows | interior columns << 8 | interior rows << 16 | level << 24
    u32 where;    // cell row | cell column << 12 | skip << 24 | dwords_inside << 25
    int gw;       // level width
    float scale;  // scale_factors_[level]
    int pitch;    // row pitch of the level plane (levels >= 1)
    long long pyr_off; // byte offset of the level inside one frame's pyramid block (levels >= 1)
};
static_assert(sizeof(fast_cell_rec) == 32, "one s_load_dwordx8");

fast_cell_rec make_fast_cell_rec(const level_geom& g, int level, int ci, int cj) {
    fast_cell_rec r{};
    const int max_border_x = g.w - kBorder, max_border_y = g.h - kBorder;
    const int min_x = kBorder + cj * kCell, min_y = kBorder + ci * kCell;
    // cells the reference skips (orb_extractor.cc:310,323)
    bool skip = (max_border_y - kOverlap <= min_y) || (max_border_x - kOverlap <= min_x);
    const int max_x = std::min(min_x + kCell + kOverlap, max_border_x);
    const int max_y = std::min(min_y + kCell + kOverlap, max_border_y);
    const int th = max_y - min_y, iw = max_x - min_x - 6, ih = th - 6;
    skip = skip || iw <= 0 || ih <= 0;
    r.origin = static_cast<u32>(min_x) | (static_cast<u32>(min_y) << 16);
    r.extent = skip ? 0u : (static_cast<u32>(th) | (static_cast<u32>(iw) << 8) | (static_cast<u32>(ih) << 16));
    r.extent |= static_cast<u32>(level) << 24;
    r.where = static_cast<u32>(ci) | (static_cast<u32>(cj) << 12) | (skip ? 1u << 24 : 0u) | (min_x + 71 <= g.w ? 1u << 25 : 0u);
    r.gw = g.w;
    r.scale = g.scale;
    r.pitch = g.pitch;
    r.pyr_off = g.pyr_off;
    return r;
}

__device__ __forceinline__ void fast_geometry(const plan_dev& p, const image_src& s, const fast_cell_rec* __restrict__ cell_tab, int frame, int cell,
                                              fast_item& it) {
    const fast_cell_rec rec = cell_tab[cell];
    it.ci = (int)(rec.where & 0xFFFu);
    it.cj = (int)((rec.where >> 12) & 0xFFFu);
    it.out_cell = (long long)frame * p.total_cells + cell;
    it.gw = rec.gw;
    it.scale = rec.scale;
    it.min_x = (int)(rec.origin & 0xFFFFu);
    it.min_y = (int)(rec.origin >> 16);
    it.th = (int)(rec.extent & 0xFFu);
    it.iw = (int)((rec.extent >> 8) & 0xFFu);
    it.ih = (int)((rec.extent >> 16) & 0xFFu);
    bool skip = ((rec.where >> 24) & 1u) != 0;
    if (!skip && s.mask) {
        // orb_extractor.cc:332-337
        const int max_x = it.min_x + it.iw + 6, max_y = it.min_y + it.th;
        skip = mask_is_zero(s, it.min_y, it.min_x, rec.scale) || mask_is_zero(s, max_y, it.min_x, rec.scale)
               || mask_is_zero(s, it.min_y, max_x, rec.scale) || mask_is_zero(s, max_y, max_x, rec.scale);
    }
    it.skip = skip ? 1 : 0;
    it.dwords_inside = (int)((rec.where >> 25) & 1u);
    if ((rec.extent >> 24) == 0) {
        it.pitch = s.row_stride0;
        it.img = s.img0 + frame * s.frame_stride0;
    }
    else {
        it.pitch = rec.pitch;
        it.img = s.pyr + frame * p.frame_pyr_bytes + rec.pyr_off;
    }
}

typedef u32 u32_unaligned __attribute__((aligned(1)));

// Tile rows are 18 dwords starting one pixel left of the window, so that interior column c sits at
// byte c + 4 of its row (strips of four interior pixels are dword aligned in LDS).
template <typename Tile>
__device__ __forceinline__ void fast_load_tile(const fast_item& it, int tid, int lane, int wave, Tile& tile, Tile& tile6) {
    if (it.dwords_inside) {
        // every tile dword lies inside its image row: a wave fetches three rows at a time (18
        // lanes per row), all six fetches of a lane in flight before the first LDS write.  The
        // addresses have no particular alignment (the memory pipeline splits a dword that straddles
        // a cache line); rows past the window read its last row again and are not stored.
        const int lr = (lane * 57) >> 10; // lane / 18 for lane < 64
        const int kk = lane - 18 * lr;
        const int r0 = wave * 3 + lr;
        const uint8_t* base = it.img + (long long)it.min_y * it.pitch + (it.min_x - 1) + 4 * kk;
        // byte offsets of the rows r0, r0 + 12, ..., clamped to the window's last row: one multiplication
        const u32 step = 12u * (u32)it.pitch, last = (u32)(it.th - 1) * (u32)it.pitch;
        u32 off = (u32)r0 * (u32)it.pitch;
        u32 v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            v[j] = *reinterpret_cast<const u32_unaligned*>(base + min(off, last));
            off += step;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (lr < 3 && r0 + 12 * j < 70) {
                reinterpret_cast<u32*>(&tile[0])[(r0 + 12 * j) * 18 + kk] = v[j];
                reinterpret_cast<u32*>(&tile6[0])[(r0 + 12 * j) * 18 + kk] = (v[j] >> 2) & 0x3F3F3F3Fu;
            }
        }
    }
    else {
        // a window at the right image border: byte by byte where a dword would cross the row end
        for (int i = tid; i < it.th * 18; i += 256) {
            const int r = i / 18, k = i - r * 18;
            const int x = it.min_x - 1 + 4 * k;
            const uint8_t* src = it.img + (long long)(it.min_y + r) * it.pitch;
            u32 v;
            if (x + 3 < it.gw) {
                v = load_u32_unaligned(src + x);
            }
            else {
                v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (x + b < it.gw) {
                        v |= (u32)src[x + b] << (8 * b);
                    }
                }
            }
            reinterpret_cast<u32*>(&tile[0])[r * 18 + k] = v;
            reinterpret_cast<u32*>(&tile6[0])[r * 18 + k] = (v >> 2) & 0x3F3F3F3Fu;
        }
    }
}

// Grid: x = the cells of a frame (rounded up to a multiple of 8 workgroups), y = frames starting at
// frame0.  Workgroup x runs on XCD x % 8; the cells of a frame are cut into eight contiguous runs
// (neighbouring cells share image rows and meet in the XCD's L2) and the run an XCD gets rotates with
// the frame, because the runs differ in cost (the last one holds the small pyramid levels).
__global__ __launch_bounds__(256) void k_fast_cells(plan_dev p, image_src s, const fast_cell_rec* __restrict__ cell_tab, u32* __restrict__ slots,
                                                    u32* __restrict__ cell_counts, int frame0) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[70 * kTileStride];
    // one array, two uses: during the rejection test the tile's pixels >> 2 (same layout as `tile`),
    // afterwards the score plane (66 rows of 68 bytes)
    __shared__ __attribute__((aligned(16))) uint8_t plane[70 * kTileStride];
    static_assert(sizeof(plane) >= 66 * 68 + 8 && sizeof(plane) == 315 * 16, "the score plane fits, whole 16-byte stores clear it");
    auto& tile6 = plane;
    auto& score = plane;
    // survivors of the rejection test, one entry per pixel: those that passed the dark test (or both)
    // from the front, those that passed the bright test only from the back
    __shared__ unsigned short work[4096];
    __shared__ __attribute__((aligned(8))) u32 keep32[128]; // 64 rows x 64 columns, two dwords per row
    __shared__ int rowbase[65];
    __shared__ u32 nwork; // dark-side items | bright-side items << 16
    __shared__ int ncorner;
    __shared__ int total;
    constexpr int kCompactMax = 1365; // 3 * kCompactMax < 4096: dark, corner and kept lists fit side by side
    bool compact = false;

    const int per_run = (p.total_cells + 7) >> 3;
    const int cell = (int)((blockIdx.x + blockIdx.y) & 7u) * per_run + (int)(blockIdx.x >> 3);
    if (cell >= p.total_cells || (int)(blockIdx.x >> 3) >= per_run) {
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    fast_item cur;
    fast_geometry(p, s, cell_tab, frame0 + (int)blockIdx.y, cell, cur);
    if (cur.skip) {
        if (tid == 0) {
            cell_counts[cur.out_cell] = 0;
        }
        return;
    }
    fast_load_tile(cur, tid, lane, wave, tile, tile6);
    // (the counters and the bitmap of the first pass are cleared in front of the same barrier)
    if (tid == 0) {
        nwork = 0;
        ncorner = 0;
        total = 0;
    }
    if (tid < 128) {
        keep32[tid] = 0;
    }
    lds_barrier();

    int thr = min(max(p.ini_thr, 0), 255);
    const int min_thr = min(max(p.min_thr, 0), 255);
    // stage A geometry of this thread: eight columns (two strips of four, `pair`) of the rows r_first and
    // r_first + 32; two mask bits per pixel, set for the pixels inside the interior.  The 32 lanes of a
    // half wave cover eight pairs of the rows r, r + 8, r + 16, r + 24: with 18 dwords per tile row
    // those rows start 16 banks apart (18 * 8 = 144 = 2 * 64 + 16), so the 8-byte reads of a half wave
    // fall on 64 different banks.
    const int pair = lane & 7, r_first = 2 * wave + (lane >> 5) + 8 * ((lane >> 3) & 3);
    u32 valid_bits = 0xFFFFu; // bit 8 * k + j: pixel j of row r_first + 32 * k lies inside the interior
    if (cur.iw < 64 || cur.ih < 64) { // a window cut by the image border (same decision in every thread)
        valid_bits = 0;
        const u32 cols = (1u << min(max(cur.iw - 8 * pair, 0), 8)) - 1u;
        #pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (r_first + 32 * k < cur.ih) {
                valid_bits |= cols << (8 * k);
            }
        }
    }
    #pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        // stage A: rejection test, 8 strip pairs x 64 rows, two rows per thread.  Bit 8 * k + j of `md` /
        // `mbr` = pixel j (0..7) of row k passed the dark / bright test (one v_dot4 per side and strip
        // turns four flag bytes into mask bits).  Survivors are appended to the work list: a wave prefix
        // sum over the per-lane counts, one LDS atomic per wave, then every lane writes its own items,
        // the dark-side ones first.  An item is row << 6 | column; bit 12 of a dark-side item = the pixel
        // passed the bright test as well.
        {
            const u32 kq = (u32)(128 - ((thr + 1) >> 2)) * 0x01010101u;
            u32 accd[2], accb[2];
            // byte offsets of the thread's two 8-byte columns in tile row r_first
            const int lo = r_first * kTileStride + 8 * pair;
            int hi = lo + 8;
            asm volatile("" : "+v"(hi));
            #pragma unroll
            for (int k = 0; k < 2; ++k) {
                accd[k] = 0;
                accb[k] = 0;
                if (k == 1 && cur.ih <= 32) {
                    continue; // a window of at most 32 rows (bottom edge of a level): no thread has a second row
                }
                const int trow = 32 * k + 3; // relative to r_first
                u32 c[4], mm[4], pp[4], u3[4], d3[4];
                tile_row_quad(tile6, trow, lo, hi, c);
                tile_row_quad(tile6, trow - 2, lo, hi, mm);
                tile_row_quad(tile6, trow + 2, lo, hi, pp);
                tile_row_quad(tile6, trow - 3, lo, hi, u3);
                tile_row_quad(tile6, trow + 3, lo, hi, d3);
                u32 xd0, xb0, xd1, xb1;
                fast_quick_flags(c[0], c[1], c[2], mm[0], mm[1], mm[2], pp[0], pp[1], pp[2], u3[1], d3[1], kq, xd0, xb0);
                fast_quick_flags(c[1], c[2], c[3], mm[1], mm[2], mm[3], pp[1], pp[2], pp[3], u3[2], d3[2], kq, xd1, xb1);
                accd[k] = __builtin_amdgcn_udot4(xd1, 0x80402010u, __builtin_amdgcn_udot4(xd0, 0x08040201u, 0u, false), false);
                accb[k] = __builtin_amdgcn_udot4(xb1, 0x80402010u, __builtin_amdgcn_udot4(xb0, 0x08040201u, 0u, false), false);
            }
            const u32 md = (accd[0] | (accd[1] << 8)) & valid_bits;   // passed the dark test (maybe both)
            const u32 mbr = (accb[0] | (accb[1] << 8)) & valid_bits;  // passed the bright test (maybe both)
            const u32 mb = mbr & ~md;                                 // the bright test only
            const u32 both = mbr & md;
            const int cnt_d = __popc(md);
            const int cnt = cnt_d | (__popc(mb) << 16);
            const int incl = wave_incl_scan(cnt);
            const int wave_total = __builtin_amdgcn_readlane(incl, 63);
            if (wave_total) {
                u32 base = 0;
                if (lane == 0) {
                    base = atomicAdd(&nwork, (u32)wave_total);
                }
                const u32 first = (u32)__builtin_amdgcn_readfirstlane((int)base) + (u32)(incl - cnt);
                // the lane's item of rank i (dark-side ones first) goes to work[first_d + i] or, a
                // bright-side one, to work[4095 - first_b - (i - cnt_d)]; as byte offsets with i2 = 2 i:
                const u32 off_d = 2u * (first & 0xFFFFu);
                const u32 end_b = 2u * (4095u - (first >> 16)) + 2u * (u32)cnt_d;
                const u32 item0 = ((u32)r_first << 6) | (8u * (u32)pair);
                u32 px = md | (mb << 16); // bit q: dark-side item of pixel q, bit 16 + q: bright-side item
                const u32 both12 = both << 12;
                u32 i2 = 0;
                while (px) {
                    const u32 b = (u32)__builtin_ctz(px);
                    px &= px - 1u;
                    // pixel q = b & 15 = 8 k + j: row + 32 k, column + j
                    // (`both` has no bit above 15 and none at a bright-only pixel: shifting by b itself is enough)
                    const u32 it = (item0 | (b & 7u)) | ((b << 8) & 0x800u) | ((both12 >> b) & 0x1000u);
                    const u32 at = b < 16u ? off_d + i2 : end_b - i2;
                    *reinterpret_cast<unsigned short*>(reinterpret_cast<uint8_t*>(work) + at) = (unsigned short)it;
                    i2 += 2u;
                }
            }
        }
        lds_barrier();
        // the rejection test is done with the six-bit tile: its memory becomes the score plane, all zero
        {
            const uint4 z = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(score)[tid] = z;
            if (tid < 315 - 256) {
                reinterpret_cast<uint4*>(score)[256 + tid] = z;
            }
        }
        lds_barrier();
        // stage B: exact score of the survivors from the side that passed (a pixel that passed the
        // test on one side only cannot reach the threshold on the other, so the side score
        // decides; both sides for the rare pixel that passed both); below the threshold: 0.
        // Item i < n_dark comes from the front of `work`, the others from its back: the threads of a
        // wave take consecutive items, so all but one wave run one side's code only.
        // Items that reach the threshold go to a second list in the unused middle of `work` (when the
        // lists are short enough for that and for the third list below), the suppression pass then
        // only walks those.
        const u32 nwp = nwork;
        const int n_dark = (int)(nwp & 0xFFFFu), nw = n_dark + (int)(nwp >> 16);
        compact = nw <= kCompactMax;
        unsigned short* corners = work + n_dark; // the unused middle of `work`: at least 4096 - nw entries
        auto item_at = [&](int i) -> u32 { return i < n_dark ? work[i] : work[4095 - (i - n_dark)]; };
        for (int i = tid; i < nw; i += 256) {
            const u32 item = item_at(i);
            const int r = (item >> 6) & 63, c = item & 63;
            const int toff = (r + 3) * kTileStride + c + 4;
            int sc;
            if (i < n_dark) {
                sc = fast_dark_score(&tile[toff]) - 1;
                if (item & 0x1000u) {
                    // (the offset passes through an empty asm: otherwise the compiler hoists the bright
                    // score, which both arms contain, in front of the branch and every item pays for it)
                    int toff2 = toff;
                    asm volatile("" : "+v"(toff2));
                    sc = max(sc, fast_bright_score(&tile[toff2]) - 1);
                }
            }
            else {
                sc = fast_bright_score(&tile[toff]) - 1;
            }
            if (sc >= thr) { // (the plane is all zero: only corners are written)
                score[(r + 1) * 68 + c + 1] = (uint8_t)sc;
                if (compact) {
                    corners[atomicAdd(&ncorner, 1)] = (unsigned short)item;
                }
            }
        }
        lds_barrier();
        // non-maximum suppression (every pixel that is not on the list scores 0); in compact mode
        // the kept items are collected from the end of `work` downwards (the bright-side items there
        // are not read any more; dark + corner + kept items together stay below 4096 entries)
        const int nms_n = compact ? ncorner : nw;
        for (int i = tid; i < nms_n; i += 256) {
            const u32 item = compact ? (u32)corners[i] : item_at(i);
            const int r = (item >> 6) & 63, c = item & 63;
            const uint8_t* sp = &score[(r + 1) * 68 + c + 1];
            const int sc = sp[0];
            if (sc >= thr) {
                int m = 0;
                #pragma unroll
                for (int dy = -1; dy <= 1; ++dy) {
                    #pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (dy != 0 || dx != 0) {
                            m = max(m, (int)sp[dy * 68 + dx]);
                        }
                    }
                }
                if (sc > m) {
                    atomicOr(&keep32[2 * r + (c >> 5)], 1u << (c & 31));
                    const int k = atomicAdd(&total, 1);
                    if (compact) {
                        work[4095 - k] = (unsigned short)item;
                    }
                }
            }
        }
        lds_barrier();
        if (total != 0 || pass == 1 || min_thr == thr) {
            break;
        }
        thr = min_thr; // cell empty at the initial threshold: retry with the minimum one
        for (int i = tid; i < 70 * 18; i += 256) {
            reinterpret_cast<u32*>(&tile6[0])[i] = (reinterpret_cast<const u32*>(&tile[0])[i] >> 2) & 0x3F3F3F3Fu;
        }
        if (tid == 0) {
            nwork = 0;
            ncorner = 0;
            total = 0;
        }
        if (tid < 128) {
            keep32[tid] = 0;
        }
        lds_barrier();
    }
    if (total == 0) {
        if (tid == 0) {
            cell_counts[cur.out_cell] = 0;
        }
        return;
    }

    // the kept items (end of `work` downwards), or every survivor of the rejection test (both ends of `work`)
    const int n_dark_f = (int)(nwork & 0xFFFFu);
    const int n_items = compact ? total : n_dark_f + (int)(nwork >> 16);
    auto final_item = [&](int i) -> u32 { return compact ? work[4095 - i] : (i < n_dark_f ? work[i] : work[4095 - (i - n_dark_f)]); };
    if (s.mask) {
        // per-keypoint mask test, orb_extractor.cc:362 (after the emptiness test)
        for (int i = tid; i < n_items; i += 256) {
            const u32 item = final_item(i);
            const int r = (item >> 6) & 63, c = item & 63;
            if ((keep32[2 * r + (c >> 5)] >> (c & 31)) & 1u) {
                const unsigned y = kBorder + cur.ci * kCell + 3 + r;
                const unsigned x = kBorder + cur.cj * kCell + 3 + c;
                if (mask_is_zero(s, y, x, cur.scale)) {
                    atomicAnd(&keep32[2 * r + (c >> 5)], ~(1u << (c & 31)));
                }
            }
        }
        lds_barrier();
    }

    if (wave == 0) {
        const uint2 kr = reinterpret_cast<const uint2*>(keep32)[lane];
        const int cnt = __popc(kr.x) + __popc(kr.y);
        const int inc = wave_incl_scan(cnt);
        rowbase[lane] = inc - cnt;
        if (lane == 63) {
            rowbase[64] = inc;
        }
    }
    lds_barrier();

    // raster order: position = keypoints in the rows above + kept pixels to the left in the row
    u32* out = slots + cur.out_cell * kSlotsPerCell;
    for (int i = tid; i < n_items; i += 256) {
        const u32 item = final_item(i);
        const int r = (item >> 6) & 63, c = item & 63;
        const u64 bits = (u64)keep32[2 * r] | ((u64)keep32[2 * r + 1] << 32);
        if ((bits >> c) & 1ull) {
            const int pos = rowbase[r] + __popcll(bits & ((1ull << c) - 1ull));
            const u32 x = cur.cj * kCell + 3 + c; // keypt.pt.x += j * cell_size, orb_extractor.cc:359
            const u32 y = cur.ci * kCell + 3 + r;
            out[pos] = x | (y << 12) | ((u32)score[(r + 1) * 68 + c + 1] << 24);
        }
    }
    if (tid == 0) {
        cell_counts[cur.out_cell] = rowbase[64];
    }

}

// -------------------------------------------------------------------------------------------------
// 7x7 Gaussian blur, sigma 2, BORDER_REFLECT_101, 8-bit: separable integer kernel
// k = cvRound(g * 256) = {18, 34, 49, 55, 49, 34, 18}; exact integer row pass, column pass
// (sum + 2^15) >> 16 saturated (OpenCV 3.3.1 generic path; the Q8.8 fixed-point path of 3.4.1 .. 4.1.x computes
// the same integers from the same coefficients).  A second instance uses the coefficients that sum to 256
// exactly, {18, 34, 48, 56, 48, 34, 18} (OpenCV 3.4.9 / 4.2.0 and newer), selected per handle
// (ovs_orb_set_blur_kernel).
// -------------------------------------------------------------------------------------------------
#ifndef OVS_BLUR_ROWS
#define OVS_BLUR_ROWS 64
#endif
constexpr int kBlurRows = OVS_BLUR_ROWS;  // output rows per wave tile
constexpr int kBlurCols = 256; // output columns per wave tile (four per lane)

// The blur is the exact integer double sum  out = min((sum_j sum_i k[j] k[i] p[y+j][x+i] + 2^15) >> 16, 255):
// the row pass never rounds (<= 255 * 257 = 65535), so the passes may run in either order and
// on any grouping of taps.  One wavefront owns a tile of 256 columns x 64 rows and walks down
// the rows with no LDS, no barrier and no exchange between lanes:
//   * a lane fetches the twelve source bytes around its four columns (the dword left of its own, its
//     own, the one right of it) with three dword loads of any alignment; the second and third hit the
//     cache lines of the first.  BORDER_REFLECT_101 along the row is part of the fetch: every dword of
//     the reflected row is four bytes of ONE dword of the real row in some order (see blur_ext_dword),
//     so a lane at an image border loads from a shifted offset and reorders with one v_perm; tiles
//     that touch no border skip the v_perm (rows are reflected in scalar code);
//   * row pass on raw bytes: the seven taps are laid over the three dwords as they are and reduced
//     with v_dot4_u32_u8 (ten for four pixels);
//   * two consecutive row-pass results are packed as u16 pairs, so the column pass is four
//     v_dot2_u32_u16 per pixel; the accumulator starts at 0xFF008000 and every dot saturates, which
//     makes byte 2 of the result the rounded, 255-clamped output with no extra instruction.
__device__ __forceinline__ u32 udot2_sat(u32 a, u32 w, u32 acc) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, w), acc, true);
}

struct blur_row {
    u32 h[4]; // row-pass results of the lane's four columns
};

// single reflection, valid for -len < p < 2 * len - 1
__device__ __forceinline__ int reflect101_near(int p, int len) {
    p = p < 0 ? -p : p;
    return p >= len ? 2 * len - 2 - p : p;
}

// The dword of the reflected row ext[] (ext[x] = row[x] for 0 <= x < w, row[-x] left of it,
// row[2w - 2 - x] right of it) that starts at column c (a multiple of 4, -4 <= c) as a dword load at
// byte offset `off` of the real row followed by a byte permutation `sel` (v_perm selector, both
// sources the loaded dword):
//   c = -4            columns 4, 3, 2, 1: the dword at offset 1 reversed
//   c + 3 < w         the dword itself
//   c < w <= c + 3    k = w - c columns inside, then w - 2, w - 3, ...: all of them bytes of the dword at w - 4
//   w <= c <= w + 3   columns 2w - 2 - c downwards: the dword at 2w - 5 - c reversed
//   beyond            never used by a stored pixel: any valid offset
// (w >= 8; every offset lies in [0, w - 4].)
__device__ __forceinline__ void blur_ext_dword(int c, int w, u32& off, u32& sel) {
    constexpr u32 kIdentity = 0x03020100u, kReverse = 0x00010203u;
    off = (u32)min(max(c, 0), w - 4);
    sel = kIdentity;
    if (c < 0) {
        off = 1;
        sel = kReverse;
    }
    else if (c + 3 < w) {
    }
    else if (c < w) {
        const int k = w - c;
        sel = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sel |= (u32)(i < k ? i + 4 - k : k + 2 - i) << (8 * i);
        }
    }
    else if (c <= w + 3) {
        off = (u32)(2 * w - 5 - c);
        sel = kReverse;
    }
}

// Row pass for the lane's four columns from the twelve bytes (l, c, r).  The window of output column
// j (0..3) is bytes j + 1 .. j + 7 of the twelve: instead of cutting it out with v_alignbyte, the taps
// are laid over the three dwords as they are (weights of bytes outside the window are 0), which takes
// 2 + 3 + 3 + 2 v_dot4 for the four columns and no shifts.
// KSET: the coefficient set, OVS_BLUR_ROUNDED {18, 34, 49, 55, ...} or OVS_BLUR_NORMALISED {18, 34, 48, 56, ...}
template <int KSET>
struct blur_taps {
    static constexpr u32 k0 = 18, k1 = 34, k2 = KSET == OVS_BLUR_NORMALISED ? 48 : 49, k3 = KSET == OVS_BLUR_NORMALISED ? 56 : 55;
};
template <int KSET>
__device__ __forceinline__ blur_row blur_row_pass(u32 l, u32 c, u32 r) {
    constexpr u32 k0 = blur_taps<KSET>::k0, k1 = blur_taps<KSET>::k1, k2 = blur_taps<KSET>::k2, k3 = blur_taps<KSET>::k3;
    constexpr u32 kL0 = (k0 << 8) | (k1 << 16) | (k2 << 24), kC0 = k3 | (k2 << 8) | (k1 << 16) | (k0 << 24);
    constexpr u32 kL1 = (k0 << 16) | (k1 << 24), kC1 = k2 | (k3 << 8) | (k2 << 16) | (k1 << 24), kR1 = k0;
    constexpr u32 kL2 = (k0 << 24), kC2 = k1 | (k2 << 8) | (k3 << 16) | (k2 << 24), kR2 = k1 | (k0 << 8);
    constexpr u32 kC3 = k0 | (k1 << 8) | (k2 << 16) | (k3 << 24), kR3 = k2 | (k1 << 8) | (k0 << 16);
    blur_row out;
    out.h[0] = __builtin_amdgcn_udot4(c, kC0, __builtin_amdgcn_udot4(l, kL0, 0u, false), false);
    out.h[1] = __builtin_amdgcn_udot4(r, kR1, __builtin_amdgcn_udot4(c, kC1, __builtin_amdgcn_udot4(l, kL1, 0u, false), false), false);
    out.h[2] = __builtin_amdgcn_udot4(r, kR2, __builtin_amdgcn_udot4(c, kC2, __builtin_amdgcn_udot4(l, kL2, 0u, false), false), false);
    out.h[3] = __builtin_amdgcn_udot4(r, kR3, __builtin_amdgcn_udot4(c, kC3, 0u, false), false);
    return out;
}

// Column pass for one output row from four row pairs q[m][col] = h(row 2m) | h(row 2m + 1) << 16.
// ODD = false: rows 0..6 of the eight;  ODD = true: rows 1..7.
template <bool ODD, int KSET>
__device__ __forceinline__ u32 blur_col_pass(const u32 (&q0)[4], const u32 (&q1)[4], const u32 (&q2)[4], const u32 (&q3)[4]) {
    constexpr u32 k0 = blur_taps<KSET>::k0, k1 = blur_taps<KSET>::k1, k2 = blur_taps<KSET>::k2, k3 = blur_taps<KSET>::k3;
    constexpr u32 w0 = ODD ? (k0 << 16) : (k0 | (k1 << 16));
    constexpr u32 w1 = ODD ? (k1 | (k2 << 16)) : (k2 | (k3 << 16));
    constexpr u32 w2 = ODD ? (k3 | (k2 << 16)) : (k2 | (k1 << 16));
    constexpr u32 w3 = ODD ? (k1 | (k0 << 16)) : k0;
    u32 v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        u32 a = udot2_sat(q0[c], w0, 0xFF008000u);
        a = udot2_sat(q1[c], w1, a);
        a = udot2_sat(q2[c], w2, a);
        v[c] = udot2_sat(q3[c], w3, a);
    }
    // byte 2 of each accumulator
    const u32 lo = __builtin_amdgcn_perm(v[1], v[0], 0x0c0c0602u);
    const u32 hi = __builtin_amdgcn_perm(v[3], v[2], 0x06020c0cu);
    return lo | hi;
}

// The three dwords of one source row, loaded ahead of their use.
struct blur_raw {
    u32 l, c, r;
};

// Per-lane, per-tile constants of the row fetch; BORDER: the tile touches the first or the last column
// of the image, its lanes reorder what they load.
template <bool BORDER, int KSET>
struct blur_src {
    static constexpr int kset = KSET;
    const uint8_t* img;
    u32 pitch;
    u32 off_l, off_c, off_r;
    u32 sel_l, sel_c, sel_r;
    __device__ __forceinline__ blur_raw issue(int row) const {
        const uint8_t* rp = img + (size_t)((u32)row * pitch);
        // (the offsets pass through an empty asm so that their zero extension stays next to the loads and
        // the loads take the scalar-base + 32-bit-offset form instead of per-lane 64-bit addresses)
        u32 ol = off_l, oc = off_c, orr = off_r;
        asm volatile("" : "+v"(ol), "+v"(oc), "+v"(orr));
        blur_raw w;
        w.c = *reinterpret_cast<const u32_unaligned*>(rp + oc);
        w.l = *reinterpret_cast<const u32_unaligned*>(rp + ol);
        w.r = *reinterpret_cast<const u32_unaligned*>(rp + orr);
        return w;
    }
    __device__ __forceinline__ blur_row finish(const blur_raw& w) const {
        if (BORDER) {
            return blur_row_pass<KSET>(__builtin_amdgcn_perm(w.l, w.l, sel_l), __builtin_amdgcn_perm(w.c, w.c, sel_c), __builtin_amdgcn_perm(w.r, w.r, sel_r));
        }
        return blur_row_pass<KSET>(w.l, w.c, w.r);
    }
};

// The row loop of a wave tile: walks down the tile's source rows two at a time (the loads of the next
// pair in flight while the current pair is reduced) and emits two output rows per step.  Row pair t
// of the tile is image rows y0 - 3 + 2t, y0 - 2 + 2t; output rows 2i, 2i + 1 need pairs i .. i + 3.
template <typename Src>
__device__ __forceinline__ void blur_tile_rows(const level_geom& g, const Src& src, uint8_t* dst, int y0, int col, bool store_lane) {
    const int nrows = min(kBlurRows, g.h - y0);
    const int last = g.h + 2; // rows past the image that the last output row reaches: clamp, then reflect
    auto row_of = [&](int t, int k) { return reflect101_near(min(y0 - 3 + 2 * t + k, last), g.h); };
    u32 q[4][4];
    blur_raw pa = src.issue(row_of(0, 0)), pb = src.issue(row_of(0, 1));
    auto step = [&](int t, int qi) { // starts pair t + 1, reduces pair t into q[qi] (qi = t & 3, a constant after unrolling)
        const blur_raw na = src.issue(row_of(t + 1, 0));
        const blur_raw nb = src.issue(row_of(t + 1, 1));
        const blur_row a = src.finish(pa);
        const blur_row b = src.finish(pb);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            q[qi][c] = a.h[c] | (b.h[c] << 16);
        }
        pa = na;
        pb = nb;
    };
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        step(t, t);
    }
#pragma unroll 1
    for (int o = 0; o < nrows; o += 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int oo = o + 2 * i; // output rows oo, oo + 1: pairs i .. i + 3 (mod 4: the loop advances by four pairs)
            if (oo < nrows) {
                step((o >> 1) + i + 3, (i + 3) & 3);
                const u32 r0 = blur_col_pass<false, Src::kset>(q[i & 3], q[(i + 1) & 3], q[(i + 2) & 3], q[(i + 3) & 3]);
                const u32 r1 = blur_col_pass<true, Src::kset>(q[i & 3], q[(i + 1) & 3], q[(i + 2) & 3], q[(i + 3) & 3]);
                if (store_lane) {
                    u32 oc = (u32)col;
                    asm volatile("" : "+v"(oc)); // scalar row base + 32-bit lane offset, as for the loads
                    uint8_t* drow = dst + (size_t)((u32)oo * (u32)g.pitch);
                    *reinterpret_cast<u32*>(drow + oc) = r0;
                    if (oo + 1 < nrows) {
                        uint8_t* drow1 = dst + (size_t)((u32)(oo + 1) * (u32)g.pitch);
                        u32 oc1 = (u32)col;
                        asm volatile("" : "+v"(oc1));
                        *reinterpret_cast<u32*>(drow1 + oc1) = r1;
                    }
                }
            }
        }
    }
}

template <int KSET>
__device__ __forceinline__ void blur_tile(const level_geom& g, const uint8_t* img, u32 pitch, uint8_t* dst, int y0, int cb, int lane) {
    const int col = cb + 4 * lane;
    const bool store_lane = col < g.pitch;
    if (cb > 0 && cb + kBlurCols + 4 <= g.w) {
        // every source dword of the tile lies inside its image row
        const blur_src<false, KSET> src{img, pitch, (u32)col - 4u, (u32)col, (u32)col + 4u, 0u, 0u, 0u};
        blur_tile_rows(g, src, dst, y0, col, store_lane);
    }
    else {
        blur_src<true, KSET> src;
        src.img = img;
        src.pitch = pitch;
        blur_ext_dword(col - 4, g.w, src.off_l, src.sel_l);
        blur_ext_dword(col, g.w, src.off_c, src.sel_c);
        blur_ext_dword(col + 4, g.w, src.off_r, src.sel_r);
        blur_tile_rows(g, src, dst, y0, col, store_lane);
    }
}

template <int KSET>
__global__ __launch_bounds__(256) void k_gauss7(plan_dev p, image_src s, uint8_t* __restrict__ blur, int n_frames) {
    const int groups = (p.total_blur_tiles + 3) >> 2; // four wave tiles per workgroup
    const int blk = xcd_contiguous_block(groups * n_frames);
    if (blk < 0) {
        return;
    }
    const int frame = blk / groups;
    const int lane = threadIdx.x & 63;
    const int t = __builtin_amdgcn_readfirstlane((blk - frame * groups) * 4 + (threadIdx.x >> 6));
    if (t >= p.total_blur_tiles) {
        return;
    }
    int level = 0;
#pragma unroll 1
    for (int l = 1; l < p.num_levels; ++l) {
        if (t >= p.lv[l].blur_tile_base) {
            level = l;
        }
    }
    const level_geom& g = p.lv[level];
    const int rel = t - g.blur_tile_base;
    const int ty = rel / g.blur_tiles_x, tx = rel - ty * g.blur_tiles_x;
    const int y0 = ty * kBlurRows, cb = tx * kBlurCols;
    long long pitch64;
    const uint8_t* img = level_ptr(p, s, frame, level, pitch64);
    uint8_t* dst = blur + frame * p.frame_blur_bytes + g.blur_off + (long long)y0 * g.pitch;
    // (library planes are 256-byte aligned with a pitch that is a multiple of 64; a caller's level-0 image
    // may start anywhere and have any stride: the row fetch takes any alignment)
    blur_tile<KSET>(g, img, (u32)pitch64, dst, y0, cb, lane);
}

// -------------------------------------------------------------------------------------------------
// Quadtree keypoint distribution (orb_extractor.cc:398-590, orb_extractor_node.cc:6-50), one
// workgroup per (frame, level).
//
// The serial algorithm keeps a std::list of nodes, each owning a vector of keypoints; here the
// node list is an array in list order (node id == list position, double buffered in LDS) and every
// candidate keypoint carries the id of the node it currently belongs to.  One sweep of the serial
// loop ("split every non-leaf node, children pushed to the list front, parent erased") becomes:
//   1. candidates vote into their node's four child counters (LDS atomics),
//   2. an exclusive scan over the nodes in list order of the number of non-empty children gives
//      each child its creation index s; the new list is [children in reverse creation order]
//      followed by the untouched leaf nodes in their old order, i.e. child position = C - 1 - s,
//   3. candidates move to their child's id.
// The fill-up phase (orb_extractor.cc:444-472) sorts the splittable nodes created by the previous
// pass by (keypoint count, node address) descending; node addresses are replaced by creation
// order (later created first), which for nodes sitting at list positions 0..C-1 in reverse
// creation order is "count descending, then list position ascending".  Splitting stops at the
// first node in that order at which the list reaches the quota, found with a scan over the
// sorted pool.  Finally each node emits its first keypoint of maximum response (atomicMax on
// response << 24 | (2^24 - 1 - original index)).
// -------------------------------------------------------------------------------------------------
constexpr u32 kLeaf = 0x80000000u; // is_leaf_node_ (only set by initialize_nodes, orb_extractor.cc:545)
constexpr u32 kProc = 0x20000000u; // split in the current fill-up pass
constexpr u32 kCnt = 0x00FFFFFFu;

__device__ __forceinline__ int child_of(short4 b, int x, int y) {
    const int hx = (b.z - b.x + 1) >> 1; // cvCeil((end - begin) / 2.0)
    const int hy = (b.w - b.y + 1) >> 1;
    return (b.x + hx <= x ? 1 : 0) + (b.y + hy <= y ? 2 : 0);
}

__device__ __forceinline__ short4 child_bounds(short4 b, int c) {
    const short hx = (short)((b.z - b.x + 1) >> 1);
    const short hy = (short)((b.w - b.y + 1) >> 1);
    short4 r;
    r.x = (c & 1) ? (short)(b.x + hx) : b.x;
    r.z = (c & 1) ? b.z : (short)(b.x + hx);
    r.y = (c & 2) ? (short)(b.y + hy) : b.y;
    r.w = (c & 2) ? b.w : (short)(b.y + hy);
    return r;
}

// In-place exclusive scan of a[0..n) in LDS by the whole workgroup; returns the total.
// Workgroup barrier of the quadtree kernel.  NODES_LDS: the node arrays are in LDS, a plain barrier.
// Otherwise they are in a global workspace, where atomics execute in the L2 while plain loads may be
// served by the compute unit's vector cache: stores and atomics are complete before the barrier and
// the cache is invalidated after it.
template <bool NODES_LDS>
__device__ __forceinline__ void qt_sync() {
    if (NODES_LDS) {
        __syncthreads();
    }
    else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// Node array of the quadtree kernel.  In LDS an index is used as it is (a read past the allocation returns
// 0 there and a write is dropped).  In the global workspace (CLAMP) every index is clamped to the array, so
// that no path through the kernel, whatever it has computed, touches memory outside its workgroup's slice.
template <typename T, bool CLAMP>
struct qt_arr {
    T* p;
    u32 last; // index of the last element
    __device__ __forceinline__ T& operator[](u32 i) const { return p[CLAMP ? min(i, last) : i]; }
    __device__ __forceinline__ T& operator[](int i) const { return (*this)[(u32)i]; }
};

template <bool NODES_LDS, typename A>
__device__ u32 block_excl_scan(A a, int n, u32* red) {
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nwaves = nt >> 6;
    const int per = (n + nt - 1) / nt;
    const int b = min(tid * per, n), e = min(b + per, n);
    u32 sum = 0;
    for (int i = b; i < e; ++i) {
        sum += a[i];
    }
    const u32 inc = (u32)wave_incl_scan((int)sum);
    qt_sync<NODES_LDS>(); // red may still be read by a previous call
    if (lane == 63) {
        red[wave] = inc;
    }
    qt_sync<NODES_LDS>();
    u32 woff = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) {
        const u32 v = red[w];
        woff += (w < wave) ? v : 0;
        tot += v;
    }
    u32 run = woff + inc - sum;
    for (int i = b; i < e; ++i) {
        const u32 v = a[i];
        a[i] = run;
        run += v;
    }
    qt_sync<NODES_LDS>();
    return tot;
}

__device__ __forceinline__ int nonzero4(const u32* c) {
    return (c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0);
}

// ctr[target] += 1 for every lane with `valid`, for counters that many lanes of a wave hit at once (the
// first passes of a level have a handful of nodes for thousands of candidates, and LDS atomics on one
// address are served one lane at a time): the wave peels off one distinct target per round and its
// first lane adds the number of lanes that share it.  Called by all lanes of the wave.
template <typename A>
__device__ __forceinline__ void lds_count_aggregated(A ctr, u32 target, bool valid) {
    u64 todo = __ballot(valid);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const u32 t = (u32)__builtin_amdgcn_readlane((int)target, leader);
        const u64 same = __ballot(valid && target == t);
        if (lane == leader) {
            atomicAdd(&ctr[t], (u32)__popcll(same));
        }
        todo &= ~same;
    }
}
constexpr int kAggregateBelow = 32; // counters: above this the lanes of a wave rarely meet

// Candidate record: x | y << 16 in one array (written once), response << 24 | current node id in a
// second one (rewritten by every sweep: a wave's 64 records are 256 contiguous bytes, whole cache
// lines, where an interleaved layout would rewrite half of every line).
// Every pass over the candidates is latency bound (global loads feeding LDS atomics), and the
// largest level of a frame is the critical path of the whole launch, so the passes are kept few
// and deep: moving the candidates to their child nodes and voting into the child counters of the
// NEXT sweep are one pass (the counters are double buffered), and each thread issues four loads
// before it touches the first result.
constexpr int kCandDepth = 4; // candidate loads a thread keeps in flight
template <typename F>
__device__ __forceinline__ void for_each_candidate(const u32* __restrict__ cnode, int N, F&& f) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int i0 = tid; i0 < N; i0 += kCandDepth * nt) {
        u32 cd[kCandDepth];
#pragma unroll
        for (int u = 0; u < kCandDepth; ++u) {
            const int i = i0 + u * nt;
            cd[u] = cnode[min(i, N - 1)];
        }
#pragma unroll
        for (int u = 0; u < kCandDepth; ++u) {
            const int i = i0 + u * nt;
            if (i < N) {
                f(i, cd[u]);
            }
        }
    }
}

// One pass that moves every candidate from its node `n` of the current list to a node `nn` of the
// list being built and votes for the next split of `nn`.  The per-candidate work is a chain of
// dependent LDS reads (node record -> child id -> record of the child) that ends in an LDS atomic,
// and an atomic keeps the thread's later LDS reads behind it; the chain is therefore walked for a
// whole batch of candidates at a time, stage by stage, so that a batch costs three LDS round trips
// instead of five per candidate.
//   stage1(n)                      an extra word read together with the record of node n
//   stage2(n, rec, b, aux, x, y)   the new node id (may read one more LDS word)
template <typename NC, typename NB, typename Stage1, typename Stage2>
__device__ __forceinline__ void move_and_vote(const u32* __restrict__ cxy, u32* __restrict__ cnode, int N, NC nc_from, NB nb_from,
                                              NC nc_to, NB nb_to, NC votes, int n_votes, Stage1&& stage1, Stage2&& stage2) {
    // n_votes = number of vote counters of this pass (the same in every thread)
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool aggregate = n_votes <= kAggregateBelow;
    for (int i0 = tid; i0 < N; i0 += kCandDepth * nt) {
        uint2 cd[kCandDepth];
#pragma unroll
        for (int u = 0; u < kCandDepth; ++u) {
            cd[u] = make_uint2(cxy[min(i0 + u * nt, N - 1)], cnode[min(i0 + u * nt, N - 1)]);
        }
        u32 rec[kCandDepth], aux[kCandDepth];
        short4 b[kCandDepth];
#pragma unroll
        for (int u = 0; u < kCandDepth; ++u) {
            const u32 n = cd[u].y & kCnt;
            rec[u] = nc_from[n];
            b[u] = nb_from[n];
            aux[u] = stage1(n);
        }
        u32 nn[kCandDepth];
#pragma unroll
        for (int u = 0; u < kCandDepth; ++u) {
            nn[u] = stage2(cd[u].y & kCnt, rec[u], b[u], aux[u], (int)(cd[u].x & 0xFFFF), (int)(cd[u].x >> 16));
            // A lane past the last candidate re-reads candidate N - 1, possibly after its owner has moved it:
            // the node id it then holds belongs to the other list and the id derived from it is arbitrary.
            // Nothing such a lane computes is used, but the record reads below must stay inside the arrays
            // (in LDS a read past the allocation returns 0; in the global workspace it is a fault).
            nn[u] = i0 + u * nt < N ? nn[u] : 0u;
        }
        u32 rec2[kCandDepth];
        short4 b2[kCandDepth];
#pragma unroll
        for (int u = 0; u < kCandDepth; ++u) {
            rec2[u] = nc_to[nn[u]];
            b2[u] = nb_to[nn[u]];
        }
#pragma unroll
        for (int u = 0; u < kCandDepth; ++u) {
            const int i = i0 + u * nt;
            const bool votes_now = i < N && !(rec2[u] & kLeaf);
            const u32 target = 4 * nn[u] + child_of(b2[u], (int)(cd[u].x & 0xFFFF), (int)(cd[u].x >> 16));
            if (i < N) {
                cnode[i] = (cd[u].y & 0xFF000000u) | nn[u];
            }
            if (aggregate) {
                lds_count_aggregated(votes, target, votes_now);
            }
            else if (votes_now) {
                atomicAdd(&votes[target], 1u);
            }
        }
    }
}

// A launch covers the levels level0 .. level0 + gridDim.y - 1 and sizes its node arrays for `ncap`, the
// largest node capacity among them (72 bytes per node).  NODES_LDS: the arrays are in LDS (levels whose
// quota fits: up to about 1130 keypoints per level); otherwise in `node_ws`, one slice per workgroup
// (slower by the latency of every access, but any quota below 8188 per level works).
constexpr int kQtWordsPerNode = 18;
constexpr int kQtMaxPasses = 64;
template <bool NODES_LDS>
__global__ __launch_bounds__(kQtThreadsLive) void k_quadtree(plan_dev p, const u32* __restrict__ slots, const u32* __restrict__ cell_counts,
                                                         u32* __restrict__ cand_xy_all, u32* __restrict__ cand_node_all,
                                                         u32* __restrict__ lvl_xy, uint8_t* __restrict__ lvl_resp, int* __restrict__ lvl_count,
                                                         int level0, int ncap, u32* node_ws) {
    extern __shared__ u32 smem[];
    // Workgroups are dealt round-robin to the 8 XCDs in launch order and a level-0 workgroup runs
    // about three times as long as one of the last level: with the frame as the fast grid dimension
    // every XCD gets the same mix of levels, and the long workgroups start first.
    const int level = level0 + (int)blockIdx.y, frame = blockIdx.x;
    const level_geom& g = p.lv[level];
    const int tid = threadIdx.x, nt = blockDim.x;
    u32* nodes = NODES_LDS ? smem : node_ws + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ((size_t)ncap * kQtWordsPerNode);

    typedef qt_arr<u32, !NODES_LDS> arr_u;
    typedef qt_arr<short4, !NODES_LDS> arr_b;
    const u32 last1 = (u32)ncap - 1, last4 = 4 * (u32)ncap - 1;
    arr_b nb_cur{reinterpret_cast<short4*>(nodes), last1};
    arr_b nb_nxt{nb_cur.p + ncap, last1};
    arr_u nc_cur{reinterpret_cast<u32*>(nb_nxt.p + ncap), last1};
    arr_u nc_nxt{nc_cur.p + ncap, last1};
    arr_u cc{nc_nxt.p + ncap, last4};   // 4 * ncap child counters of the current list, later child ids
    arr_u cc2{cc.p + 4 * ncap, last4};  // the same for the list being built
    arr_u t0{cc2.p + 4 * ncap, last1};
    arr_u t1{t0.p + ncap, last1};
    arr_u t2{t1.p + ncap, last1};
    arr_u t3{t2.p + ncap, last1};
    u32* cellpref = NODES_LDS ? t3.p + ncap : smem; // max_cells + 1 (always in LDS, like the scan scratch and the scalars)
    u32* red = cellpref + p.max_cells + 1; // 16
    u32* sh = red + 16;                    // scalars

    const int ncells = g.cells_x * g.cells_y;
    const long long cell0 = (long long)frame * p.total_cells + g.cell_base;
    u32* cxy = cand_xy_all + frame * p.frame_cand_cap + g.cand_base;
    u32* cnode = cand_node_all + frame * p.frame_cand_cap + g.cand_base;
    const int quota = g.quota;

    for (int c = tid; c < ncells; c += nt) {
        cellpref[c] = cell_counts[cell0 + c];
    }
    qt_sync<NODES_LDS>();
    const int N = (int)block_excl_scan<NODES_LDS>(cellpref, ncells, red);
    if (tid == 0) {
        cellpref[ncells] = N;
    }
    const int n0 = g.nx * g.ny;
    for (int q = tid; q < n0; q += nt) {
        t0[q] = 0;
    }
    // Initial node of a keypoint (orb_extractor.cc:528-535): column (unsigned)(pt.x / delta_x) in
    // double precision.  The quotient is monotone in the integer pt.x, so column k starts at the
    // smallest x whose quotient reaches k; those few thresholds are found once with the same
    // division and the per-candidate work becomes integer compares.  (t2 / t3 hold them: the
    // node arrays have at least 4 * nx * ny + 8 entries.)
    arr_u thx = t2;
    arr_u thy = t3;
    for (int k = tid; k < g.nx + g.ny; k += nt) {
        const bool is_x = k < g.nx;
        const int kk = is_x ? k : k - g.nx;
        const double delta = is_x ? g.delta_x : g.delta_y;
        int x = max((int)((double)kk * delta) - 2, 0);
        while ((unsigned)((double)(float)x / delta) < (unsigned)kk) {
            ++x;
        }
        (is_x ? thx : thy)[kk] = (u32)x; // threshold 0 is 0
    }
    qt_sync<NODES_LDS>();

    // dense candidates in the reference's order: cells row-major, raster order inside a cell
    for (int i0 = tid; i0 < N; i0 += 4 * nt) {
        u32 v[4];
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * nt, N - 1);
            idx[u] = i;
            // cell of candidate i: last c with cellpref[c] <= i
            int lo = 0, hi = ncells - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((int)cellpref[mid] <= i) {
                    lo = mid;
                }
                else {
                    hi = mid - 1;
                }
            }
            v[u] = slots[(cell0 + lo) * kSlotsPerCell + (i - (int)cellpref[lo])];
        }
        // all LDS reads of the batch before its first LDS atomic: an atomic orders every later LDS
        // access of the thread behind it
        u32 ni[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 x = v[u] & 0xFFFu, y = (v[u] >> 12) & 0xFFFu;
            u32 ix = 0, iy = 0;
            for (int k = 1; k < g.nx; ++k) {
                ix += x >= thx[k];
            }
            for (int k = 1; k < g.ny; ++k) {
                iy += y >= thy[k];
            }
            ni[u] = min(ix + iy * (unsigned)g.nx, (unsigned)(n0 - 1));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool valid = i0 + u * nt < N;
            if (valid) {
                const u32 x = v[u] & 0xFFFu, y = (v[u] >> 12) & 0xFFFu, r = v[u] >> 24;
                cxy[idx[u]] = x | (y << 16);
                cnode[idx[u]] = (r << 24) | ni[u];
            }
            // (a level starts with one node per image-aspect unit: a handful of counters for all candidates)
            if (n0 <= kAggregateBelow) {
                lds_count_aggregated(t0, ni[u], valid);
            }
            else if (valid) {
                atomicAdd(&t0[ni[u]], 1u);
            }
        }
    }
    qt_sync<NODES_LDS>();
    for (int q = tid; q < n0; q += nt) {
        t1[q] = t0[q] != 0;
    }
    qt_sync<NODES_LDS>();
    int S = (int)block_excl_scan<NODES_LDS>(t1, n0, red);
    for (int q = tid; q < n0; q += nt) {
        const u32 cnt = t0[q];
        if (cnt) {
            const int ix = q % g.nx, iy = q / g.nx;
            short4 b;
            b.x = (short)(int)(g.delta_x * ix);
            b.y = (short)(int)(g.delta_y * iy);
            b.z = (short)(int)(g.delta_x * (ix + 1));
            b.w = (short)(int)(g.delta_y * (iy + 1));
            nb_cur[t1[q]] = b;
            nc_cur[t1[q]] = cnt | (cnt == 1 ? kLeaf : 0u);
        }
    }
    for (int i = tid; i < 4 * S; i += nt) {
        cc[i] = 0;
    }
    qt_sync<NODES_LDS>();
    // compact node ids + votes for the first sweep
    move_and_vote(
        cxy, cnode, N, t1, nb_cur, nc_cur, nb_cur, cc, 4 * S, [&](u32) { return 0u; },
        [&](u32, u32 compact_id, short4, u32, int, int) { return compact_id; });
    qt_sync<NODES_LDS>();

    bool filled = false;
    bool overflow = false;
    int front = 0; // number of freshly created nodes at the head of the list
    // Every pass of either loop splits at least one node one level deeper, and a node of a 4134-px level
    // cannot be halved more than 13 times: the loops end by themselves long before this many passes.  The
    // bound only makes sure that every wave reaches the end of the kernel whatever the node arrays hold.
    int passes_left = kQtMaxPasses;

    // ---- orb_extractor.cc:411-442: split every non-leaf node per sweep -------------------------
    // invariant at the top: cc holds the child counts of every non-leaf node of the current list
    while (true) {
        const int prev = S;
        if (tid == 0) {
            sh[0] = 0;
        }
        // one scan for both prefix sums: children created before node q in the low half, leaves
        // before it in the high half (both stay below 2^16: at most 4 * ncap children)
        for (int q = tid; q < S; q += nt) {
            const bool leaf = nc_cur[q] & kLeaf;
            t0[q] = leaf ? 0x10000u : (u32)nonzero4(&cc[4 * q]);
        }
        qt_sync<NODES_LDS>();
        const u32 both = block_excl_scan<NODES_LDS>(t0, S, red);
        const int C = (int)(both & 0xFFFFu);
        const int Lf = (int)(both >> 16);
        const int Snew = C + Lf;
        if (Snew > ncap || --passes_left < 0) {
            overflow = true;
            break;
        }
        u32 local_pool = 0;
        for (int q = tid; q < S; q += nt) {
            if (nc_cur[q] & kLeaf) {
                const int qn = C + (int)(t0[q] >> 16);
                nb_nxt[qn] = nb_cur[q];
                nc_nxt[qn] = nc_cur[q];
                cc[4 * q] = qn;
            }
            else {
                const short4 b = nb_cur[q];
                const int created_before = (int)(t0[q] & 0xFFFFu);
                int r = 0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const u32 k = cc[4 * q + c];
                    if (k) {
                        const int qn = C - 1 - (created_before + r);
                        nb_nxt[qn] = child_bounds(b, c);
                        nc_nxt[qn] = k;
                        cc[4 * q + c] = qn;
                        local_pool += k > 1;
                        ++r;
                    }
                }
            }
        }
        if (local_pool) {
            atomicAdd(&sh[0], local_pool);
        }
        for (int i = tid; i < 4 * Snew; i += nt) {
            cc2[i] = 0;
        }
        qt_sync<NODES_LDS>();
        // move every candidate to its child and vote for the next split of that child
        move_and_vote(
            cxy, cnode, N, nc_cur, nb_cur, nc_nxt, nb_nxt, cc2, 4 * Snew, [&](u32) { return 0u; },
            [&](u32 n, u32 rec, short4 b, u32, int x, int y) { return cc[4 * n + ((rec & kLeaf) ? 0 : child_of(b, x, y))]; });
        const int pool = (int)sh[0];
        qt_sync<NODES_LDS>();
        {
            arr_b tb = nb_cur;
            nb_cur = nb_nxt;
            nb_nxt = tb;
            arr_u tc = nc_cur;
            nc_cur = nc_nxt;
            nc_nxt = tc;
            tc = cc;
            cc = cc2;
            cc2 = tc;
        }
        S = Snew;
        front = C;
        if (quota <= S || S == prev) {
            filled = true;
            break;
        }
        if (quota < S + pool) {
            filled = false;
            break;
        }
    }

    // ---- orb_extractor.cc:444-472: split the fullest nodes first until the quota is reached ----
    // cc still holds the child counts of every non-leaf node, in particular of the pool
    while (!filled && !overflow) {
        const int prev = S;
        for (int q = tid; q < front; q += nt) {
            t0[q] = (nc_cur[q] & kCnt) > 1;
        }
        if (tid == 0) {
            sh[1] = 0x7FFFFFFFu;
        }
        qt_sync<NODES_LDS>();
        const int P = (int)block_excl_scan<NODES_LDS>(t0, front, red);
        if (P == 0) {
            break;
        }
        for (int q = tid; q < front; q += nt) {
            if ((nc_cur[q] & kCnt) > 1) {
                t1[t0[q]] = q;
            }
        }
        qt_sync<NODES_LDS>();
        // stable sort by count descending (rank by counting over a dense copy of the counts; the
        // pool is at most a few hundred nodes)
        for (int k = tid; k < P; k += nt) {
            t3[k] = nc_cur[t1[k]] & kCnt;
        }
        qt_sync<NODES_LDS>();
        for (int k = tid; k < P; k += nt) {
            const u32 ck = t3[k];
            int rank = 0;
            for (int j = 0; j < P; ++j) {
                const u32 cj = t3[j];
                rank += (cj > ck) || (cj == ck && j < k);
            }
            t2[rank] = t1[k];
        }
        qt_sync<NODES_LDS>();
        for (int k = tid; k < P; k += nt) {
            t0[k] = nonzero4(&cc[4 * t2[k]]);
        }
        qt_sync<NODES_LDS>();
        block_excl_scan<NODES_LDS>(t0, P, red);
        // first pool entry (in sorted order) at which the list reaches the quota
        for (int k = tid; k < P; k += nt) {
            const int created = (int)t0[k] + nonzero4(&cc[4 * t2[k]]);
            if (quota <= S + created - (k + 1)) {
                atomicMin(&sh[1], (u32)k);
            }
        }
        qt_sync<NODES_LDS>();
        const int kstar = min((int)sh[1], P - 1);
        const int Cn = (int)t0[kstar] + nonzero4(&cc[4 * t2[kstar]]);
        const int Snew = Cn + S - (kstar + 1);
        const bool filled_now = quota <= Snew;
        if (Snew > ncap || --passes_left < 0) {
            overflow = true;
            break;
        }
        for (int k = tid; k <= kstar; k += nt) {
            nc_cur[t2[k]] |= kProc;
        }
        qt_sync<NODES_LDS>();
        for (int q = tid; q < S; q += nt) {
            t3[q] = (nc_cur[q] & kProc) ? 0 : 1;
        }
        qt_sync<NODES_LDS>();
        block_excl_scan<NODES_LDS>(t3, S, red);
        for (int k = tid; k <= kstar; k += nt) {
            const u32 q = t2[k];
            const short4 b = nb_cur[q];
            int r = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32 kk = cc[4 * q + c];
                if (kk) {
                    const int qn = Cn - 1 - ((int)t0[k] + r);
                    nb_nxt[qn] = child_bounds(b, c);
                    nc_nxt[qn] = kk;
                    cc[4 * q + c] = qn;
                    ++r;
                }
            }
        }
        for (int q = tid; q < S; q += nt) {
            if (!(nc_cur[q] & kProc)) {
                const int qn = Cn + (int)t3[q];
                nb_nxt[qn] = nb_cur[q];
                nc_nxt[qn] = nc_cur[q];
            }
        }
        for (int i = tid; i < 4 * Snew; i += nt) {
            cc2[i] = 0;
        }
        qt_sync<NODES_LDS>();
        move_and_vote(
            cxy, cnode, N, nc_cur, nb_cur, nc_nxt, nb_nxt, cc2, 4 * Snew, [&](u32 n) { return t3[n]; },
            [&](u32 n, u32 rec, short4 b, u32 kept_before, int x, int y) {
                const u32 child_id = cc[4 * n + ((rec & kProc) ? child_of(b, x, y) : 0)];
                return (rec & kProc) ? child_id : (u32)Cn + kept_before;
            });
        qt_sync<NODES_LDS>();
        {
            arr_b tb = nb_cur;
            nb_cur = nb_nxt;
            nb_nxt = tb;
            arr_u tc = nc_cur;
            nc_cur = nc_nxt;
            nc_nxt = tc;
            tc = cc;
            cc = cc2;
            cc2 = tc;
        }
        S = Snew;
        front = Cn;
        if (filled_now || quota <= S || S == prev) {
            filled = true;
        }
    }

    if (overflow) {
        if (tid == 0) {
            lvl_count[frame * p.num_levels + level] = -1;
        }
        return;
    }

    // ---- orb_extractor.cc:568-590: first keypoint of maximum response per node ------------------
    qt_sync<NODES_LDS>();
    for (int q = tid; q < S; q += nt) {
        cc[q] = 0;
    }
    qt_sync<NODES_LDS>();
    for_each_candidate(cnode, N, [&](int i, u32 cd) {
        atomicMax(&cc[cd & kCnt], (cd & 0xFF000000u) | (0xFFFFFFu - (u32)i));
    });
    qt_sync<NODES_LDS>();
    const long long ob = (long long)frame * p.kp_total_cap + g.kp_base;
    for (int q = tid; q < S; q += nt) {
        const u32 v = cc[q];
        const u32 i = 0xFFFFFFu - (v & 0xFFFFFFu);
        lvl_xy[ob + q] = cxy[i];
        lvl_resp[ob + q] = (uint8_t)(v >> 24);
    }
    if (tid == 0) {
        lvl_count[frame * p.num_levels + level] = S;
    }
}

// -------------------------------------------------------------------------------------------------
// Orientation + descriptor, one wavefront per keypoint.
//   ic_angle (orb_extractor.cc:608-631): lanes 0..30 own the 31 rows of the radius-15 disc, integer
//   moments are reduced across the wave (exact in any order), cv::fastAtan2 in fp32 without fused
//   multiply-add.
//   compute_orb_descriptor (orb_extractor.cc:641-702): lane l evaluates pairs 4l..4l+3 of the
//   pattern; util::cos / util::sin (util/trigonometric.h:17-46) and the rotated coordinates are
//   evaluated in the reference's operation order in fp32, cvRound = round-half-even.
// -------------------------------------------------------------------------------------------------
// Tables of the descriptor kernel, built on the host (upload_constants) in the layout the kernel keeps
// in LDS, so that a workgroup stages them with one 16-byte load per thread:
//   g_pat[17 g + k]               (x0, y0, x1, y1) of sample pair 16 g + k as fp32 (padded so that the 16 lanes
//                                 of a keypoint's group hit distinct banks)
//   g_disc_mask[(av * 4 + mis) * 3 + part]   byte masks of the three dwords of a 12-byte part of the disc row
//                                 |v| = av (16 = empty) whose first byte lies `mis` bytes left of u = -16
__device__ float4 g_pat[16 * 17];
__device__ uint4 g_disc_mask[17 * 4 * 3];

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = (float)DBL_EPSILON;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    }
    else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) {
        a = __fsub_rn(180.f, a);
    }
    if (y < 0) {
        a = __fsub_rn(360.f, a);
    }
    return a;
}

__device__ __forceinline__ int cv_floor_f(float v) {
    const int i = (int)v;
    return i - (i > v);
}

__device__ __forceinline__ float trig_poly(float v) {
    const float c1 = 0.99940307f, c2 = -0.49558072f, c3 = 0.03679168f;
    const float v2 = __fmul_rn(v, v);
    return __fadd_rn(c1, __fmul_rn(v2, __fadd_rn(c2, __fmul_rn(c3, v2))));
}

__device__ __forceinline__ float trig_cos(float v) {
    const float PI = 3.14159265358979f;
    const float PI_2 = PI / 2.0f;
    const float TWO_PI = 2.0f * PI;
    const float INV_TWO_PI = 1.0f / TWO_PI;
    const float THREE_PI_2 = 3.0f * PI_2;
    v = __fsub_rn(v, __fmul_rn((float)cv_floor_f(__fmul_rn(v, INV_TWO_PI)), TWO_PI));
    v = (0.0f < v) ? v : -v;
    if (v < PI_2) {
        return trig_poly(v);
    }
    else if (v < PI) {
        return -trig_poly(__fsub_rn(PI, v));
    }
    else if (v < THREE_PI_2) {
        return -trig_poly(__fsub_rn(v, PI));
    }
    return trig_poly(__fsub_rn(TWO_PI, v));
}

__device__ __forceinline__ float trig_sin(float v) {
    const float PI_2 = 3.14159265358979f / 2.0f;
    return trig_cos(__fsub_rn(PI_2, v));
}

constexpr int kKpGroup = 16; // lanes per keypoint in k_orient_desc

// Sum over the 16 lanes of a DPP row, returned in every lane: four rotations inside the row
// (no LDS round trip, unlike ds_bpermute based shuffles).
__device__ __forceinline__ int row16_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false); // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false); // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false); // row_ror:1
    return v;
}
// Inclusive prefix sum over the 16 lanes of a DPP row (shifts that fill with 0).
__device__ __forceinline__ int row16_incl_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true); // row_shr:8
    return v;
}
static_assert(kKpGroup == 16, "a keypoint's lane group is one DPP row");

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16 lanes per keypoint, 16 keypoints per workgroup.  Scattered byte reads from global memory cost
// one cache-line lookup per lane, so both patches are fetched with row-contiguous dword loads:
//   * moments: a disc row is three 12-byte parts from the dword boundary left of it, one part per
//     lane and load (six loads cover the 31 rows); byte masks from an LDS table indexed by |v|,
//     the row's misalignment and the part cut the dwords to |u| <= u_max[|v|], then sum I and
//     sum (window byte index) I are v_dot4_u32_u8 chains; integer sums, reduced across the group;
//   * descriptor: the 37 x 37 neighbourhood of the blurred level (pattern radius 13 rotated:
//     <= 18.4) goes to LDS as 37 rows of 11 aligned dwords; the 512 rotated sample reads are LDS
//     byte reads.  Lane g owns pairs 16g .. 16g+15, i.e. bytes 2g and 2g+1 of the descriptor;
//   * pattern: converted once per workgroup to fp32 in LDS (padded so a group's 16 lanes hit
//     distinct banks); the rotation runs on packed fp32 (v_pk_mul_f32 / v_pk_add_f32 are plain
//     IEEE multiplies and adds, so the reference's mul, mul, add order and rounding are kept),
//     cvRound = adding 1.5 * 2^23 (nearest even, the integer lands in the low mantissa bits);
//   * descriptor bits are collected with sub + v_alignbit (sign bit shifted in), reversed once.
constexpr int kPatchR = 18;                 // rows / columns -18 .. 18
constexpr int kPatchRows = 2 * kPatchR + 1; // 37
constexpr int kPatchDwords = 12;            // 48 bytes per row: 37 + up to 3 bytes of alignment, fetched as three 16-byte parts
constexpr int kPatchBytes = kPatchRows * kPatchDwords * 4;

__global__ __launch_bounds__(256) void k_orient_desc(plan_dev p, image_src s, const uint8_t* __restrict__ blur,
                                                     const u32* __restrict__ lvl_xy, const uint8_t* __restrict__ lvl_resp,
                                                     const int* __restrict__ lvl_count,
                                                     ovs_keypoint* __restrict__ out_kps, uint8_t* __restrict__ out_desc,
                                                     int* __restrict__ out_counts, int cap, int blocks_per_frame, int n_frames) {
    __shared__ float4 pat[16 * 17];   // (x0, y0, x1, y1) of pair 16 g + k at [17 g + k]
    __shared__ __attribute__((aligned(16))) u32 disc_mask[17][4][3][4]; // byte masks of a 12-byte part of a disc row by |v| (16 = empty) and row misalignment
    __shared__ __attribute__((aligned(16))) uint8_t patch[16][kPatchBytes];

    const int blk = xcd_contiguous_block(blocks_per_frame * n_frames);
    if (blk < 0) {
        return;
    }
    const int frame = blk / blocks_per_frame;
    const int fblk = blk - frame * blocks_per_frame; // block of 16 output slots inside the frame
    const int gl = threadIdx.x & (kKpGroup - 1);
    const int grp = threadIdx.x / kKpGroup;
    // Output slot oi of the frame (level-major concatenation, orb_extractor.cc:120-143) is keypoint
    // oi - before(level) of the level whose running count first exceeds oi.  The per-level counts of
    // the frame are the same for every thread of the workgroup: scalar loads (every workgroup of the
    // frame reads the same few dwords, so all but the first hit the scalar cache) and scalar sums; a
    // workgroup past the frame's last keypoint leaves before it has issued a vector instruction.
    const int* __restrict__ cnt = lvl_count + frame * p.num_levels;
    int incl_l[16];
    int total = 0, bad = 0;
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        const int cl = cnt[l]; // (the array ends with 16 spare entries: every load is in bounds, no branch between them)
        const int c = l < p.num_levels ? cl : 0;
        bad |= c < 0 ? 1 : 0;
        total += max(c, 0);
        incl_l[l] = total;
    }
    const int oi = fblk * (256 / kKpGroup) + grp;
    if (oi == 0 && gl == 0) {
        // a negative count flags a truncated frame (capacity) or an internal overflow (INT_MIN)
        out_counts[frame] = bad ? (int)0x80000000 : (total > cap ? -total : total);
    }
    const int n_out = bad ? 0 : min(total, cap);
    if (fblk * (256 / kKpGroup) >= n_out) {
        return; // the whole workgroup is past the last keypoint (same decision in every thread)
    }
    {
        const int t = threadIdx.x;
        pat[t] = g_pat[t];
        if (t < 16 * 17 - 256) {
            pat[256 + t] = g_pat[256 + t];
        }
        if (t < 17 * 4 * 3) {
            reinterpret_cast<uint4*>(&disc_mask[0][0][0][0])[t] = g_disc_mask[t];
        }
    }
    __syncthreads();
    if (oi >= n_out) {
        return;
    }
    // level = number of levels whose inclusive count is <= oi; before = the count of the lower levels
    int level = 0, before = 0;
#pragma unroll
    for (int l = 0; l < 16; ++l) {
        if (l < p.num_levels && incl_l[l] <= oi) {
            level = l + 1;
            before = incl_l[l];
        }
    }
    const level_geom& g = p.lv[level];
    const int j = g.kp_base + (oi - before); // slot in the per-frame level-keypoint arrays

    const long long li = (long long)frame * p.kp_total_cap + j;
    const u32 xy = lvl_xy[li];
    const int px = (int)(xy & 0xFFFF) + kBorder, py = (int)(xy >> 16) + kBorder; // pt += min_border, orb_extractor.cc:384-385
    const int resp = lvl_resp[li];

    // ---- blurred neighbourhood -> LDS (issued first: the moments below hide its latency) ----
    // blurred planes of the frame as one buffer; rows of a plane share their alignment (pitch is a
    // multiple of 64), so the patch is 37 rows of 11 dwords starting at the aligned address at or
    // below its first byte
    const __amdgpu_buffer_rsrc_t bplane = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(blur) + frame * p.frame_blur_bytes, 0,
                                                                            (int)p.frame_blur_bytes, 0x00020000);
    const int step = g.pitch;
    const int first = (int)g.blur_off + __mul24(py - kPatchR, step) + px - kPatchR; // byte (-18, -18)
    const int first_al = first & ~3, sh = first & 3;
    u32* my_patch = reinterpret_cast<u32*>(&patch[grp][0]);
    // A row is three 16-byte parts, a part is one lane's buffer_load_dwordx4 (the texture path
    // spends its time per lane address, not per byte: 7 wide loads instead of 26 dword loads):
    // slot 16 k + gl is part (slot % 3) of row (slot / 3), 111 slots, the last load has one spare lane.
    constexpr int kPatchLoads = (3 * kPatchRows + kKpGroup - 1) / kKpGroup; // 7
    static_assert(kPatchDwords == 12, "three 16-byte parts per row");
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32x4 staged[kPatchLoads];
#pragma unroll
    for (int k = 0; k < kPatchLoads; ++k) {
        const int slot = 16 * k + gl;
        const int row = (slot * 171) >> 9; // slot / 3 for slot < 112
        staged[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bplane, first_al + __mul24(row, step) + 16 * (slot - 3 * row), 0, 0));
    }

    // ---- intensity centroid: m_10 = sum u * I, m_01 = sum v * I over the disc ----
    // A disc row is the 31 bytes u = -15 .. 15; from the dword boundary at or below u = -16 that is
    // at most 36 bytes = three 12-byte parts, each one lane's 12-byte load: 93 parts in six loads
    // of the group's 16 lanes (the three spare slots read row 16 through an empty mask).
    // (All products here have small factors: 24-bit multiplies.  A 32-bit or 64-bit integer multiply
    // takes four times as long to issue, and the row address of every part used to be one.)
    long long pitch;
    const uint8_t* img = level_ptr(p, s, frame, level, pitch);
    typedef u32 u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    const int pitch32 = (int)pitch; // a level plane is far smaller than 2^31 bytes
    const uint8_t* win = img + (__mul24(py, pitch32) + (px - 16)); // (v, u) = (0, -16)
    const u32 win_lo = (u32)reinterpret_cast<uintptr_t>(win);
    const uint4* masks = reinterpret_cast<const uint4*>(&disc_mask[0][0][0][0]);
    int m10 = 0, m01 = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const u32 slot = 16u * k + (u32)gl;
        const u32 ri = __umul24(slot, 171u) >> 9; // slot / 3
        const u32 part = slot - __umul24(ri, 3u);
        const int v = (int)ri - kHalfPatch;       // 16 in the spare slots
        const u32 av = (u32)(v < 0 ? -v : v);
        const int off = __mul24(v, pitch32);      // row v relative to the window row
        const u32 mis = (win_lo + (u32)off) & 3u; // misalignment of the row's byte u = -16
        const int part12 = (int)__umul24(part, 12u);
        const u32x3_a4 w = *reinterpret_cast<const u32x3_a4*>(win + (off - (int)mis + part12));
        const uint4 mk = masks[__umul24(av, 12u) + __umul24(mis, 3u) + part];
        const u32 w0 = w.x & mk.x, w1 = w.y & mk.y, w2 = w.z & mk.z;
        // sum of the part's bytes and sum of (byte index inside the part) * byte
        const int s0 = (int)__builtin_amdgcn_udot4(w0, 0x01010101u, __builtin_amdgcn_udot4(w1, 0x01010101u, __builtin_amdgcn_udot4(w2, 0x01010101u, 0u, false), false), false);
        const int s1 = (int)__builtin_amdgcn_udot4(w0, 0x03020100u, __builtin_amdgcn_udot4(w1, 0x07060504u, __builtin_amdgcn_udot4(w2, 0x0B0A0908u, 0u, false), false), false);
        m10 += s1 + __mul24(part12 - 16 - (int)mis, s0); // u = 12 part + byte index - 16 - mis
        m01 += __mul24(v, s0);
    }
    m10 = row16_sum_i32(m10);
    m01 = row16_sum_i32(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    const float arad = (float)((double)angle * 3.14159265358979323846 / 180.0);
    const float ca = trig_cos(arad), sa = trig_sin(arad);

#pragma unroll
    for (int k = 0; k < kPatchLoads; ++k) {
        const int slot = 16 * k + gl;
        if (slot < 3 * kPatchRows) {
            reinterpret_cast<u32x4*>(my_patch)[slot] = staged[k]; // rows are 48 bytes: part `slot` sits at 16 * slot
        }
    }
    // the 16 lanes of a group are in one wavefront: their LDS writes are visible to each other
    // once the wave's own LDS operations have completed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- rotated BRIEF on the staged neighbourhood ----
    // cvRound without a conversion: x + 1.5 * 2^23 rounds x to the nearest integer, ties to even,
    // exactly as v_rndne / cvRound do (|x| < 2^22), and leaves it in the low mantissa bits:
    // bits(x + M) = 0x4B400000 + round(x).  v_mul_i32_i24 reads bits 0..23 only, i.e.
    // 0x400000 + round(x), so  row * 44 + col = mad24(bits_row, 44, bits_col) - kMagicBias  and the
    // constant moves into the base address: one v_pk_add_f32 per sample point instead of two
    // v_rndne + two v_cvt.
    constexpr float kMagic = 12582912.0f;
    constexpr int kMagicBias = 0x400000 * (kPatchDwords * 4) + 0x4B400000;
    const uint8_t* centre = &patch[grp][0] + kPatchR * (kPatchDwords * 4) + kPatchR + sh - kMagicBias;
    const f32x2 sc = {sa, ca};  // (sin, cos)
    const f32x2 cs = {ca, -sa}; // (cos, -sin): y * (cos, -sin) added to x * (sin, cos) gives (row, col)
    const f32x2 magic = {kMagic, kMagic};
    u32 bits = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 pt = pat[gl * 17 + k];
        // row = x * sin + y * cos, col = x * cos - y * sin  (orb_extractor.cc:676-678); the
        // negated product y * (-sin) equals -(y * sin) exactly, so the subtraction is unchanged
        const f32x2 rc0 = ((f32x2){pt.x, pt.x} * sc + (f32x2){pt.y, pt.y} * cs) + magic;
        const f32x2 rc1 = ((f32x2){pt.z, pt.z} * sc + (f32x2){pt.w, pt.w} * cs) + magic;
        const int v0 = centre[__mul24(__float_as_int(rc0.x), kPatchDwords * 4) + __float_as_int(rc0.y)];
        const int v1 = centre[__mul24(__float_as_int(rc1.x), kPatchDwords * 4) + __float_as_int(rc1.y)];
        bits = __builtin_amdgcn_alignbit(bits, (u32)(v0 - v1), 31); // bits << 1 | (v0 < v1)
    }
    bits = __brev(bits) >> 16; // pair k of the lane at bit k
    reinterpret_cast<unsigned short*>(out_desc + ((long long)frame * cap + oi) * 32)[gl] = (unsigned short)bits;

    if (gl < 7) {
        float fx = (float)px, fy = (float)py;
        if (level != 0) {
            fx = __fmul_rn(fx, g.scale); // correct_keypoint_scale, orb_extractor.cc:598-606
            fy = __fmul_rn(fy, g.scale);
        }
        u32 field;
        switch (gl) {
            case 0: field = __float_as_uint(fx); break;
            case 1: field = __float_as_uint(fy); break;
            case 2: field = __float_as_uint(g.patch_size); break;
            case 3: field = __float_as_uint(angle); break;
            case 4: field = __float_as_uint((float)resp); break;
            case 5: field = (u32)level; break;
            default: field = 0xFFFFFFFFu; break; // class_id = -1
        }
        reinterpret_cast<u32*>(out_kps + (long long)frame * cap + oi)[gl] = field;
    }
}

} // namespace

// =================================================================================================
// Host side
// =================================================================================================
struct ovs_orb {
    int device = 0;
    // orb_params
    uint32_t max_num_keypts = 2000;
    float scale_factor = 1.2f;
    uint32_t num_levels = 8;
    uint32_t ini_fast_thr = 20;
    uint32_t min_fast_thr = 7;
    std::vector<float> mask_rects; // 4 per rectangle

    std::vector<float> scale_factors, inv_scale_factors, level_sigma_sq, inv_level_sigma_sq;
    std::vector<uint32_t> num_keypts_per_level;
    int u_max[16];

    // plan for the current image size
    int plan_w = 0, plan_h = 0;
    plan_dev plan;
    ovs::dev_buffer tabs;
    ovs::dev_buffer cell_tab; // fast_cell_rec per flattened cell of a frame
    bool constants_uploaded = false;

    // rectangle mask (orb_extractor.cc:82-85: created once, for the size of the first image)
    bool mask_is_initialized = false;
    ovs::dev_buffer rect_mask;
    int rect_mask_w = 0, rect_mask_h = 0;

    int blur_kernel = OVS_BLUR_ROUNDED; // coefficient set of the 7 x 7 blur
    // batch-capacity dependent buffers
    int batch_cap = 0;
    ovs::dev_buffer pyr, blur, slots, cell_counts, cand, lvl_xy, lvl_resp, lvl_count, qt_ws;
    // staging for the single-frame host entry point
    ovs::dev_buffer st_img, st_mask, st_out;  // st_out: count (256 B), keypoints, descriptors
    ovs::pinned_buffer pin_in, pin_out;
    hipStream_t own_stream = nullptr;
    // the blur runs beside FAST + quadtree on a second stream (fork after the pyramid, join before
    // the descriptors)
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;

    // state of the last extraction (for the pyramid accessors)
    const uint8_t* last_img0 = nullptr;
    long long last_row_stride0 = 0, last_frame_stride0 = 0;
    int last_frames = 0;
};

namespace {

void orb_initialize(ovs_orb* h) {
    // orb_extractor::initialize(), orb_extractor.cc:205-239
    const uint32_t L = h->num_levels;
    h->scale_factors.resize(L);
    h->inv_scale_factors.resize(L);
    h->level_sigma_sq.resize(L);
    h->inv_level_sigma_sq.resize(L);
    ovs_calc_scale_factors(L, h->scale_factor, h->scale_factors.data());
    ovs_calc_inv_scale_factors(L, h->scale_factor, h->inv_scale_factors.data());
    ovs_calc_level_sigma_sq(L, h->scale_factor, h->level_sigma_sq.data());
    ovs_calc_inv_level_sigma_sq(L, h->scale_factor, h->inv_level_sigma_sq.data());

    h->num_keypts_per_level.resize(L);
    double desired_num_keypts_per_scale
        = h->max_num_keypts * (1.0 - 1.0 / h->scale_factor)
          / (1.0 - std::pow(1.0 / h->scale_factor, static_cast<double>(L)));
    unsigned int total_num_keypts = 0;
    for (unsigned int level = 0; level < L - 1; ++level) {
        h->num_keypts_per_level.at(level) = std::round(desired_num_keypts_per_scale);
        total_num_keypts += h->num_keypts_per_level.at(level);
        desired_num_keypts_per_scale *= 1.0 / h->scale_factor;
    }
    h->num_keypts_per_level.at(L - 1) = std::max(static_cast<int>(h->max_num_keypts) - static_cast<int>(total_num_keypts), 0);

    const int half = kHalfPatch;
    std::vector<int> u_max(half + 2, 0);
    const unsigned int vmax = std::floor(half * std::sqrt(2.0) / 2 + 1);
    const unsigned int vmin = std::ceil(half * std::sqrt(2.0) / 2);
    for (unsigned int v = 0; v <= vmax; ++v) {
        u_max.at(v) = std::round(std::sqrt(static_cast<double>(half * half) - static_cast<double>(v * v)));
    }
    for (unsigned int v = half, v0 = 0; vmin <= v; --v) {
        while (u_max.at(v0) == u_max.at(v0 + 1)) {
            ++v0;
        }
        u_max.at(v) = v0;
        ++v0;
    }
    for (int i = 0; i < 16; ++i) {
        h->u_max[i] = u_max[i];
    }
    h->plan_w = h->plan_h = 0; // geometry depends on the parameters
    h->constants_uploaded = false;
}

inline short saturate_short(float v) {
    const int iv = static_cast<int>(lrintf(v));
    return static_cast<short>(std::min(std::max(iv, -32768), 32767));
}

// xofs / ialpha (and yofs / ibeta) of cv::resize INTER_LINEAR, 8-bit fixed point path
void resize_tables(int ssize, int dsize, bool clamp, int* ofs, int* coef) {
    const double inv_scale = static_cast<double>(dsize) / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int sidx = static_cast<int>(f);
        sidx -= (sidx > f); // cvFloor
        f -= sidx;
        if (clamp) {
            if (sidx < 0) {
                f = 0;
                sidx = 0;
            }
            if (sidx >= ssize - 1) {
                f = 0;
                sidx = ssize - 1;
            }
        }
        ofs[d] = sidx;
        const unsigned a0 = static_cast<unsigned short>(saturate_short((1.f - f) * 2048));
        const unsigned a1 = static_cast<unsigned short>(saturate_short(f * 2048));
        coef[d] = static_cast<int>(a0 | (a1 << 16));
    }
}

// true when the row-streaming resize kernel applies: every aligned group of four output columns
// reads at most 8 consecutive source bytes and the source rows advance by one or two per output row
bool resize_tables_stream(const int* xofs, int dw, const int* yofs, int dh) {
    for (int d = 0; d + 1 < dw; ++d) {
        if (xofs[d + 1] < xofs[d]) {
            return false;
        }
    }
    for (int d0 = 0; d0 < dw; d0 += 4) {
        const int last = std::min(d0 + 3, dw - 1);
        if (xofs[last] + 1 - xofs[d0] >= 8) {
            return false;
        }
    }
    if (dh < 1 || yofs[0] < 0) {
        return false;
    }
    for (int d = 0; d + 1 < dh; ++d) {
        const int step = yofs[d + 1] - yofs[d];
        if (step < 1 || step > 2) {
            return false;
        }
    }
    return true;
}

bool is_area_fast_2x(int sw, int sh, int dw, int dh) {
    const double scale_x = 1. / (static_cast<double>(dw) / sw);
    const double scale_y = 1. / (static_cast<double>(dh) / sh);
    const int iscale_x = static_cast<int>(lrint(scale_x));
    const int iscale_y = static_cast<int>(lrint(scale_y));
    return std::fabs(scale_x - iscale_x) < DBL_EPSILON && std::fabs(scale_y - iscale_y) < DBL_EPSILON
           && iscale_x == 2 && iscale_y == 2;
}

// grid size of a kernel that maps its blocks with xcd_contiguous_block()
unsigned xcd_grid(long long total) { return static_cast<unsigned>(((total + 7) / 8) * 8); }

size_t quadtree_lds_bytes(const plan_dev& p, int ncap) {
    // nb x2 (8 B), nc x2 (4 B), cc x2 (16 B), t0..t3 (16 B) per node + cell prefix + scratch
    static_assert(kQtWordsPerNode * 4 == 16 + 8 + 32 + 16, "node record");
    return static_cast<size_t>(ncap) * (kQtWordsPerNode * 4) + static_cast<size_t>(p.max_cells + 1) * 4 + (16 + 8) * 4;
}
constexpr size_t kQtMaxLds = 160 * 1024;
// levels [0, first_lds_level) keep their node arrays in the global workspace
int quadtree_first_lds_level(const plan_dev& p) {
    int first = 0;
    for (int l = 0; l < p.num_levels; ++l) {
        if (quadtree_lds_bytes(p, p.lv[l].ncap) > kQtMaxLds) {
            first = l + 1;
        }
    }
    return first;
}

int build_plan(ovs_orb* h, int width, int height) {
    if (h->plan_w == width && h->plan_h == height) {
        return OVS_OK;
    }
    const int L = static_cast<int>(h->num_levels);
    if (L > kMaxLevels) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 16 pyramid levels are not supported");
    }
    if (width > 4096 + 2 * kBorder || height > 4096 + 2 * kBorder) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "images larger than 4134 pixels per side are not supported");
    }
    plan_dev p;
    std::memset(&p, 0, sizeof(p));
    p.num_levels = L;
    p.ini_thr = static_cast<int>(h->ini_fast_thr);
    p.min_thr = static_cast<int>(h->min_fast_thr);
    long long pyr_off = 0, blur_off = 0, cand_base = 0;
    int cell_base = 0, kp_base = 0, tab_off = 0, blur_tile_base = 0;
    std::vector<int> tabs;
    for (int l = 0; l < L; ++l) {
        level_geom& g = p.lv[l];
        const double scale = h->scale_factors.at(l);
        // orb_extractor.cc:267-268
        g.w = l == 0 ? width : static_cast<int>(std::round(width * 1.0 / scale));
        g.h = l == 0 ? height : static_cast<int>(std::round(height * 1.0 / scale));
        if (g.w <= 2 * kBorder || g.h <= 2 * kBorder) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "image too small for the pyramid");
        }
        g.pitch = (g.w + 63) & ~63;
        g.pyr_off = pyr_off;
        if (l >= 1) {
            pyr_off += static_cast<long long>(g.pitch) * g.h;
        }
        g.blur_off = blur_off;
        blur_off += static_cast<long long>(g.pitch) * g.h;
        const unsigned bw = g.w - 2 * kBorder, bh = g.h - 2 * kBorder;
        g.cells_x = static_cast<int>(bw / kCell) + 1; // std::ceil(width / cell_size) + 1 with integer division
        g.cells_y = static_cast<int>(bh / kCell) + 1;
        g.cell_base = cell_base;
        cell_base += g.cells_x * g.cells_y;
        g.quota = static_cast<int>(h->num_keypts_per_level.at(l));
        // initial nodes, orb_extractor.cc:480-499 (min_x == min_y == 19, so the `max_x - min_y`
        // of the portrait branch equals the width)
        const double ratio = static_cast<double>(bw) / bh;
        if (ratio > 1) {
            g.nx = static_cast<int>(static_cast<unsigned>(std::round(ratio)));
            g.ny = 1;
            g.delta_x = static_cast<double>(bw) / g.nx;
            g.delta_y = bh;
        }
        else {
            g.nx = 1;
            g.ny = static_cast<int>(static_cast<unsigned>(std::round(1 / ratio)));
            g.delta_x = bw;
            g.delta_y = static_cast<double>(bh) / g.ny;
        }
        // list length bound: a sweep runs only while nodes + splittable <= quota, which caps the
        // next list at 2 * quota; the first sweep yields at most 4 nodes per initial node; the
        // fill-up phase overshoots by at most 3
        g.ncap = std::max(2 * g.quota, 4 * g.nx * g.ny) + 8;
        g.kp_base = kp_base;
        kp_base += g.ncap;
        g.cand_base = cand_base;
        cand_base += static_cast<long long>(g.cells_x) * g.cells_y * kSlotsPerCell;
        g.scale = h->scale_factors.at(l);
        g.patch_size = static_cast<float>(static_cast<unsigned int>(31 * h->scale_factors.at(l)));
        g.blur_tile_base = blur_tile_base;
        g.blur_tiles_x = (g.w + kBlurCols - 1) / kBlurCols; // wave tiles of kBlurCols x kBlurRows
        g.blur_tiles_y = (g.h + kBlurRows - 1) / kBlurRows;
        blur_tile_base += g.blur_tiles_x * g.blur_tiles_y;
        g.tab_off = tab_off;
        g.area2x = 0;
        if (l >= 1) {
            const level_geom& gs = p.lv[l - 1];
            g.area2x = is_area_fast_2x(gs.w, gs.h, g.w, g.h) ? 1 : 0;
            const int w4 = (g.w + 3) & ~3;
            const int seg = (2 * w4 + 2 * g.h + 3) & ~3; // keeps every level's tables 16-byte aligned
            tabs.resize(tab_off + seg, 0);
            resize_tables(gs.w, g.w, true, tabs.data() + tab_off, tabs.data() + tab_off + w4);
            resize_tables(gs.h, g.h, false, tabs.data() + tab_off + 2 * w4, tabs.data() + tab_off + 2 * w4 + g.h);
            g.resize_rows = (!g.area2x && resize_tables_stream(tabs.data() + tab_off, g.w, tabs.data() + tab_off + 2 * w4, g.h)) ? 1 : 0;
            tab_off += seg;
        }
        p.max_cells = std::max(p.max_cells, g.cells_x * g.cells_y);
        p.max_ncap = std::max(p.max_ncap, g.ncap);
    }
    p.total_cells = cell_base;
    p.total_blur_tiles = blur_tile_base;
    p.kp_total_cap = kp_base;
    p.frame_pyr_bytes = (pyr_off + 255) & ~255ll;
    p.frame_blur_bytes = (blur_off + 255) & ~255ll;
    p.frame_cand_cap = cand_base;
    if (p.max_ncap >= 16384) {
        // (the sweeps pack two node counts of at most 4 * ncap into one 32-bit word)
        return ovs::fail(OVS_ERR_UNSUPPORTED, "more than 8187 keypoints per pyramid level are not supported");
    }
    {
        // dynamic LDS of the quadtree kernel: granted once per device, to the most a workgroup can
        // have, and never lowered (plans of other handles and threads share the kernel)
        static std::mutex mu;
        static std::set<int> granted;
        int dev = 0;
        OVS_HIP(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lock(mu);
        if (!granted.count(dev)) {
            OVS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            granted.insert(dev);
        }
    }
    {
        std::vector<fast_cell_rec> cell_tab(static_cast<size_t>(p.total_cells));
        for (int l = 0; l < L; ++l) {
            const level_geom& g = p.lv[l];
            for (int c = 0; c < g.cells_x * g.cells_y; ++c) {
                cell_tab[g.cell_base + c] = make_fast_cell_rec(g, l, c / g.cells_x, c % g.cells_x);
            }
        }
        int rc = h->cell_tab.reserve(std::max<size_t>(cell_tab.size() * sizeof(fast_cell_rec), 256));
        if (rc != OVS_OK) {
            return rc;
        }
        OVS_HIP(hipMemcpy(h->cell_tab.ptr, cell_tab.data(), cell_tab.size() * sizeof(fast_cell_rec), hipMemcpyHostToDevice));
    }
    if (!tabs.empty()) {
        int rc = h->tabs.reserve(tabs.size() * sizeof(int));
        if (rc != OVS_OK) {
            return rc;
        }
        OVS_HIP(hipMemcpy(h->tabs.ptr, tabs.data(), tabs.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    h->plan = p;
    h->plan_w = width;
    h->plan_h = height;
    h->batch_cap = 0; // buffers depend on the geometry
    return OVS_OK;
}

int ensure_buffers(ovs_orb* h, int n_frames) {
    if (n_frames <= h->batch_cap) {
        return OVS_OK;
    }
    const plan_dev& p = h->plan;
    const size_t B = static_cast<size_t>(n_frames);
    int rc;
    if ((rc = h->pyr.reserve(std::max<size_t>(B * p.frame_pyr_bytes, 256))) != OVS_OK) return rc;
    if ((rc = h->blur.reserve(B * p.frame_blur_bytes)) != OVS_OK) return rc;
    if ((rc = h->slots.reserve(B * p.total_cells * kSlotsPerCell * sizeof(u32))) != OVS_OK) return rc;
    if ((rc = h->cell_counts.reserve(B * p.total_cells * sizeof(u32))) != OVS_OK) return rc;
    if ((rc = h->cand.reserve(B * p.frame_cand_cap * sizeof(uint2))) != OVS_OK) return rc;
    if ((rc = h->lvl_xy.reserve(B * p.kp_total_cap * sizeof(u32))) != OVS_OK) return rc;
    if ((rc = h->lvl_resp.reserve(B * p.kp_total_cap)) != OVS_OK) return rc;
    if ((rc = h->lvl_count.reserve((B * p.num_levels + 16) * sizeof(int))) != OVS_OK) return rc; // + 16: k_orient_desc reads 16 entries from a frame's first
    if (const int gl = quadtree_first_lds_level(p)) {
        // node arrays of the levels whose quota does not fit in LDS: one slice per (level, frame)
        int ncap = 0;
        for (int l = 0; l < gl; ++l) {
            ncap = std::max(ncap, p.lv[l].ncap);
        }
        if ((rc = h->qt_ws.reserve(B * gl * static_cast<size_t>(ncap) * kQtWordsPerNode * sizeof(u32))) != OVS_OK) return rc;
    }
    h->batch_cap = n_frames;
    return OVS_OK;
}

int ensure_rect_mask(ovs_orb* h, int width, int height) {
    // orb_extractor.cc:82-85, :248-261.  The filled rectangle covers the closed box
    // [x_min, x_max] x [y_min, y_max]; the anti-aliased rim cv::rectangle adds is non-zero and
    // cannot be seen by the `== 0` test.
    if (h->mask_is_initialized || h->mask_rects.empty()) {
        return OVS_OK;
    }
    std::vector<uint8_t> m(static_cast<size_t>(width) * height, 255);
    const unsigned cols = width, rows = height;
    for (size_t r = 0; r + 3 < h->mask_rects.size(); r += 4) {
        const unsigned int x_min = std::round(cols * h->mask_rects[r]);
        const unsigned int x_max = std::round(cols * h->mask_rects[r + 1]);
        const unsigned int y_min = std::round(rows * h->mask_rects[r + 2]);
        const unsigned int y_max = std::round(rows * h->mask_rects[r + 3]);
        const int x0 = std::max(static_cast<int>(x_min), 0), x1 = std::min(static_cast<int>(x_max), width - 1);
        const int y0 = std::max(static_cast<int>(y_min), 0), y1 = std::min(static_cast<int>(y_max), height - 1);
        for (int y = y0; y <= y1; ++y) {
            for (int x = x0; x <= x1; ++x) {
                m[static_cast<size_t>(y) * width + x] = 0;
            }
        }
    }
    int rc = h->rect_mask.reserve(m.size());
    if (rc != OVS_OK) {
        return rc;
    }
    OVS_HIP(hipMemcpy(h->rect_mask.ptr, m.data(), m.size(), hipMemcpyHostToDevice));
    h->rect_mask_w = width;
    h->rect_mask_h = height;
    h->mask_is_initialized = true;
    return OVS_OK;
}

int upload_constants(ovs_orb* h) {
    if (h->constants_uploaded) {
        return OVS_OK;
    }
    std::vector<float> pat(4 * 16 * 17, 0.f);
    for (int t = 0; t < 256; ++t) {
        for (int k = 0; k < 4; ++k) {
            pat[4 * ((t >> 4) * 17 + (t & 15)) + k] = static_cast<float>(ovs_orb_pattern[4 * t + k]);
        }
    }
    std::vector<uint32_t> masks(4 * 17 * 4 * 3, 0u);
    for (int av = 0; av < 17; ++av) {
        const int dmax = av <= kHalfPatch ? h->u_max[av] : -1;
        for (int mis = 0; mis < 4; ++mis) {
            for (int part = 0; part < 3; ++part) {
                for (int d = 0; d < 3; ++d) {
                    // window byte wb = 12 part + 4 d + b of a row whose first byte lies `mis` bytes left
                    // of u = -16 has u = wb - 16 - mis
                    uint32_t m = 0;
                    for (int b = 0; b < 4; ++b) {
                        const int u = 12 * part + 4 * d + b - 16 - mis;
                        m |= (std::abs(u) <= dmax) ? (0xFFu << (8 * b)) : 0u;
                    }
                    masks[4 * ((av * 4 + mis) * 3 + part) + d] = m;
                }
            }
        }
    }
    OVS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_pat), pat.data(), pat.size() * sizeof(float)));
    OVS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_disc_mask), masks.data(), masks.size() * sizeof(uint32_t)));
    h->constants_uploaded = true;
    return OVS_OK;
}

int validate_rects(const ovs_orb_params* params) {
    // orb_params.cc:14-24
    for (uint32_t i = 0; i < params->num_mask_rects; ++i) {
        const float* v = params->mask_rects + 4 * i;
        if (v[0] >= v[1]) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "x_max must be greater than x_min");
        }
        if (v[2] >= v[3]) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "y_max must be greater than x_min");
        }
    }
    return OVS_OK;
}

} // namespace

extern "C" {

int ovs_orb_create(const ovs_orb_params* params, int device, ovs_orb** out) {
    if (!params || !out) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null argument");
    }
    if (params->num_levels == 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "num_levels must be positive");
    }
    if (params->num_mask_rects && !params->mask_rects) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "mask_rects is null");
    }
    int rc = validate_rects(params);
    if (rc != OVS_OK) {
        return rc;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        return ovs::fail(OVS_ERR_HIP, "no HIP device available (this library has no CPU path)");
    }
    if (device < 0 || device >= ndev) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    }
    auto* h = new ovs_orb();
    h->device = device;
    h->max_num_keypts = params->max_num_keypts;
    h->scale_factor = params->scale_factor;
    h->num_levels = params->num_levels;
    h->ini_fast_thr = params->ini_fast_thr;
    h->min_fast_thr = params->min_fast_thr;
    h->mask_rects.assign(params->mask_rects, params->mask_rects + 4 * static_cast<size_t>(params->num_mask_rects));
    orb_initialize(h);
    *out = h;
    return OVS_OK;
}

void ovs_orb_destroy(ovs_orb* h) {
    if (!h) {
        return;
    }
    ovs::device_guard guard(h->device);
    for (ovs::dev_buffer* b : {&h->tabs, &h->cell_tab, &h->rect_mask, &h->pyr, &h->blur, &h->slots, &h->cell_counts, &h->cand,
                               &h->lvl_xy, &h->lvl_resp, &h->lvl_count, &h->st_img, &h->st_mask, &h->st_out}) {
        b->release();
    }
    h->pin_in.release();
    h->pin_out.release();
    if (h->own_stream) {
        (void)hipStreamDestroy(h->own_stream);
    }
    if (h->side_stream) {
        (void)hipStreamDestroy(h->side_stream);
    }
    if (h->ev_fork) {
        (void)hipEventDestroy(h->ev_fork);
    }
    if (h->ev_join) {
        (void)hipEventDestroy(h->ev_join);
    }
    delete h;
}

uint32_t ovs_orb_get_max_num_keypoints(const ovs_orb* h) { return h->max_num_keypts; }
int ovs_orb_set_max_num_keypoints(ovs_orb* h, uint32_t v) {
    h->max_num_keypts = v;
    orb_initialize(h);
    return OVS_OK;
}
float ovs_orb_get_scale_factor(const ovs_orb* h) { return h->scale_factor; }
int ovs_orb_set_scale_factor(ovs_orb* h, float v) {
    h->scale_factor = v;
    orb_initialize(h);
    return OVS_OK;
}
uint32_t ovs_orb_get_num_scale_levels(const ovs_orb* h) { return h->num_levels; }
int ovs_orb_set_num_scale_levels(ovs_orb* h, uint32_t v) {
    if (v == 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "num_levels must be positive");
    }
    h->num_levels = v;
    orb_initialize(h);
    return OVS_OK;
}
uint32_t ovs_orb_get_initial_fast_threshold(const ovs_orb* h) { return h->ini_fast_thr; }
int ovs_orb_set_initial_fast_threshold(ovs_orb* h, uint32_t v) {
    h->ini_fast_thr = v; // no re-initialisation, orb_extractor.cc:177-179
    h->plan.ini_thr = static_cast<int>(v);
    return OVS_OK;
}
int ovs_orb_get_blur_kernel(const ovs_orb* h) { return h->blur_kernel; }
int ovs_orb_set_blur_kernel(ovs_orb* h, int kernel_variant) {
    if (!h || (kernel_variant != OVS_BLUR_ROUNDED && kernel_variant != OVS_BLUR_NORMALISED)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "unknown blur kernel variant");
    }
    h->blur_kernel = kernel_variant;
    return OVS_OK;
}
uint32_t ovs_orb_get_minimum_fast_threshold(const ovs_orb* h) { return h->min_fast_thr; }
int ovs_orb_set_minimum_fast_threshold(ovs_orb* h, uint32_t v) {
    h->min_fast_thr = v;
    h->plan.min_thr = static_cast<int>(v);
    return OVS_OK;
}
int ovs_orb_get_scale_factors(const ovs_orb* h, float* out) {
    std::copy(h->scale_factors.begin(), h->scale_factors.end(), out);
    return OVS_OK;
}
int ovs_orb_get_inv_scale_factors(const ovs_orb* h, float* out) {
    std::copy(h->inv_scale_factors.begin(), h->inv_scale_factors.end(), out);
    return OVS_OK;
}
int ovs_orb_get_level_sigma_sq(const ovs_orb* h, float* out) {
    std::copy(h->level_sigma_sq.begin(), h->level_sigma_sq.end(), out);
    return OVS_OK;
}
int ovs_orb_get_inv_level_sigma_sq(const ovs_orb* h, float* out) {
    std::copy(h->inv_level_sigma_sq.begin(), h->inv_level_sigma_sq.end(), out);
    return OVS_OK;
}
int ovs_orb_get_num_keypts_per_level(const ovs_orb* h, uint32_t* out) {
    std::copy(h->num_keypts_per_level.begin(), h->num_keypts_per_level.end(), out);
    return OVS_OK;
}

int ovs_orb_level_size(const ovs_orb* h, int width, int height, int level, int* w, int* hh) {
    if (level < 0 || level >= static_cast<int>(h->num_levels)) {
        return ovs::fail(OVS_ERR_OUT_OF_RANGE, "level out of range");
    }
    const double scale = h->scale_factors.at(level);
    *w = level == 0 ? width : static_cast<int>(std::round(width * 1.0 / scale));
    *hh = level == 0 ? height : static_cast<int>(std::round(height * 1.0 / scale));
    return OVS_OK;
}

int ovs_orb_keypoint_capacity(const ovs_orb* h, int width, int height) {
    int total = 0;
    for (uint32_t l = 0; l < h->num_levels; ++l) {
        int w, hh;
        ovs_orb_level_size(h, width, height, static_cast<int>(l), &w, &hh);
        if (w <= 2 * kBorder || hh <= 2 * kBorder) {
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "image too small for the pyramid");
        }
        const double ratio = static_cast<double>(w - 2 * kBorder) / (hh - 2 * kBorder);
        const int n0 = ratio > 1 ? static_cast<int>(std::round(ratio)) : static_cast<int>(std::round(1 / ratio));
        total += std::max(2 * static_cast<int>(h->num_keypts_per_level.at(l)), 4 * n0) + 8;
    }
    return total;
}

int ovs_orb_extract_batch_dev(ovs_orb* h, const uint8_t* d_imgs, int n_frames, int width, int height,
                              size_t row_stride, size_t frame_stride,
                              const uint8_t* d_mask, size_t mask_stride,
                              ovs_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap,
                              void* stream_) {
    if (!h) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null handle");
    }
    if (!d_imgs || width <= 0 || height <= 0 || n_frames <= 0) {
        return OVS_OK; // empty image: outputs untouched (orb_extractor.cc:70-72)
    }
    if (!d_kps || !d_desc || !d_counts || cap <= 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null output buffer");
    }
    if (row_stride < static_cast<size_t>(width)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "row stride smaller than the width");
    }
    if (row_stride >= (1u << 23) || static_cast<unsigned long long>(height - 1) * row_stride + static_cast<unsigned long long>(width) > (1ull << 31)) {
        // the blur and descriptor kernels form row offsets inside a frame with 24-bit / 32-bit products
        return ovs::fail(OVS_ERR_UNSUPPORTED, "row stride of 2^23 bytes or more, or a frame that spans more than 2^31 bytes, is not supported");
    }
    ovs::device_guard guard(h->device);
    if (!guard.ok) {
        return ovs::fail(OVS_ERR_HIP, "cannot select the handle's device");
    }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int rc;
    if ((rc = build_plan(h, width, height)) != OVS_OK) return rc;
    if ((rc = ensure_buffers(h, n_frames)) != OVS_OK) return rc;
    if ((rc = upload_constants(h)) != OVS_OK) return rc;
    if ((rc = ensure_rect_mask(h, width, height)) != OVS_OK) return rc;
    const plan_dev& p = h->plan;

    image_src src;
    src.img0 = d_imgs;
    src.row_stride0 = static_cast<long long>(row_stride);
    src.frame_stride0 = static_cast<long long>(frame_stride);
    src.pyr = h->pyr.as<uint8_t>();
    src.mask = nullptr;
    src.mask_stride = 0;
    src.mask_w = width;
    src.mask_h = height;
    if (d_mask) {
        src.mask = d_mask;
        src.mask_stride = static_cast<long long>(mask_stride);
    }
    else if (h->mask_is_initialized) {
        // the rectangle mask keeps the size of the first image it was created for
        src.mask = h->rect_mask.as<uint8_t>();
        src.mask_stride = h->rect_mask_w;
        src.mask_w = h->rect_mask_w;
        src.mask_h = h->rect_mask_h;
    }

    for (int l = 1; l < p.num_levels; ++l) {
        const level_geom& g = p.lv[l];
        if (g.resize_rows) {
            // (a single frame: 8.9 -> 4.9 us per level with four rows per tile, the seven launches are a chain)
            const int rows = n_frames <= kLiveFrames ? kResizeRowsLive : kResizeRows;
            const int tiles_x = (g.w + 255) / 256, tiles = tiles_x * ((g.h + rows - 1) / rows);
            OVS_LAUNCH("k_resize_rows", k_resize_rows, dim3(xcd_grid((tiles + 3) / 4 * n_frames)), dim3(256), 0, stream, p, src, h->tabs.as<int>(), l, tiles_x, tiles, n_frames,
                       rows);
        }
        else {
            dim3 grid((g.w + 255) / 256, (g.h + 3) / 4, n_frames), block(64, 4, 1);
            OVS_LAUNCH("k_resize_linear", k_resize_linear, grid, block, 0, stream, p, src, h->tabs.as<int>(), l);
        }
    }
    // FAST (vector-issue bound) and the quadtree (latency bound) on the caller's stream, the blur
    // (bandwidth bound) beside them on the handle's second stream.  While per-kernel timing is on,
    // everything stays on one stream so that each launch is measured alone.
    // OVS_SERIAL_EXTRACT=1 in the environment keeps them on one stream as well (profiler runs that
    // want every kernel's duration undisturbed).
    static const bool serial_env = [] {
        const char* e = std::getenv("OVS_SERIAL_EXTRACT");
        return e && e[0] == '1';
    }();
    const bool forked = !ovs::timing_enabled() && !serial_env;
    hipStream_t blur_stream = stream;
    if (forked) {
        if (!h->side_stream) {
            OVS_HIP(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            OVS_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            OVS_HIP(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        blur_stream = h->side_stream;
        OVS_HIP(hipEventRecord(h->ev_fork, stream));
        OVS_HIP(hipStreamWaitEvent(blur_stream, h->ev_fork, 0));
    }
    if (h->blur_kernel == OVS_BLUR_NORMALISED) {
        OVS_LAUNCH("k_gauss7", k_gauss7<OVS_BLUR_NORMALISED>, dim3(xcd_grid((p.total_blur_tiles + 3) / 4 * n_frames)), dim3(256), 0, blur_stream,
                   p, src, h->blur.as<uint8_t>(), n_frames);
    }
    else {
        OVS_LAUNCH("k_gauss7", k_gauss7<OVS_BLUR_ROUNDED>, dim3(xcd_grid((p.total_blur_tiles + 3) / 4 * n_frames)), dim3(256), 0, blur_stream,
                   p, src, h->blur.as<uint8_t>(), n_frames);
    }
    if (forked) {
        OVS_HIP(hipEventRecord(h->ev_join, blur_stream));
    }
    for (int f0 = 0; f0 < n_frames; f0 += 65535) { // the y extent of a grid is limited to 65535
        OVS_LAUNCH("k_fast_cells", k_fast_cells, dim3(xcd_grid(p.total_cells), std::min(n_frames - f0, 65535)), dim3(256), 0, stream,
                   p, src, h->cell_tab.as<fast_cell_rec>(), h->slots.as<u32>(), h->cell_counts.as<u32>(), f0);
    }
    {
        // levels whose node arrays fit in LDS in one launch (LDS sized for the largest of them), the others
        // (quotas beyond about 1130 keypoints per level) in a launch that keeps them in a global workspace
        const int first_lds = quadtree_first_lds_level(p);
        const int qt_threads = n_frames <= kLiveFrames ? kQtThreadsLive : kQtThreads;
        u32* cand_xy = h->cand.as<u32>();
        u32* cand_node = h->cand.as<u32>() + static_cast<long long>(n_frames) * p.frame_cand_cap;
        if (first_lds > 0) {
            int ncap = 0;
            for (int l = 0; l < first_lds; ++l) {
                ncap = std::max(ncap, p.lv[l].ncap);
            }
            const size_t lds = static_cast<size_t>(p.max_cells + 1) * 4 + (16 + 8) * 4;
            OVS_LAUNCH("k_quadtree", k_quadtree<false>, dim3(n_frames, first_lds), dim3(qt_threads), lds, stream,
                       p, h->slots.as<u32>(), h->cell_counts.as<u32>(), cand_xy, cand_node,
                       h->lvl_xy.as<u32>(), h->lvl_resp.as<uint8_t>(), h->lvl_count.as<int>(), 0, ncap, h->qt_ws.as<u32>());
        }
        if (first_lds < p.num_levels) {
            int ncap = 0;
            for (int l = first_lds; l < p.num_levels; ++l) {
                ncap = std::max(ncap, p.lv[l].ncap);
            }
            OVS_LAUNCH("k_quadtree", k_quadtree<true>, dim3(n_frames, p.num_levels - first_lds), dim3(qt_threads), quadtree_lds_bytes(p, ncap), stream,
                       p, h->slots.as<u32>(), h->cell_counts.as<u32>(), cand_xy, cand_node,
                       h->lvl_xy.as<u32>(), h->lvl_resp.as<uint8_t>(), h->lvl_count.as<int>(), first_lds, ncap, static_cast<u32*>(nullptr));
        }
    }
    if (forked) {
        OVS_HIP(hipStreamWaitEvent(stream, h->ev_join, 0));
    }
    const int od_blocks = std::max((std::min(p.kp_total_cap, cap) + 15) / 16, 1);
    OVS_LAUNCH("k_orient_desc", k_orient_desc, dim3(xcd_grid(static_cast<long long>(od_blocks) * n_frames)), dim3(256), 0, stream,
               p, src, h->blur.as<uint8_t>(), h->lvl_xy.as<u32>(), h->lvl_resp.as<uint8_t>(), h->lvl_count.as<int>(),
               d_kps, d_desc, d_counts, cap, od_blocks, n_frames);
    OVS_HIP(hipGetLastError());

    h->last_img0 = d_imgs;
    h->last_row_stride0 = src.row_stride0;
    h->last_frame_stride0 = src.frame_stride0;
    h->last_frames = n_frames;
    return OVS_OK;
}

int ovs_orb_extract(ovs_orb* h, const uint8_t* img, int width, int height, size_t stride,
                    const uint8_t* mask, size_t mask_stride,
                    ovs_keypoint* kps, uint8_t* desc, int cap, int* n) {
    if (!h) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null handle");
    }
    if (!img || width <= 0 || height <= 0) {
        return OVS_OK; // empty image: outputs untouched
    }
    if (!kps || !desc || !n || cap <= 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null output buffer");
    }
    ovs::device_guard guard(h->device);
    if (!guard.ok) {
        return ovs::fail(OVS_ERR_HIP, "cannot select the handle's device");
    }
    int rc;
    if (!h->own_stream) {
        OVS_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    }
    const int dcap = ovs_orb_keypoint_capacity(h, width, height);
    if (dcap < 0) {
        return dcap;
    }
    // Image (and mask) rows are packed into page-locked staging memory and sent with one DMA each; the
    // device copy keeps the packed layout (row pitch = width: level 0 may have any pitch).
    const size_t w = static_cast<size_t>(width), img_bytes = w * height;
    const size_t kp_bytes = static_cast<size_t>(dcap) * sizeof(ovs_keypoint), desc_bytes = static_cast<size_t>(dcap) * 32;
    if ((rc = h->st_img.reserve(img_bytes + 256)) != OVS_OK) return rc;
    if ((rc = h->st_out.reserve(256 + kp_bytes + desc_bytes)) != OVS_OK) return rc;
    if ((rc = h->pin_in.reserve((mask ? 2 : 1) * img_bytes)) != OVS_OK) return rc;
    if ((rc = h->pin_out.reserve(256 + kp_bytes + desc_bytes)) != OVS_OK) return rc;
    auto pack = [&](uint8_t* dst, const uint8_t* src, size_t src_stride) {
        if (src_stride == w) {
            std::memcpy(dst, src, img_bytes);
        }
        else {
            for (int y = 0; y < height; ++y) {
                std::memcpy(dst + w * y, src + src_stride * y, w);
            }
        }
    };
    pack(h->pin_in.as<uint8_t>(), img, stride);
    OVS_HIP(hipMemcpyAsync(h->st_img.ptr, h->pin_in.ptr, img_bytes, hipMemcpyHostToDevice, h->own_stream));
    const uint8_t* d_mask = nullptr;
    if (mask) {
        if ((rc = h->st_mask.reserve(img_bytes + 256)) != OVS_OK) return rc;
        pack(h->pin_in.as<uint8_t>() + img_bytes, mask, mask_stride);
        OVS_HIP(hipMemcpyAsync(h->st_mask.ptr, h->pin_in.as<uint8_t>() + img_bytes, img_bytes, hipMemcpyHostToDevice, h->own_stream));
        d_mask = h->st_mask.as<uint8_t>();
    }
    uint8_t* d_out = h->st_out.as<uint8_t>();
    auto* d_count = reinterpret_cast<int32_t*>(d_out);
    auto* d_kps = reinterpret_cast<ovs_keypoint*>(d_out + 256);
    uint8_t* d_desc = d_out + 256 + kp_bytes;
    rc = ovs_orb_extract_batch_dev(h, h->st_img.as<uint8_t>(), 1, width, height, w, img_bytes, d_mask, w, d_kps, d_desc, d_count, dcap, h->own_stream);
    if (rc != OVS_OK) {
        return rc;
    }
    // One round trip for the usual case: the count and as many records as a full quota plus a margin
    // come back together; a frame with more keypoints fetches the remainder afterwards.
    uint8_t* p_out = h->pin_out.as<uint8_t>();
    const size_t guess = std::min<size_t>(static_cast<size_t>(dcap), static_cast<size_t>(h->max_num_keypts) + h->max_num_keypts / 4 + 64);
    OVS_HIP(hipMemcpyAsync(p_out, d_out, 256 + guess * sizeof(ovs_keypoint), hipMemcpyDeviceToHost, h->own_stream));
    OVS_HIP(hipMemcpyAsync(p_out + 256 + kp_bytes, d_desc, guess * 32, hipMemcpyDeviceToHost, h->own_stream));
    OVS_HIP(hipStreamSynchronize(h->own_stream));
    const int count = *reinterpret_cast<const int32_t*>(p_out);
    if (count < 0) {
        return ovs::fail(OVS_ERR_CAPACITY, "internal keypoint capacity exceeded");
    }
    *n = count;
    if (count > cap) {
        return ovs::fail(OVS_ERR_CAPACITY, "keypoint buffer too small");
    }
    if (static_cast<size_t>(count) > guess) {
        const size_t rest = static_cast<size_t>(count) - guess;
        OVS_HIP(hipMemcpyAsync(p_out + 256 + guess * sizeof(ovs_keypoint), d_kps + guess, rest * sizeof(ovs_keypoint), hipMemcpyDeviceToHost, h->own_stream));
        OVS_HIP(hipMemcpyAsync(p_out + 256 + kp_bytes + guess * 32, d_desc + guess * 32, rest * 32, hipMemcpyDeviceToHost, h->own_stream));
        OVS_HIP(hipStreamSynchronize(h->own_stream));
    }
    if (count > 0) {
        std::memcpy(kps, p_out + 256, static_cast<size_t>(count) * sizeof(ovs_keypoint));
        std::memcpy(desc, p_out + 256 + kp_bytes, static_cast<size_t>(count) * 32);
    }
    return OVS_OK;
}

int ovs_orb_pyramid_level_dev(ovs_orb* h, int frame, int level,
                              const uint8_t** d_ptr, int* width, int* height, size_t* stride) {
    if (!h || h->last_frames == 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "no extraction has run on this handle");
    }
    if (frame < 0 || frame >= h->last_frames || level < 0 || level >= h->plan.num_levels) {
        return ovs::fail(OVS_ERR_OUT_OF_RANGE, "frame or level out of range");
    }
    const level_geom& g = h->plan.lv[level];
    *width = g.w;
    *height = g.h;
    if (level == 0) {
        *d_ptr = h->last_img0 + frame * h->last_frame_stride0;
        *stride = static_cast<size_t>(h->last_row_stride0);
    }
    else {
        *d_ptr = h->pyr.as<uint8_t>() + frame * h->plan.frame_pyr_bytes + g.pyr_off;
        *stride = static_cast<size_t>(g.pitch);
    }
    return OVS_OK;
}

int ovs_orb_download_pyramid_level(ovs_orb* h, int frame, int level, uint8_t* out, size_t out_stride) {
    const uint8_t* d = nullptr;
    int w = 0, hh = 0;
    size_t stride = 0;
    int rc = ovs_orb_pyramid_level_dev(h, frame, level, &d, &w, &hh, &stride);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(h->device);
    OVS_HIP(hipDeviceSynchronize());
    OVS_HIP(hipMemcpy2D(out, out_stride, d, stride, w, hh, hipMemcpyDeviceToHost));
    return OVS_OK;
}

int ovs_orb_download_blurred_level(ovs_orb* h, int frame, int level, uint8_t* out, size_t out_stride) {
    if (!h || h->last_frames == 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "no extraction has run on this handle");
    }
    if (frame < 0 || frame >= h->last_frames || level < 0 || level >= h->plan.num_levels) {
        return ovs::fail(OVS_ERR_OUT_OF_RANGE, "frame or level out of range");
    }
    const level_geom& g = h->plan.lv[level];
    ovs::device_guard guard(h->device);
    OVS_HIP(hipDeviceSynchronize());
    OVS_HIP(hipMemcpy2D(out, out_stride, h->blur.as<uint8_t>() + frame * h->plan.frame_blur_bytes + g.blur_off, g.pitch,
                        g.w, g.h, hipMemcpyDeviceToHost));
    return OVS_OK;
}

} // extern "C"
