// ovs_pre.hip -- the image preparation that runs before feature::orb_extractor::extract() in the
// reference's drivers, for gfx950, so that a batch of frames can be uploaded once as raw camera
// bytes:
//   util::convert_to_grayscale     src/openvslam/util/image_converter.cc:8-39   (cv::cvtColor 8U)
//   util::equalize_histogram       src/openvslam/util/image_converter.cc:45-61  (5 % .. 95 % stretch)
//   util::stereo_rectifier::rectify src/openvslam/util/stereo_rectifier.cc:56-60 (cv::remap INTER_LINEAR)
//
// All three are byte streams bounded by HBM bandwidth: aligned dword loads and stores, no reuse
// beyond what the caches give the bilinear gather.
//
// Kernels
//   k_gray            4 pixels per lane: 3 or 4 interleaved bytes -> (c0*k0 + c1*k1 + c2*k2 + 2^13) >> 14
//   k_hist8 / k_hist16  per-frame histogram (LDS sub-histogram per wave / global atomics)
//   k_select16        5 % and 95 % order statistics of a 65536-bin histogram
//   k_equalize8       per block: order statistics from the 256-bin histogram, 256-entry table in LDS,
//                     table lookup in place
//   k_equalize16      float scale / shift / round / saturate, 16 bit -> 8 bit
//   k_remap           one output pixel per lane: fixed-point coordinates (1/32 pixel), four guarded
//                     taps, 15-bit weights
#include "ovs_match_common.h"

#include <cmath>
#include <cstring>
#include <new>

using namespace ovsm;

namespace {

// ---- cv::cvtColor(RGB/BGR[A] -> GRAY), 8 bit ----------------------------------------------------
// OpenCV 3.3.1 RGB2Gray<uchar>: coefficients B 1868, G 9617, R 4899 at 14 fractional bits, rounding
// constant 2^13 (modules/imgproc/src/color.cpp, yuv_shift = 14).
constexpr u32 kB2Y = 1868, kG2Y = 9617, kR2Y = 4899;

template <int CN>
__global__ __launch_bounds__(256) void k_gray(const uint8_t* __restrict__ src, int width, int height, long long src_row_stride,
                                              long long src_frame_stride, u32 k0, u32 k1, u32 k2, uint8_t* __restrict__ dst,
                                              long long dst_row_stride, long long dst_frame_stride, int dst_dword_ok) {
    const int y = blockIdx.y, f = blockIdx.z;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= width) {
        return;
    }
    const uint8_t* row = src + f * src_frame_stride + y * src_row_stride;
    uint8_t* out = dst + f * dst_frame_stride + y * dst_row_stride + x0;
    const uint8_t* p = row + (long long)x0 * CN;
    if (x0 + 4 <= width) {
        // 4 * CN bytes starting at p: aligned dwords around them, realigned with v_alignbyte.  The
        // dword after the last needed one is read only when it holds a needed byte.
        const unsigned sh = (unsigned)(reinterpret_cast<uintptr_t>(p) & 3);
        const u32* a = reinterpret_cast<const u32*>(p - sh);
        u32 d[CN + 1];
#pragma unroll
        for (int i = 0; i < CN; ++i) {
            d[i] = a[i];
        }
        d[CN] = sh ? a[CN] : 0u;
        u32 w[CN];
#pragma unroll
        for (int i = 0; i < CN; ++i) {
            w[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
        }
        u32 g[4];
        if (CN == 3) {
            const u32 b[12] = {w[0] & 255, (w[0] >> 8) & 255, (w[0] >> 16) & 255, w[0] >> 24, w[1] & 255, (w[1] >> 8) & 255,
                               (w[1] >> 16) & 255, w[1] >> 24, w[2] & 255, (w[2] >> 8) & 255, (w[2] >> 16) & 255, w[2] >> 24};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                g[i] = (b[3 * i] * k0 + b[3 * i + 1] * k1 + b[3 * i + 2] * k2 + (1u << 13)) >> 14;
            }
        }
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 v = w[i % CN];
                g[i] = ((v & 255) * k0 + ((v >> 8) & 255) * k1 + ((v >> 16) & 255) * k2 + (1u << 13)) >> 14;
            }
        }
        if (dst_dword_ok) {
            *reinterpret_cast<u32*>(out) = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
        }
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out[i] = (uint8_t)g[i];
            }
        }
    }
    else {
        for (int i = 0; x0 + i < width; ++i) {
            out[i] = (uint8_t)((p[CN * i] * k0 + p[CN * i + 1] * k1 + p[CN * i + 2] * k2 + (1u << 13)) >> 14);
        }
    }
}

// ---- equalize_histogram ---------------------------------------------------------------------------
constexpr int kHistRowsPerBlock = 16;

// bytes [row, row + width) seen as the aligned dwords that cover them
struct dword_span {
    const u32* base;
    int first, ndw; // first valid byte inside dword 0; number of dwords
};
__device__ __forceinline__ dword_span span_of(const uint8_t* row, int width) {
    const int sh = (int)(reinterpret_cast<uintptr_t>(row) & 3);
    dword_span s;
    s.base = reinterpret_cast<const u32*>(row - sh);
    s.first = sh;
    s.ndw = (sh + width + 3) >> 2;
    return s;
}

__global__ __launch_bounds__(256) void k_hist8(const uint8_t* __restrict__ img, int width, int height, long long row_stride, long long frame_stride,
                                               u32* __restrict__ hist /* [frames][256] */) {
    __shared__ u32 sub[4][256];
    const int tid = threadIdx.x, wave = tid >> 6, f = blockIdx.y;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sub[i][tid] = 0;
    }
    __syncthreads();
    const int y0 = blockIdx.x * kHistRowsPerBlock, y1 = min(y0 + kHistRowsPerBlock, height);
    for (int y = y0; y < y1; ++y) {
        const dword_span s = span_of(img + f * frame_stride + y * row_stride, width);
        for (int k = tid; k < s.ndw; k += 256) {
            const u32 v = s.base[k];
            const int lo = k == 0 ? s.first : 0;
            const int hi = min(4, s.first + width - 4 * k);
            for (int j = lo; j < hi; ++j) {
                atomicAdd(&sub[wave][(v >> (8 * j)) & 255], 1u);
            }
        }
    }
    __syncthreads();
    const u32 total = sub[0][tid] + sub[1][tid] + sub[2][tid] + sub[3][tid];
    if (total) {
        atomicAdd(&hist[(size_t)f * 256 + tid], total);
    }
}

// sorted[k] of the histogrammed values = smallest v with cumulative count > k
__device__ __forceinline__ void block_order_stats_256(const u32* __restrict__ hist, u32 k_lo, u32 k_hi, u32* scan /* LDS 256 */, int* sel /* LDS 2 */) {
    const int tid = threadIdx.x;
    scan[tid] = hist[tid];
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const u32 t = tid >= d ? scan[tid - d] : 0u;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    const u32 incl = scan[tid], excl = incl - hist[tid];
    if (excl <= k_lo && k_lo < incl) {
        sel[0] = tid;
    }
    if (excl <= k_hi && k_hi < incl) {
        sel[1] = tid;
    }
    __syncthreads();
}

// convertTo(CV_8U, alpha, beta) of OpenCV 3.3.1 for 8U / 16U sources: float multiply, float add (not
// fused), round half to even, saturate.  alpha = 255.0 / (h - l), beta = -255.0 * l / (h - l) are
// evaluated in double and converted to float (image_converter.cc:52,59).
__device__ __forceinline__ u32 stretch(float v, float alpha, float beta) {
    const float r = __fadd_rn(__fmul_rn(v, alpha), beta);
    const int i = __float2int_rn(r);
    return (u32)min(max(i, 0), 255);
}

__global__ __launch_bounds__(256) void k_equalize8(uint8_t* __restrict__ img, int width, int height, long long row_stride, long long frame_stride,
                                                   const u32* __restrict__ hist, u32 k_lo, u32 k_hi, int* __restrict__ flat_flags) {
    __shared__ u32 scan[256];
    __shared__ int sel[2];
    __shared__ uint8_t lut[256];
    const int tid = threadIdx.x, f = blockIdx.y;
    block_order_stats_256(hist + (size_t)f * 256, k_lo, k_hi, scan, sel);
    const int l = sel[0], h = sel[1];
    if (h == l) {
        // 255.0 / 0: the reference's result is not defined; the frame is left as it is and flagged
        if (blockIdx.x == 0 && tid == 0) {
            flat_flags[f] = 1;
        }
        return;
    }
    const float alpha = (float)__ddiv_rn(255.0, (double)(h - l));
    const float beta = (float)__ddiv_rn(__dmul_rn(-255.0, (double)l), (double)(h - l));
    lut[tid] = (uint8_t)stretch((float)tid, alpha, beta);
    __syncthreads();
    const int y0 = blockIdx.x * kHistRowsPerBlock, y1 = min(y0 + kHistRowsPerBlock, height);
    for (int y = y0; y < y1; ++y) {
        uint8_t* row = img + f * frame_stride + y * row_stride;
        const dword_span s = span_of(row, width);
        u32* base = const_cast<u32*>(s.base);
        for (int k = tid; k < s.ndw; k += 256) {
            const u32 v = base[k];
            const int lo = k == 0 ? s.first : 0;
            const int hi = min(4, s.first + width - 4 * k);
            if (lo == 0 && hi == 4) {
                base[k] = (u32)lut[v & 255] | ((u32)lut[(v >> 8) & 255] << 8) | ((u32)lut[(v >> 16) & 255] << 16) | ((u32)lut[v >> 24] << 24);
            }
            else {
                // a dword shared with the neighbouring row (or with bytes that are not pixels):
                // only this row's bytes are written
                uint8_t* b = reinterpret_cast<uint8_t*>(base + k);
                for (int j = lo; j < hi; ++j) {
                    b[j] = lut[(v >> (8 * j)) & 255];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_hist16(const uint16_t* __restrict__ img, int width, int height, long long row_stride, long long frame_stride,
                                                u32* __restrict__ hist /* [frames][65536] */) {
    const int f = blockIdx.y;
    const int y0 = blockIdx.x * kHistRowsPerBlock, y1 = min(y0 + kHistRowsPerBlock, height);
    u32* hf = hist + (size_t)f * 65536;
    for (int y = y0; y < y1; ++y) {
        const uint16_t* row = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(img) + f * frame_stride + y * row_stride);
        for (int x = threadIdx.x; x < width; x += 256) {
            atomicAdd(&hf[row[x]], 1u);
        }
    }
}

__global__ __launch_bounds__(256) void k_select16(const u32* __restrict__ hist, u32 k_lo, u32 k_hi, int* __restrict__ lh /* [frames][2] */) {
    __shared__ u32 part[256];
    const int tid = threadIdx.x, f = blockIdx.x;
    const u32* hf = hist + (size_t)f * 65536;
    u32 sum = 0;
    for (int i = 0; i < 256; ++i) {
        sum += hf[tid * 256 + i];
    }
    part[tid] = sum;
    __syncthreads();
    u32 before = 0;
    for (int i = 0; i < tid; ++i) {
        before += part[i];
    }
    const u32 ks[2] = {k_lo, k_hi};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (before <= ks[s] && ks[s] < before + sum) {
            u32 run = before;
            for (int i = 0; i < 256; ++i) {
                run += hf[tid * 256 + i];
                if (ks[s] < run) {
                    lh[2 * f + s] = tid * 256 + i;
                    break;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_equalize16(const uint16_t* __restrict__ img, int width, int height, long long row_stride, long long frame_stride,
                                                    const int* __restrict__ lh, uint8_t* __restrict__ dst, long long dst_row_stride,
                                                    long long dst_frame_stride, int* __restrict__ flat_flags) {
    const int f = blockIdx.y;
    const int l = lh[2 * f], h = lh[2 * f + 1];
    const bool flat = h == l;
    if (flat && blockIdx.x == 0 && threadIdx.x == 0) {
        flat_flags[f] = 1;
    }
    const float alpha = flat ? 0.0f : (float)__ddiv_rn(255.0, (double)(h - l));
    const float beta = flat ? 0.0f : (float)__ddiv_rn(__dmul_rn(-255.0, (double)l), (double)(h - l));
    const int y0 = blockIdx.x * kHistRowsPerBlock, y1 = min(y0 + kHistRowsPerBlock, height);
    for (int y = y0; y < y1; ++y) {
        const uint16_t* row = reinterpret_cast<const uint16_t*>(reinterpret_cast<const uint8_t*>(img) + f * frame_stride + y * row_stride);
        uint8_t* out = dst + f * dst_frame_stride + y * dst_row_stride;
        for (int x = threadIdx.x; x < width; x += 256) {
            out[x] = (uint8_t)stretch((float)row[x], alpha, beta);
        }
    }
}

// ---- cv::remap(INTER_LINEAR, BORDER_CONSTANT 0) with CV_32FC1 maps --------------------------------
// OpenCV converts the float maps to fixed point first (modules/imgproc/src/imgwarp.cpp, remap):
// s = cvRound(map * 32), integer part s >> 5 saturated to int16, fraction s & 31.  The bilinear
// weights of fraction (fx, fy) are (32 - fx)(32 - fy) * 32, fx (32 - fy) * 32, (32 - fx) fy * 32 and
// fx fy * 32 (sum 2^15; the table entry for fx = fy = 0 is stored as 32767 + a stray 1 on the
// opposite tap, which yields the same pixel as 32768 for every 8-bit input), taps outside the
// source read the border value 0, and the sum is rounded with (v + 2^14) >> 15.
__global__ __launch_bounds__(256) void k_remap(const uint8_t* __restrict__ src, int src_w, int src_h, long long src_row_stride, long long src_frame_stride,
                                               const int* __restrict__ map_xy, const uint16_t* __restrict__ map_frac, int width, int height,
                                               uint8_t* __restrict__ dst, long long dst_row_stride, long long dst_frame_stride) {
    // one output pixel per lane, pixels of a frame numbered row by row: neighbouring lanes read
    // neighbouring map entries and (the warp being smooth) neighbouring source pixels
    const int f = blockIdx.y;
    const unsigned m = blockIdx.x * 256u + threadIdx.x;
    if (m >= (unsigned)width * (unsigned)height) {
        return;
    }
    const int y = (int)(m / (unsigned)width), x = (int)(m - (unsigned)y * (unsigned)width);
    const uint8_t* S = src + f * src_frame_stride;
    const int xy = map_xy[m];
    const u32 fr = map_frac[m];
    const int sx = (short)(xy & 0xFFFF), sy = xy >> 16;
    const u32 fx = fr & 31, fy = fr >> 5;
    const bool x_in0 = (unsigned)sx < (unsigned)src_w, x_in1 = (unsigned)(sx + 1) < (unsigned)src_w;
    const bool y_in0 = (unsigned)sy < (unsigned)src_h, y_in1 = (unsigned)(sy + 1) < (unsigned)src_h;
    const uint8_t* r0 = S + (long long)sy * src_row_stride + sx;
    const uint8_t* r1 = r0 + src_row_stride;
    const u32 v00 = (x_in0 && y_in0) ? r0[0] : 0u;
    const u32 v01 = (x_in1 && y_in0) ? r0[1] : 0u;
    const u32 v10 = (x_in0 && y_in1) ? r1[0] : 0u;
    const u32 v11 = (x_in1 && y_in1) ? r1[1] : 0u;
    const u32 w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    dst[f * dst_frame_stride + y * dst_row_stride + x] = (uint8_t)((v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1u << 14)) >> 15);
}

bool dword_aligned(const void* p, size_t row_stride, size_t frame_stride) {
    return ((reinterpret_cast<uintptr_t>(p) | row_stride | frame_stride) & 3) == 0;
}

// cvRound(v * 32) of the SSE2 and scalar paths: round half to even; NaN and values beyond int32
// become INT_MIN
int fixed_coord(float v) {
    const float s = v * 32.0f;
    if (!(std::fabs(s) < 2147483648.0f)) {
        return INT32_MIN;
    }
    return static_cast<int>(std::lrintf(s));
}

int saturate_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

u32 order_index(double fraction, size_t n) { return static_cast<u32>(static_cast<unsigned int>(fraction * static_cast<double>(n))); }

} // namespace

struct ovs_remap {
    int device = 0;
    int width = 0, height = 0;
    ovs::dev_buffer xy, frac;
};

extern "C" {

int ovs_convert_to_grayscale_dev(const uint8_t* d_src, int n_frames, int width, int height, int channels,
                                 size_t src_row_stride, size_t src_frame_stride, int color_order,
                                 uint8_t* d_dst, size_t dst_row_stride, size_t dst_frame_stride, int device, void* stream_) {
    if (!d_src || !d_dst || n_frames <= 0 || width <= 0 || height <= 0 || n_frames > 65535 || height > 65535
        || src_row_stride < static_cast<size_t>(width) * channels || dst_row_stride < static_cast<size_t>(width)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (channels != 3 && channels != 4) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "convert_to_grayscale takes 3- or 4-channel images (1-channel images are passed on unchanged by the reference)");
    }
    if (color_order != OVS_COLOR_RGB && color_order != OVS_COLOR_BGR) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "color order Gray leaves a multi-channel image as it is in the reference; nothing to convert");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const u32 k0 = color_order == OVS_COLOR_BGR ? kB2Y : kR2Y, k2 = color_order == OVS_COLOR_BGR ? kR2Y : kB2Y;
    const dim3 grid((width + 1023) / 1024, height, n_frames);
    const int dw = dword_aligned(d_dst, dst_row_stride, dst_frame_stride) ? 1 : 0;
    if (channels == 3) {
        OVS_LAUNCH("k_gray", k_gray<3>, grid, dim3(256), 0, st, d_src, width, height, (long long)src_row_stride, (long long)src_frame_stride, k0, kG2Y, k2,
                   d_dst, (long long)dst_row_stride, (long long)dst_frame_stride, dw);
    }
    else {
        OVS_LAUNCH("k_gray", k_gray<4>, grid, dim3(256), 0, st, d_src, width, height, (long long)src_row_stride, (long long)src_frame_stride, k0, kG2Y, k2,
                   d_dst, (long long)dst_row_stride, (long long)dst_frame_stride, dw);
    }
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

int ovs_convert_to_grayscale(const uint8_t* src, int width, int height, int channels, size_t src_stride, int color_order,
                             uint8_t* dst, size_t dst_stride, int device) {
    if (!src || !dst || width <= 0 || height <= 0 || (channels != 3 && channels != 4)) {
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
    const size_t in_row = static_cast<size_t>(width) * channels, in_bytes = in_row * height, out_bytes = static_cast<size_t>(width) * height;
    if ((rc = ctx->arena.reserve(al(in_bytes) + al(out_bytes))) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_in = cv.take<uint8_t>(in_bytes);
    auto* d_out = cv.take<uint8_t>(out_bytes);
    if ((rc = ctx->mirror.reserve(ctx->arena.bytes)) != OVS_OK) {
        return rc;
    }
    OVS_HIP(image_up(ctx, ctx->stream, d_in, src, src_stride, in_row, height));
    if ((rc = ovs_convert_to_grayscale_dev(d_in, 1, width, height, channels, in_row, in_bytes, color_order, d_out, width, out_bytes, device, ctx->stream)) != OVS_OK) {
        return rc;
    }
    OVS_HIP(image_down(ctx, ctx->stream, dst, dst_stride, d_out, width, height));
    return OVS_OK;
}

int ovs_equalize_histogram_dev(uint8_t* d_img, int n_frames, int width, int height, size_t row_stride, size_t frame_stride,
                               int32_t* d_flat_flags, int device, void* stream_) {
    if (!d_img || !d_flat_flags || n_frames <= 0 || n_frames > 65535 || width <= 0 || height <= 0 || row_stride < static_cast<size_t>(width)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    thread_ctx* ctx = ctx_for_stream(device, st);
    const size_t hist_bytes = static_cast<size_t>(n_frames) * 256 * sizeof(u32);
    if ((rc = ctx->arena.reserve(al(hist_bytes))) != OVS_OK) {
        return rc;
    }
    u32* d_hist = ctx->arena.as<u32>();
    OVS_HIP(hipMemsetAsync(d_hist, 0, hist_bytes, st));
    OVS_HIP(hipMemsetAsync(d_flat_flags, 0, static_cast<size_t>(n_frames) * sizeof(int32_t), st));
    const size_t n = static_cast<size_t>(width) * height;
    const dim3 grid((height + kHistRowsPerBlock - 1) / kHistRowsPerBlock, n_frames);
    OVS_LAUNCH("k_hist8", k_hist8, grid, dim3(256), 0, st, d_img, width, height, (long long)row_stride, (long long)frame_stride, d_hist);
    OVS_LAUNCH("k_equalize8", k_equalize8, grid, dim3(256), 0, st, d_img, width, height, (long long)row_stride, (long long)frame_stride, d_hist,
               order_index(0.05, n), order_index(0.95, n), d_flat_flags);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

int ovs_equalize_histogram_u16_dev(const uint16_t* d_img, int n_frames, int width, int height, size_t row_stride, size_t frame_stride,
                                   uint8_t* d_dst, size_t dst_row_stride, size_t dst_frame_stride,
                                   int32_t* d_flat_flags, int device, void* stream_) {
    if (!d_img || !d_dst || !d_flat_flags || n_frames <= 0 || n_frames > 4096 || width <= 0 || height <= 0
        || row_stride < static_cast<size_t>(width) * 2 || (row_stride & 1) || (frame_stride & 1) || (reinterpret_cast<uintptr_t>(d_img) & 1)
        || dst_row_stride < static_cast<size_t>(width)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    thread_ctx* ctx = ctx_for_stream(device, st);
    const size_t hist_bytes = static_cast<size_t>(n_frames) * 65536 * sizeof(u32);
    if ((rc = ctx->arena.reserve(al(hist_bytes) + al(static_cast<size_t>(n_frames) * 2 * sizeof(int)))) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_hist = cv.take<u32>(static_cast<size_t>(n_frames) * 65536);
    auto* d_lh = cv.take<int>(static_cast<size_t>(n_frames) * 2);
    OVS_HIP(hipMemsetAsync(d_hist, 0, hist_bytes, st));
    OVS_HIP(hipMemsetAsync(d_flat_flags, 0, static_cast<size_t>(n_frames) * sizeof(int32_t), st));
    const size_t n = static_cast<size_t>(width) * height;
    const dim3 grid((height + kHistRowsPerBlock - 1) / kHistRowsPerBlock, n_frames);
    OVS_LAUNCH("k_hist16", k_hist16, grid, dim3(256), 0, st, d_img, width, height, (long long)row_stride, (long long)frame_stride, d_hist);
    OVS_LAUNCH("k_select16", k_select16, dim3(n_frames), dim3(256), 0, st, d_hist, order_index(0.05, n), order_index(0.95, n), d_lh);
    OVS_LAUNCH("k_equalize16", k_equalize16, grid, dim3(256), 0, st, d_img, width, height, (long long)row_stride, (long long)frame_stride, d_lh,
               d_dst, (long long)dst_row_stride, (long long)dst_frame_stride, d_flat_flags);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

int ovs_equalize_histogram(const void* src, int width, int height, size_t src_stride, int bits, uint8_t* dst, size_t dst_stride, int device) {
    if (!src || !dst || width <= 0 || height <= 0 || (bits != 8 && bits != 16)) {
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
    // staging lives in this thread's arena; the device entry points use the arena that belongs to
    // (device, internal stream), a different one
    const size_t bpp = bits / 8, in_row = static_cast<size_t>(width) * bpp, in_bytes = in_row * height, out_bytes = static_cast<size_t>(width) * height;
    if ((rc = ctx->arena.reserve(al(in_bytes) + al(out_bytes) + al(sizeof(int32_t)))) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_in = cv.take<uint8_t>(in_bytes);
    auto* d_out = cv.take<uint8_t>(out_bytes);
    auto* d_flag = cv.take<int32_t>(1);
    if ((rc = ctx->mirror.reserve(ctx->arena.bytes)) != OVS_OK) {
        return rc;
    }
    OVS_HIP(image_up(ctx, ctx->stream, d_in, src, src_stride, in_row, height));
    if (bits == 8) {
        rc = ovs_equalize_histogram_dev(d_in, 1, width, height, in_row, in_bytes, d_flag, device, ctx->stream);
    }
    else {
        rc = ovs_equalize_histogram_u16_dev(reinterpret_cast<const uint16_t*>(d_in), 1, width, height, in_row, in_bytes, d_out, width, out_bytes, d_flag,
                                            device, ctx->stream);
    }
    if (rc != OVS_OK) {
        return rc;
    }
    int32_t flag = 0;
    OVS_HIP(hipMemcpyAsync(&flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
    OVS_HIP(image_down(ctx, ctx->stream, dst, dst_stride, bits == 8 ? d_in : d_out, width, height));
    if (flag) {
        return ovs::fail(OVS_ERR_UNSUPPORTED, "the 5 % and 95 % levels of the image coincide: the reference divides by zero here");
    }
    return OVS_OK;
}

int ovs_remap_create(const float* map_x, const float* map_y, int width, int height, int device, ovs_remap** out) {
    if (!map_x || !map_y || width <= 0 || height <= 0 || height > 65535 || !out) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    const size_t n = static_cast<size_t>(width) * height;
    std::vector<int> xy(n);
    std::vector<uint16_t> frac(n);
    for (size_t i = 0; i < n; ++i) {
        const int sx = fixed_coord(map_x[i]), sy = fixed_coord(map_y[i]);
        const int ix = saturate_short(sx >> 5), iy = saturate_short(sy >> 5);
        xy[i] = static_cast<int>((static_cast<uint32_t>(iy) << 16) | (static_cast<uint32_t>(ix) & 0xFFFFu));
        frac[i] = static_cast<uint16_t>(((sy & 31) << 5) | (sx & 31));
    }
    ovs_remap* r = new (std::nothrow) ovs_remap;
    if (!r) {
        return ovs::fail(OVS_ERR_HIP, "out of host memory");
    }
    r->device = device;
    r->width = width;
    r->height = height;
    if ((rc = r->xy.reserve(n * sizeof(int))) != OVS_OK || (rc = r->frac.reserve(n * sizeof(uint16_t))) != OVS_OK) {
        r->xy.release();
        r->frac.release();
        delete r;
        return rc;
    }
    hipError_t e = hipMemcpy(r->xy.ptr, xy.data(), n * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        e = hipMemcpy(r->frac.ptr, frac.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        r->xy.release();
        r->frac.release();
        delete r;
        OVS_HIP(e);
    }
    *out = r;
    return OVS_OK;
}

void ovs_remap_destroy(ovs_remap* r) {
    if (!r) {
        return;
    }
    ovs::device_guard guard(r->device);
    r->xy.release();
    r->frac.release();
    delete r;
}

int ovs_remap_apply_dev(ovs_remap* r, const uint8_t* d_src, int n_frames, int src_width, int src_height,
                        size_t src_row_stride, size_t src_frame_stride,
                        uint8_t* d_dst, size_t dst_row_stride, size_t dst_frame_stride, void* stream_) {
    if (!r || !d_src || !d_dst || n_frames <= 0 || n_frames > 65535 || src_width <= 0 || src_height <= 0
        || src_row_stride < static_cast<size_t>(src_width) || dst_row_stride < static_cast<size_t>(r->width)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    ovs::device_guard guard(r->device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    const dim3 grid(static_cast<unsigned>((static_cast<size_t>(r->width) * r->height + 255) / 256), n_frames);
    OVS_LAUNCH("k_remap", k_remap, grid, dim3(256), 0, st, d_src, src_width, src_height, (long long)src_row_stride, (long long)src_frame_stride,
               r->xy.as<int>(), r->frac.as<uint16_t>(), r->width, r->height, d_dst, (long long)dst_row_stride, (long long)dst_frame_stride);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

int ovs_remap_apply(ovs_remap* r, const uint8_t* src, int src_width, int src_height, size_t src_stride, uint8_t* dst, size_t dst_stride) {
    if (!r || !src || !dst || src_width <= 0 || src_height <= 0) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    ovs::device_guard guard(r->device);
    thread_ctx* ctx = ctx_for(r->device);
    int rc;
    if ((rc = ensure_stream(ctx)) != OVS_OK) {
        return rc;
    }
    const size_t in_bytes = static_cast<size_t>(src_width) * src_height, out_bytes = static_cast<size_t>(r->width) * r->height;
    if ((rc = ctx->arena.reserve(al(in_bytes) + al(out_bytes))) != OVS_OK) {
        return rc;
    }
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_in = cv.take<uint8_t>(in_bytes);
    auto* d_out = cv.take<uint8_t>(out_bytes);
    if ((rc = ctx->mirror.reserve(ctx->arena.bytes)) != OVS_OK) {
        return rc;
    }
    OVS_HIP(image_up(ctx, ctx->stream, d_in, src, src_stride, src_width, src_height));
    if ((rc = ovs_remap_apply_dev(r, d_in, 1, src_width, src_height, src_width, in_bytes, d_out, r->width, out_bytes, ctx->stream)) != OVS_OK) {
        return rc;
    }
    OVS_HIP(image_down(ctx, ctx->stream, dst, dst_stride, d_out, r->width, r->height));
    return OVS_OK;
}

} // extern "C"
