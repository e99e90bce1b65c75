/*
 * ovo_pre.cc -- CPU ORACLE (test infrastructure): the image preparation in front of the extractor.
 *
 *   util::convert_to_grayscale      src/openvslam/util/image_converter.cc:8-39
 *   util::equalize_histogram        src/openvslam/util/image_converter.cc:45-61
 *   util::stereo_rectifier::rectify src/openvslam/util/stereo_rectifier.cc:56-60
 *
 * The reference delegates the arithmetic to OpenCV (cv::cvtColor, cv::Mat::convertTo, cv::remap;
 * 3.3.1 is what it was linked against), which is not available here and for which the reference's
 * tests hold no golden pixels.  The functions below restate the published generic (non-IPP)
 * algorithms of OpenCV 3.3.1 modules/imgproc/src/color.cpp, modules/core/src/convert.cpp and
 * modules/imgproc/src/imgwarp.cpp: PARITY UNPINNED against a real OpenCV build.
 */
#include "ovo_internal.h"

#include <algorithm>
#include <climits>
#include <cstring>

namespace {

// RGB2Gray<uchar>: yuv_shift = 14, R2Y = 4899, G2Y = 9617, B2Y = 1868; the table form
// tab[r] + tab[g] + tab[b] with the rounding constant folded into one entry equals this sum.
inline uint8_t gray_of(unsigned c0, unsigned c1, unsigned c2, unsigned k0, unsigned k1, unsigned k2) {
    return static_cast<uint8_t>((c0 * k0 + c1 * k1 + c2 * k2 + (1u << 13)) >> 14);
}

// saturate_cast<uchar>(float): cvRound then clamp
inline uint8_t saturate_u8(float v) {
    const int i = ovo::cv_round(v);
    return static_cast<uint8_t>(i < 0 ? 0 : (i > 255 ? 255 : i));
}

// cvtScale_<T, uchar, float>: dst = saturate_cast<uchar>(src * scale + shift), float arithmetic
template <class T>
void convert_scale_to_u8(const T* src, size_t sstride_bytes, int w, int h, double alpha, double beta, uint8_t* dst, size_t dstride) {
    const float a = static_cast<float>(alpha), b = static_cast<float>(beta);
    for (int y = 0; y < h; ++y) {
        const T* s = reinterpret_cast<const T*>(reinterpret_cast<const uint8_t*>(src) + sstride_bytes * y);
        uint8_t* d = dst + dstride * y;
        for (int x = 0; x < w; ++x) {
            const float m = static_cast<float>(s[x]) * a;
            d[x] = saturate_u8(m + b);
        }
    }
}

template <class T>
int equalize(const T* src, int w, int h, size_t sstride_bytes, uint8_t* dst, size_t dstride) {
    // image_converter.cc:47-60: copy, sort, take the elements at (unsigned)(0.05 n) and (unsigned)(0.95 n)
    std::vector<T> vec;
    vec.reserve(static_cast<size_t>(w) * h);
    for (int y = 0; y < h; ++y) {
        const T* s = reinterpret_cast<const T*>(reinterpret_cast<const uint8_t*>(src) + sstride_bytes * y);
        vec.insert(vec.end(), s, s + w);
    }
    std::sort(vec.begin(), vec.end());
    const auto l = vec.at(static_cast<unsigned int>(0.05 * vec.size()));
    const auto hh = vec.at(static_cast<unsigned int>(0.95 * vec.size()));
    if (hh == l) {
        return ovo::fail(OVS_ERR_UNSUPPORTED, "the 5 % and 95 % levels coincide (division by zero in the reference)");
    }
    convert_scale_to_u8(src, sstride_bytes, w, h, 255.0 / (hh - l), -255.0 * l / (hh - l), dst, dstride);
    return OVS_OK;
}

// initInterTab2D(INTER_LINEAR, fixpt = true): 32 x 32 entries of 2 x 2 short weights, built in the
// order and with the sum correction of imgwarp.cpp (the correction scans the 2 x 2 window that
// starts at (ksize / 2, ksize / 2), which for ksize = 2 reaches into the following, still zero,
// entries of the static table).
struct bilinear_tab {
    short w[32 * 32 * 4 + 8];
    bilinear_tab() {
        std::memset(w, 0, sizeof(w));
        const int ksize = 2;
        float tab1d[32][2];
        for (int i = 0; i < 32; ++i) {
            const float x = i * (1.0f / 32);
            tab1d[i][0] = 1.0f - x;
            tab1d[i][1] = x;
        }
        short* itab = w;
        for (int i = 0; i < 32; ++i) {
            for (int j = 0; j < 32; ++j, itab += ksize * ksize) {
                int isum = 0;
                for (int k1 = 0; k1 < ksize; ++k1) {
                    const float vy = tab1d[i][k1];
                    for (int k2 = 0; k2 < ksize; ++k2) {
                        const float v = vy * tab1d[j][k2];
                        const int r = ovo::cv_round(v * 32768.0f);
                        const short sv = static_cast<short>(r > SHRT_MAX ? SHRT_MAX : (r < SHRT_MIN ? SHRT_MIN : r));
                        itab[k1 * ksize + k2] = sv;
                        isum += sv;
                    }
                }
                if (isum != 32768) {
                    const int diff = isum - 32768;
                    const int ksize2 = ksize / 2;
                    int Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                    for (int k1 = ksize2; k1 < ksize2 + 2; ++k1) {
                        for (int k2 = ksize2; k2 < ksize2 + 2; ++k2) {
                            if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2]) {
                                mk1 = k1, mk2 = k2;
                            }
                            else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2]) {
                                Mk1 = k1, Mk2 = k2;
                            }
                        }
                    }
                    if (diff < 0) {
                        itab[Mk1 * ksize + Mk2] = static_cast<short>(itab[Mk1 * ksize + Mk2] - diff);
                    }
                    else {
                        itab[mk1 * ksize + mk2] = static_cast<short>(itab[mk1 * ksize + mk2] - diff);
                    }
                }
            }
        }
    }
};

inline int fixed_coord(float v) {
    // cvRound(v * INTER_TAB_SIZE); cvtss2si returns INT_MIN for NaN and out-of-range values
    const float s = v * 32.0f;
    if (!(std::fabs(s) < 2147483648.0f)) {
        return INT_MIN;
    }
    return ovo::cv_round(s);
}

inline int saturate_short(int v) { return v < SHRT_MIN ? SHRT_MIN : (v > SHRT_MAX ? SHRT_MAX : v); }

} // namespace

extern "C" {

int ovo_convert_to_grayscale(const uint8_t* src, int width, int height, int channels, size_t src_stride, int color_order,
                             uint8_t* dst, size_t dst_stride) {
    if (!src || !dst || width <= 0 || height <= 0 || (channels != 3 && channels != 4)) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (color_order != OVS_COLOR_RGB && color_order != OVS_COLOR_BGR) {
        return ovo::fail(OVS_ERR_UNSUPPORTED, "color order Gray: the image stays multi-channel (image_converter.cc:11-13)");
    }
    // COLOR_BGR2GRAY: blue first; COLOR_RGB2GRAY: the coefficients of the first and third channel swap
    const unsigned k0 = color_order == OVS_COLOR_BGR ? 1868 : 4899, k1 = 9617, k2 = color_order == OVS_COLOR_BGR ? 4899 : 1868;
    for (int y = 0; y < height; ++y) {
        const uint8_t* s = src + src_stride * y;
        uint8_t* d = dst + dst_stride * y;
        for (int x = 0; x < width; ++x, s += channels) {
            d[x] = gray_of(s[0], s[1], s[2], k0, k1, k2);
        }
    }
    return OVS_OK;
}

int ovo_equalize_histogram(const void* src, int width, int height, size_t src_stride, int bits, uint8_t* dst, size_t dst_stride) {
    if (!src || !dst || width <= 0 || height <= 0) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if (bits == 8) {
        return equalize(static_cast<const uint8_t*>(src), width, height, src_stride, dst, dst_stride);
    }
    if (bits == 16) {
        return equalize(static_cast<const uint16_t*>(src), width, height, src_stride, dst, dst_stride);
    }
    return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "bits must be 8 or 16");
}

/* cv::remap(src, dst, map_x, map_y, INTER_LINEAR) with CV_32FC1 maps, BORDER_CONSTANT, value 0 */
int ovo_remap_linear(const uint8_t* src, int src_width, int src_height, size_t src_stride, const float* map_x, const float* map_y,
                     int width, int height, uint8_t* dst, size_t dst_stride) {
    if (!src || !dst || !map_x || !map_y || width <= 0 || height <= 0 || src_width <= 0 || src_height <= 0) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    static const bilinear_tab tab;
    const unsigned width1 = static_cast<unsigned>(std::max(src_width - 1, 0)), height1 = static_cast<unsigned>(std::max(src_height - 1, 0));
    for (int y = 0; y < height; ++y) {
        for (int x = 0; x < width; ++x) {
            const size_t m = static_cast<size_t>(y) * width + x;
            const int fsx = fixed_coord(map_x[m]), fsy = fixed_coord(map_y[m]);
            const int sx = saturate_short(fsx >> 5), sy = saturate_short(fsy >> 5);
            const short* w = tab.w + 4 * (((fsy & 31) << 5) | (fsx & 31));
            uint8_t& D = dst[dst_stride * y + x];
            if (static_cast<unsigned>(sx) < width1 && static_cast<unsigned>(sy) < height1) {
                const uint8_t* S = src + src_stride * sy + sx;
                D = static_cast<uint8_t>((S[0] * w[0] + S[1] * w[1] + S[src_stride] * w[2] + S[src_stride + 1] * w[3] + (1 << 14)) >> 15);
            }
            else if (sx >= src_width || sx + 1 < 0 || sy >= src_height || sy + 1 < 0) {
                D = 0;
            }
            else {
                auto tap = [&](int xx, int yy) -> int {
                    return (static_cast<unsigned>(xx) < static_cast<unsigned>(src_width) && static_cast<unsigned>(yy) < static_cast<unsigned>(src_height))
                               ? src[src_stride * yy + xx]
                               : 0;
                };
                const int v = tap(sx, sy) * w[0] + tap(sx + 1, sy) * w[1] + tap(sx, sy + 1) * w[2] + tap(sx + 1, sy + 1) * w[3];
                const int r = (v + (1 << 14)) >> 15;
                D = static_cast<uint8_t>(r < 0 ? 0 : (r > 255 ? 255 : r));
            }
        }
    }
    return OVS_OK;
}

/* the weight table itself, for the tests: out = 32 * 32 * 4 shorts */
void ovo_remap_linear_table(int16_t* out) {
    static const bilinear_tab tab;
    std::memcpy(out, tab.w, 32 * 32 * 4 * sizeof(short));
}

} // extern "C"
