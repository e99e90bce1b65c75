/*
 * ovo_cvprim.cc -- CPU ORACLE (test infrastructure): restatements of the OpenCV primitives the
 * reference's ORB path calls.  OpenCV itself (3.3.1 is what the reference was linked against,
 * example/CMakeFiles/run_kitti_slam.dir/link.txt:1) is not available in this build environment
 * and the reference's tests hold no golden pixels, so these follow the published generic
 * (non-IPP, scalar) algorithms of OpenCV 3.3.1 modules/imgproc and modules/features2d:
 * PARITY UNPINNED against a real OpenCV build.
 *
 * Call sites in the reference:
 *   cvRound   src/openvslam/feature/orb_extractor.cc:611,646,677-678; match/stereo.cc:185-187;
 *             match/angle_checker.h:79
 *   cvFloor / cvCeil  data/common.cc:134-149, data/common.h:60-61, match/stereo.cc:132-133,
 *             util/trigonometric.h:27, feature/orb_extractor_node.cc:8-9
 *   cv::resize        feature/orb_extractor.cc:270
 *   cv::FAST          feature/orb_extractor.cc:340,345
 *   cv::GaussianBlur  feature/orb_extractor.cc:133
 *   cv::fastAtan2     feature/orb_extractor.cc:630
 */
#include "ovo_internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace ovo {

// ---------------------------------------------------------------------------------------------
// cv::fastAtan2(float y, float x): degrees in [0, 360).  Scalar atan_f32 of
// modules/core/src/mathfuncs_core: 7th-order odd polynomial on min/max ratio, quadrant fix-ups.
// Evaluated in float, Horner order as written, no fused multiply-add (file is built with
// -ffp-contract=off).
// ---------------------------------------------------------------------------------------------
float fast_atan2(float y, float x) {
    const float scale = static_cast<float>(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float eps = static_cast<float>(DBL_EPSILON);

    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) {
        a = 180.f - a;
    }
    if (y < 0) {
        a = 360.f - a;
    }
    return a;
}

// ---------------------------------------------------------------------------------------------
// cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR), CV_8UC1, generic path of
// modules/imgproc/src/imgwarp.cpp (3.3.1): INTER_RESIZE_COEF_BITS = 11, horizontal pass in int32
// (HResizeLinear), vertical pass VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>.
//   inv_scale = (double)dsize / ssize;  scale = 1. / inv_scale
//   fx = (float)((dx + 0.5) * scale - 0.5);  sx = cvFloor(fx);  fx -= sx
//   sx < 0 -> fx = 0, sx = 0;   sx >= ssize-1 -> fx = 0, sx = ssize-1
//   ialpha = saturate_cast<short>((1.f - fx) * 2048), saturate_cast<short>(fx * 2048)
//   rows sy, sy + 1 are clamped to [0, ssize.height - 1] (clip()), beta is not modified
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
// When both scale factors are exactly 2, OpenCV switches INTER_LINEAR to INTER_AREA
// (2x2 box mean, (a + b + c + d + 2) >> 2); that case is restated too.
// ---------------------------------------------------------------------------------------------
static inline short saturate_short(float v) {
    int iv = cv_round(v);
    return static_cast<short>(std::min(std::max(iv, -32768), 32767));
}

void resize_linear_coeffs(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& coef) {
    const double inv_scale = static_cast<double>(dsize) / ssize;
    const double scale = 1. / inv_scale;
    ofs.resize(dsize);
    coef.resize(2 * dsize);
    for (int d = 0; d < dsize; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (s < 0) {
            f = 0;
            s = 0;
        }
        if (s >= ssize - 1) {
            f = 0;
            s = ssize - 1;
        }
        ofs[d] = s;
        coef[2 * d] = saturate_short((1.f - f) * 2048);
        coef[2 * d + 1] = saturate_short(f * 2048);
    }
}

bool resize_is_area_fast_2x(int sw, int sh, int dw, int dh) {
    const double scale_x = 1. / (static_cast<double>(dw) / sw);
    const double scale_y = 1. / (static_cast<double>(dh) / sh);
    const int iscale_x = static_cast<int>(lrint(scale_x));
    const int iscale_y = static_cast<int>(lrint(scale_y));
    const bool is_area_fast = std::fabs(scale_x - iscale_x) < DBL_EPSILON && std::fabs(scale_y - iscale_y) < DBL_EPSILON;
    return is_area_fast && iscale_x == 2 && iscale_y == 2;
}

void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride,
                      uint8_t* dst, int dw, int dh, size_t dstride) {
    if (resize_is_area_fast_2x(sw, sh, dw, dh)) {
        for (int dy = 0; dy < dh; ++dy) {
            const uint8_t* s0 = src + static_cast<size_t>(2 * dy) * sstride;
            const uint8_t* s1 = s0 + sstride;
            for (int dx = 0; dx < dw; ++dx) {
                dst[dy * dstride + dx] = static_cast<uint8_t>((s0[2 * dx] + s0[2 * dx + 1] + s1[2 * dx] + s1[2 * dx + 1] + 2) >> 2);
            }
        }
        return;
    }

    std::vector<int> xofs, yofs;
    std::vector<short> ialpha, ibeta;
    resize_linear_coeffs(sw, dw, xofs, ialpha);
    // vertical: OpenCV does not clamp fy / sy when building the table; it clamps the two source
    // rows when fetching.  For sy in [0, sh-2] both formulations coincide; reproduce the original.
    {
        const double inv_scale = static_cast<double>(dh) / sh;
        const double scale = 1. / inv_scale;
        yofs.resize(dh);
        ibeta.resize(2 * dh);
        for (int d = 0; d < dh; ++d) {
            float f = static_cast<float>((d + 0.5) * scale - 0.5);
            int s = cv_floor(f);
            f -= s;
            yofs[d] = s;
            ibeta[2 * d] = saturate_short((1.f - f) * 2048);
            ibeta[2 * d + 1] = saturate_short(f * 2048);
        }
    }

    std::vector<int> row0(dw), row1(dw);
    auto hresize = [&](const uint8_t* S, std::vector<int>& D) {
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            const int sx1 = std::min(sx + 1, sw - 1);
            D[dx] = S[sx] * ialpha[2 * dx] + S[sx1] * ialpha[2 * dx + 1];
        }
    };
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    for (int dy = 0; dy < dh; ++dy) {
        const int sy0 = clip(yofs[dy], 0, sh);
        const int sy1 = clip(yofs[dy] + 1, 0, sh);
        hresize(src + static_cast<size_t>(sy0) * sstride, row0);
        hresize(src + static_cast<size_t>(sy1) * sstride, row1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* D = dst + static_cast<size_t>(dy) * dstride;
        for (int x = 0; x < dw; ++x) {
            D[x] = static_cast<uint8_t>((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cv::FAST(img, keypoints, threshold, nonmaxSuppression = true) == FAST_t<16>
// (modules/features2d/src/fast.cpp, fast_score.cpp): 16-pixel Bresenham ring, a corner needs
// >= 9 contiguous ring pixels all darker than v - t or all brighter than v + t; the score is the
// largest threshold for which the pixel is still a corner (cornerScore<16>); non-maximum
// suppression keeps a corner iff its score is strictly greater than its 8 neighbours' scores,
// non-corner neighbours counting as 0; rows/cols within 3 px of the border are never scanned.
// Output order is raster; KeyPoint(pt, size 7, angle -1, response = score, octave 0, class -1).
// ---------------------------------------------------------------------------------------------
static const int kRing[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                 {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score_16(const uint8_t* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[N];
    for (k = 0; k < N; k++) {
        d[k] = static_cast<short>(v - ptr[pixel[k]]);
    }

    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = std::min(static_cast<int>(d[k + 1]), static_cast<int>(d[k + 2]));
        a = std::min(a, static_cast<int>(d[k + 3]));
        if (a <= a0) {
            continue;
        }
        a = std::min(a, static_cast<int>(d[k + 4]));
        a = std::min(a, static_cast<int>(d[k + 5]));
        a = std::min(a, static_cast<int>(d[k + 6]));
        a = std::min(a, static_cast<int>(d[k + 7]));
        a = std::min(a, static_cast<int>(d[k + 8]));
        a0 = std::max(a0, std::min(a, static_cast<int>(d[k])));
        a0 = std::max(a0, std::min(a, static_cast<int>(d[k + 9])));
    }

    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = std::max(static_cast<int>(d[k + 1]), static_cast<int>(d[k + 2]));
        b = std::max(b, static_cast<int>(d[k + 3]));
        b = std::max(b, static_cast<int>(d[k + 4]));
        b = std::max(b, static_cast<int>(d[k + 5]));
        if (b >= b0) {
            continue;
        }
        b = std::max(b, static_cast<int>(d[k + 6]));
        b = std::max(b, static_cast<int>(d[k + 7]));
        b = std::max(b, static_cast<int>(d[k + 8]));
        b0 = std::min(b0, std::max(b, static_cast<int>(d[k])));
        b0 = std::min(b0, std::max(b, static_cast<int>(d[k + 9])));
    }

    return -b0 - 1;
}

void fast_9_16(const uint8_t* img, int cols, int rows, size_t step, int threshold, std::vector<ovs_keypoint>& keypoints) {
    const int K = 8, N = 16 + K + 1;
    int i, j, k, pixel[25];
    for (k = 0; k < 16; ++k) {
        pixel[k] = kRing[k][0] + kRing[k][1] * static_cast<int>(step);
    }
    for (; k < 25; ++k) {
        pixel[k] = pixel[k - 16];
    }

    keypoints.clear();
    if (cols < 7 || rows < 7) {
        return;
    }

    threshold = std::min(std::max(threshold, 0), 255);

    uint8_t threshold_tab[512];
    for (i = -255; i <= 255; i++) {
        threshold_tab[i + 255] = static_cast<uint8_t>(i < -threshold ? 1 : i > threshold ? 2 : 0);
    }

    std::vector<uint8_t> score_buf(static_cast<size_t>(cols) * 3, 0);
    std::vector<int> pos_buf(static_cast<size_t>(cols + 1) * 3, 0);
    uint8_t* buf[3] = {score_buf.data(), score_buf.data() + cols, score_buf.data() + 2 * cols};
    int* cpbuf[3] = {pos_buf.data() + 1, pos_buf.data() + 1 + (cols + 1), pos_buf.data() + 1 + 2 * (cols + 1)};

    for (i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + static_cast<size_t>(i) * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        std::memset(curr, 0, cols);
        int ncorners = 0;

        if (i < rows - 3) {
            for (j = 3; j < cols - 3; j++, ptr++) {
                const int v = ptr[0];
                const uint8_t* tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) {
                    continue;
                }
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) {
                    continue;
                }
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];

                if (d & 1) {
                    const int vt = v - threshold;
                    int count = 0;
                    for (k = 0; k < N; k++) {
                        const int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = static_cast<uint8_t>(corner_score_16(ptr, pixel, threshold));
                                break;
                            }
                        }
                        else {
                            count = 0;
                        }
                    }
                }
                if (d & 2) {
                    const int vt = v + threshold;
                    int count = 0;
                    for (k = 0; k < N; k++) {
                        const int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                curr[j] = static_cast<uint8_t>(corner_score_16(ptr, pixel, threshold));
                                break;
                            }
                        }
                        else {
                            count = 0;
                        }
                    }
                }
            }
        }

        cornerpos[-1] = ncorners;

        if (i == 3) {
            continue;
        }

        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];

        for (k = 0; k < ncorners; k++) {
            j = cornerpos[k];
            const int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1]
                && score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1]
                && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                ovs_keypoint kp;
                kp.x = static_cast<float>(j);
                kp.y = static_cast<float>(i - 1);
                kp.size = 7.f;
                kp.angle = -1.f;
                kp.response = static_cast<float>(score);
                kp.octave = 0;
                kp.class_id = -1;
                keypoints.push_back(kp);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101), CV_8UC1, OpenCV 3.3.1 generic
// path (modules/imgproc/src/smooth.cpp createGaussianFilter -> filter.cpp
// createSeparableLinearFilter): the float kernel getGaussianKernel(7, 2, CV_32F) is converted to
// integers with cvRound(k * 256) (symmetric smoothing kernel on 8U data, bits = 8 per pass), the
// row pass accumulates exactly in int32, the column pass ends with FixedPtCastEx<int, uchar>(16):
// saturate((sum + (1 << 15)) >> 16).  (From OpenCV 3.4.1 on, 8U blur uses a normalised Q8.8
// kernel instead and differs by +-1 grey level on some pixels; that variant is not the target.)
// ---------------------------------------------------------------------------------------------
void gaussian_kernel_7_q8(int32_t out[7]) {
    // getGaussianKernel: n = 7, sigma = 2 > 0; cf[i] = (float)exp(-0.5/sigma^2 * x^2),
    // sum in double of the float values, then cf[i] = (float)(cf[i] * (1./sum)).
    const int n = 7;
    const double sigmaX = 2.0;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    float cf[7];
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        const double t = std::exp(scale2X * x * x);
        cf[i] = static_cast<float>(t);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) {
        cf[i] = static_cast<float>(cf[i] * sum);
        out[i] = cv_round(static_cast<double>(cf[i]) * 256.0); // Mat::convertTo(CV_32S, 1 << 8)
    }
}

// The coefficients of the 8-bit GaussianBlur from OpenCV 3.4.9 / 4.2.0 on (imgproc: "bit-exact" GaussianBlur,
// getGaussianKernelFixedPoint_ED): the normalised kernel in Q8.8 with the rounding error of every coefficient
// carried into the next (error diffusion over the first half, mirrored) and the centre taking what is left
// of 256, so that the coefficients sum to 256 exactly.  OpenCV evaluates this in its software doubles
// (softdouble exp); plain IEEE doubles give the same Q8.8 values here (no product lies near a rounding
// boundary: 17.96, 33.52, 48.34).  Restated from the published algorithm: PARITY UNPINNED.
void gaussian_kernel_7_q8_normalised(int32_t out[7]) {
    const int n = 7;
    const double sigmaX = 2.0;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double values[7], sum = 0.0;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        values[i] = std::exp(scale2X * x * x);
        sum += values[i];
    }
    sum = 1.0 / sum;
    double err = 0.0;
    int32_t acc = 0;
    for (int i = 0; i < n / 2; i++) {
        const double adj_v = values[i] * sum * 256.0 + err;
        const int32_t v0 = cv_round(adj_v);
        err = adj_v - static_cast<double>(v0);
        out[i] = v0;
        out[n - 1 - i] = v0;
        acc += v0;
    }
    out[n / 2] = 256 - 2 * acc;
}

static inline int reflect101(int p, int len) {
    // cv::borderInterpolate(p, len, BORDER_REFLECT_101)
    if (static_cast<unsigned>(p) < static_cast<unsigned>(len)) {
        return p;
    }
    if (len == 1) {
        return 0;
    }
    do {
        if (p < 0) {
            p = -p - 1 + 1;
        }
        else {
            p = len - 1 - (p - len) - 1;
        }
    } while (static_cast<unsigned>(p) >= static_cast<unsigned>(len));
    return p;
}

void gaussian_blur_7x7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride, int kernel_variant) {
    int32_t k[7];
    if (kernel_variant == OVS_BLUR_NORMALISED) {
        gaussian_kernel_7_q8_normalised(k);
    }
    else {
        gaussian_kernel_7_q8(k);
    }
    std::vector<int32_t> rows(static_cast<size_t>(w) * h);
    for (int y = 0; y < h; ++y) {
        const uint8_t* S = src + static_cast<size_t>(y) * sstride;
        int32_t* R = rows.data() + static_cast<size_t>(y) * w;
        for (int x = 0; x < w; ++x) {
            int32_t s = 0;
            for (int j = 0; j < 7; ++j) {
                s += k[j] * S[reflect101(x + j - 3, w)];
            }
            R[x] = s;
        }
    }
    for (int y = 0; y < h; ++y) {
        const int32_t* R[7];
        for (int j = 0; j < 7; ++j) {
            R[j] = rows.data() + static_cast<size_t>(reflect101(y + j - 3, h)) * w;
        }
        uint8_t* D = dst + static_cast<size_t>(y) * dstride;
        for (int x = 0; x < w; ++x) {
            int32_t s = 0;
            for (int j = 0; j < 7; ++j) {
                s += k[j] * R[j][x];
            }
            const int v = (s + (1 << 15)) >> 16;
            D[x] = static_cast<uint8_t>(std::min(std::max(v, 0), 255));
        }
    }
}

} // namespace ovo

extern "C" {

int ovo_cv_round_f(float v) { return ovo::cv_round(v); }
int ovo_cv_round_d(double v) { return ovo::cv_round(v); }
int ovo_cv_floor_f(float v) { return ovo::cv_floor(v); }
int ovo_cv_ceil_f(float v) { return ovo::cv_ceil(v); }
int ovo_cv_ceil_d(double v) { return ovo::cv_ceil(v); }
float ovo_fast_atan2(float y, float x) { return ovo::fast_atan2(y, x); }

int ovo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride,
                         uint8_t* dst, int dw, int dh, size_t dstride) {
    if (!src || !dst || sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "resize: empty image");
    }
    ovo::resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
    return OVS_OK;
}

int ovo_fast_9_16(const uint8_t* img, int w, int h, size_t stride, int threshold, ovs_keypoint* kps, int cap) {
    std::vector<ovs_keypoint> v;
    ovo::fast_9_16(img, w, h, stride, threshold, v);
    const int n = static_cast<int>(v.size());
    for (int i = 0; i < std::min(n, cap); ++i) {
        kps[i] = v[i];
    }
    return n;
}

int ovo_gaussian_blur_7x7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride) {
    if (!src || !dst || w <= 0 || h <= 0) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "blur: empty image");
    }
    ovo::gaussian_blur_7x7(src, w, h, sstride, dst, dstride, OVS_BLUR_ROUNDED);
    return OVS_OK;
}

int ovo_gaussian_blur_7x7_variant(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride, int kernel_variant) {
    if (!src || !dst || w <= 0 || h <= 0 || (kernel_variant != OVS_BLUR_ROUNDED && kernel_variant != OVS_BLUR_NORMALISED)) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "blur: empty image or unknown kernel variant");
    }
    ovo::gaussian_blur_7x7(src, w, h, sstride, dst, dstride, kernel_variant);
    return OVS_OK;
}

void ovo_gaussian_kernel_7_q8(int32_t out[7]) { ovo::gaussian_kernel_7_q8(out); }
void ovo_gaussian_kernel_7_q8_normalised(int32_t out[7]) { ovo::gaussian_kernel_7_q8_normalised(out); }

} // extern "C"
