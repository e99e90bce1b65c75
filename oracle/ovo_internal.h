/* ovo_internal.h -- CPU ORACLE (test infrastructure): shared helpers of the oracle sources. */
#ifndef OVO_INTERNAL_H
#define OVO_INTERNAL_H

#include "ovo.h"

#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

namespace ovo {

int fail(int code, const std::string& msg);

// cvRound: SSE cvtss2si / cvtsd2si, i.e. round-half-to-even in the default rounding mode.
inline int cv_round(float v) { return static_cast<int>(lrintf(v)); }
inline int cv_round(double v) { return static_cast<int>(lrint(v)); }
// cvFloor / cvCeil of opencv2/core/fast_math.hpp (portable form).
inline int cv_floor(float v) {
    int i = static_cast<int>(v);
    return i - (i > v);
}
inline int cv_floor(double v) {
    int i = static_cast<int>(v);
    return i - (i > v);
}
inline int cv_ceil(float v) {
    int i = static_cast<int>(v);
    return i + (i < v);
}
inline int cv_ceil(double v) {
    int i = static_cast<int>(v);
    return i + (i < v);
}

float fast_atan2(float y, float x);
void resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride, uint8_t* dst, int dw, int dh, size_t dstride);
void fast_9_16(const uint8_t* img, int cols, int rows, size_t step, int threshold, std::vector<ovs_keypoint>& keypoints);
void gaussian_blur_7x7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride, int kernel_variant);
void gaussian_kernel_7_q8(int32_t out[7]);

// util/trigonometric.h
float trig_cos(float v);
float trig_sin(float v);

// match/base.h
inline uint32_t hamming_32(const uint8_t* a, const uint8_t* b) {
    // src/openvslam/match/base.h:18-39 (SWAR popcount over 8 x uint32)
    constexpr uint32_t mask_1 = 0x55555555U;
    constexpr uint32_t mask_2 = 0x33333333U;
    constexpr uint32_t mask_3 = 0x0F0F0F0FU;
    constexpr uint32_t mask_4 = 0x01010101U;
    uint32_t dist = 0;
    for (unsigned int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        __builtin_memcpy(&pa, a + 4 * i, 4);
        __builtin_memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v -= ((v >> 1) & mask_1);
        v = (v & mask_2) + ((v >> 2) & mask_2);
        dist += (((v + (v >> 4)) & mask_3) * mask_4) >> 24;
    }
    return dist;
}

constexpr uint32_t HAMMING_DIST_THR_LOW = 50;   // match/base.h:13
constexpr uint32_t HAMMING_DIST_THR_HIGH = 100; // match/base.h:14
constexpr uint32_t MAX_HAMMING_DIST = 256;      // match/base.h:15

// match/angle_checker.h, T = int
class angle_checker {
public:
    explicit angle_checker(unsigned int histogram_length = 30, unsigned int num_bins_thr = 3);
    void append_delta_angle(float delta_angle, int match);
    std::vector<int> get_valid_matches() const;
    std::vector<int> get_invalid_matches() const;

private:
    std::vector<unsigned int> index_sort_by_size() const;
    const unsigned int histogram_length_;
    const float inv_histogram_length_;
    std::vector<std::vector<int>> angle_histogram_;
    const unsigned int num_bins_thr_;
};

// data/common.cc:127-186 on a prebuilt grid (keypt_indices_in_cells[x][y])
using grid_cells = std::vector<std::vector<std::vector<unsigned int>>>;
grid_cells assign_keypoints_to_grid(const ovs_frame_view& frm);
std::vector<unsigned int> get_keypoints_in_cell(const ovs_frame_view& frm, const grid_cells& cells,
                                                float ref_x, float ref_y, float margin,
                                                int min_level = -1, int max_level = -1);

} // namespace ovo

#endif
