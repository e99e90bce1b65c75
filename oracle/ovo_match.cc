/*
 * ovo_match.cc -- CPU ORACLE (test infrastructure): restatement of the reference's matchers,
 * src/openvslam/match/{base.h,angle_checker.h,projection.cc,stereo.cc,robust.cc,bow_tree.cc} and
 * the grid helpers of src/openvslam/data/common.{h,cc}, over the flattened views of include/ovs.h.
 * Pointer-valued state of the reference (frm.landmarks_[i], matched_lms_in_*[i]) is represented by
 * the index of the query / keypoint whose landmark the slot holds.
 * Parity status: see ovo.h (Hamming, angle_checker and cell indices are pinned by the reference's
 * known-answer tests; the matchers themselves have no tests in the reference).
 */
#include "ovo_internal.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <numeric>
#include <stdexcept>

namespace ovo {

// ---- match/angle_checker.h:60-134 ------------------------------------------------------------
angle_checker::angle_checker(const unsigned int histogram_length, const unsigned int num_bins_thr)
    : histogram_length_(histogram_length), inv_histogram_length_(1.0f / histogram_length),
      num_bins_thr_(num_bins_thr) {
    angle_histogram_.resize(histogram_length_);
}

void angle_checker::append_delta_angle(float delta_angle, const int match) {
    if (delta_angle < 0.0) {
        delta_angle += 360.0;
    }
    if (360.0 <= delta_angle) {
        delta_angle -= 360.0;
    }
    const auto bin = static_cast<unsigned int>(cv_round(delta_angle * inv_histogram_length_));
    angle_histogram_.at(bin).push_back(match);
}

std::vector<int> angle_checker::get_valid_matches() const {
    std::vector<int> valid_matches;
    const auto bins = index_sort_by_size();
    for (unsigned int bin = 0; bin < histogram_length_; ++bin) {
        const bool is_valid_match = std::any_of(bins.begin(), bins.begin() + num_bins_thr_,
                                                [bin](const unsigned int i) { return bin == i; });
        if (is_valid_match) {
            valid_matches.insert(valid_matches.end(), angle_histogram_.at(bin).cbegin(), angle_histogram_.at(bin).cend());
        }
    }
    return valid_matches;
}

std::vector<int> angle_checker::get_invalid_matches() const {
    std::vector<int> invalid_matches;
    const auto bins = index_sort_by_size();
    for (unsigned int bin = 0; bin < histogram_length_; ++bin) {
        const bool is_valid_match = std::any_of(bins.begin(), bins.begin() + num_bins_thr_,
                                                [bin](const unsigned int i) { return bin == i; });
        if (!is_valid_match) {
            invalid_matches.insert(invalid_matches.end(), angle_histogram_.at(bin).cbegin(), angle_histogram_.at(bin).cend());
        }
    }
    return invalid_matches;
}

std::vector<unsigned int> angle_checker::index_sort_by_size() const {
    // angle_checker.h:126-134: std::sort (libstdc++ introsort, not stable) with a size-descending
    // comparator; kept as std::sort so that ties fall exactly as in a GCC build of the reference.
    const auto& vec = angle_histogram_;
    std::vector<unsigned int> indices(vec.size());
    std::iota(indices.begin(), indices.end(), 0);
    std::sort(indices.begin(), indices.end(),
              [&vec](const unsigned int a, const unsigned int b) {
                  return (vec.at(a).size() > vec.at(b).size());
              });
    return indices;
}

// ---- data/common.h:59-64, data/common.cc:97-186 ----------------------------------------------
static inline bool get_cell_indices(const ovs_grid& g, const float x, const float y, int& cell_idx_x, int& cell_idx_y) {
    cell_idx_x = cv_floor((x - g.min_x) * g.inv_cell_width);
    cell_idx_y = cv_floor((y - g.min_y) * g.inv_cell_height);
    return (0 <= cell_idx_x && cell_idx_x < static_cast<int>(g.num_cols)
            && 0 <= cell_idx_y && cell_idx_y < static_cast<int>(g.num_rows));
}

grid_cells assign_keypoints_to_grid(const ovs_frame_view& frm) {
    grid_cells keypt_indices_in_cells;
    keypt_indices_in_cells.resize(frm.grid.num_cols);
    for (auto& keypt_indices_in_row : keypt_indices_in_cells) {
        keypt_indices_in_row.resize(frm.grid.num_rows);
    }
    for (int idx = 0; idx < frm.num_keypts; ++idx) {
        const auto& keypt = frm.undist_keypts[idx];
        int cell_idx_x, cell_idx_y;
        if (get_cell_indices(frm.grid, keypt.x, keypt.y, cell_idx_x, cell_idx_y)) {
            keypt_indices_in_cells.at(cell_idx_x).at(cell_idx_y).push_back(idx);
        }
    }
    return keypt_indices_in_cells;
}

std::vector<unsigned int> get_keypoints_in_cell(const ovs_frame_view& frm, const grid_cells& keypt_indices_in_cells,
                                                const float ref_x, const float ref_y, const float margin,
                                                const int min_level, const int max_level) {
    const ovs_grid& g = frm.grid;
    std::vector<unsigned int> indices;

    const int min_cell_idx_x = std::max(0, cv_floor((ref_x - g.min_x - margin) * g.inv_cell_width));
    if (static_cast<int>(g.num_cols) <= min_cell_idx_x) {
        return indices;
    }
    const int max_cell_idx_x = std::min(static_cast<int>(g.num_cols - 1), cv_ceil((ref_x - g.min_x + margin) * g.inv_cell_width));
    if (max_cell_idx_x < 0) {
        return indices;
    }
    const int min_cell_idx_y = std::max(0, cv_floor((ref_y - g.min_y - margin) * g.inv_cell_height));
    if (static_cast<int>(g.num_rows) <= min_cell_idx_y) {
        return indices;
    }
    const int max_cell_idx_y = std::min(static_cast<int>(g.num_rows - 1), cv_ceil((ref_y - g.min_y + margin) * g.inv_cell_height));
    if (max_cell_idx_y < 0) {
        return indices;
    }

    const bool check_level = (0 < min_level) || (0 <= max_level);

    for (int cell_idx_x = min_cell_idx_x; cell_idx_x <= max_cell_idx_x; ++cell_idx_x) {
        for (int cell_idx_y = min_cell_idx_y; cell_idx_y <= max_cell_idx_y; ++cell_idx_y) {
            const auto& keypt_indices_in_cell = keypt_indices_in_cells.at(cell_idx_x).at(cell_idx_y);
            if (keypt_indices_in_cell.empty()) {
                continue;
            }
            for (unsigned int idx : keypt_indices_in_cell) {
                const auto& undist_keypt = frm.undist_keypts[idx];
                if (check_level) {
                    if (undist_keypt.octave < min_level) {
                        continue;
                    }
                    if (0 <= max_level && max_level < undist_keypt.octave) {
                        continue;
                    }
                }
                const float dist_x = undist_keypt.x - ref_x;
                const float dist_y = undist_keypt.y - ref_y;
                if (std::abs(dist_x) < margin && std::abs(dist_y) < margin) {
                    indices.push_back(idx);
                }
            }
        }
    }
    return indices;
}

static inline float stereo_x_right_of(const ovs_frame_view& f, unsigned int idx) {
    return f.stereo_x_right ? f.stereo_x_right[idx] : -1.0f;
}

} // namespace ovo

using ovo::HAMMING_DIST_THR_HIGH;
using ovo::HAMMING_DIST_THR_LOW;
using ovo::MAX_HAMMING_DIST;

struct ovo_angle_checker {
    ovo::angle_checker impl;
    ovo_angle_checker(uint32_t a, uint32_t b) : impl(a, b) {}
};

extern "C" {

uint32_t ovo_hamming_32(const uint8_t* a, const uint8_t* b) { return ovo::hamming_32(a, b); }

uint32_t ovo_hamming_64(const uint8_t* a, const uint8_t* b) {
    // src/openvslam/match/base.h:42-63
    constexpr uint64_t mask_1 = 0x5555555555555555UL;
    constexpr uint64_t mask_2 = 0x3333333333333333UL;
    constexpr uint64_t mask_3 = 0x0F0F0F0F0F0F0F0FUL;
    constexpr uint64_t mask_4 = 0x0101010101010101UL;
    unsigned int dist = 0;
    for (unsigned int i = 0; i < 4; ++i) {
        uint64_t pa, pb;
        std::memcpy(&pa, a + 8 * i, 8);
        std::memcpy(&pb, b + 8 * i, 8);
        auto v = pa ^ pb;
        v -= (v >> 1) & mask_1;
        v = (v & mask_2) + ((v >> 2) & mask_2);
        dist += (((v + (v >> 4)) & mask_3) * mask_4) >> 56;
    }
    return dist;
}

int ovo_hamming_pairs(const uint8_t* a, const uint8_t* b, int n, uint32_t* out) {
    for (int i = 0; i < n; ++i) {
        out[i] = ovo::hamming_32(a + 32 * static_cast<size_t>(i), b + 32 * static_cast<size_t>(i));
    }
    return OVS_OK;
}

int ovo_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out) {
    for (int i = 0; i < na; ++i) {
        for (int j = 0; j < nb; ++j) {
            out[static_cast<size_t>(i) * nb + j] = static_cast<uint16_t>(ovo::hamming_32(a + 32 * static_cast<size_t>(i), b + 32 * static_cast<size_t>(j)));
        }
    }
    return OVS_OK;
}

ovo_angle_checker* ovo_angle_checker_create(uint32_t histogram_length, uint32_t num_bins_thr) {
    if (num_bins_thr > histogram_length) {
        return nullptr; // assert(num_bins_thr_ <= histogram_length_), angle_checker.h:64
    }
    return new ovo_angle_checker(histogram_length, num_bins_thr);
}
void ovo_angle_checker_destroy(ovo_angle_checker* c) { delete c; }
void ovo_angle_checker_append(ovo_angle_checker* c, float delta_angle, int match) { c->impl.append_delta_angle(delta_angle, match); }
int ovo_angle_checker_valid(const ovo_angle_checker* c, int* out, int cap) {
    const auto v = c->impl.get_valid_matches();
    for (int i = 0; i < std::min<int>(cap, v.size()); ++i) {
        out[i] = v[i];
    }
    return static_cast<int>(v.size());
}
int ovo_angle_checker_invalid(const ovo_angle_checker* c, int* out, int cap) {
    const auto v = c->impl.get_invalid_matches();
    for (int i = 0; i < std::min<int>(cap, v.size()); ++i) {
        out[i] = v[i];
    }
    return static_cast<int>(v.size());
}

int ovo_angle_checker_partition(const float* delta_angles, int n, int num_bins_thr, uint8_t* is_valid) {
    // angle_checker<int>(30, num_bins_thr) with the entry number as the match (angle_checker.h:60-134)
    if (num_bins_thr < 0 || 30 < num_bins_thr) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    ovo::angle_checker checker(30, static_cast<unsigned int>(num_bins_thr));
    try {
        for (int i = 0; i < n; ++i) {
            checker.append_delta_angle(delta_angles[i], i);
        }
    }
    catch (const std::out_of_range&) {
        return ovo::fail(OVS_ERR_OUT_OF_RANGE, "rotation bin out of range");
    }
    for (const int m : checker.get_valid_matches()) {
        is_valid[m] = 1;
    }
    for (const int m : checker.get_invalid_matches()) {
        is_valid[m] = 0;
    }
    return OVS_OK;
}

int ovo_get_cell_indices(const ovs_grid* g, float x, float y, int* cell_x, int* cell_y) {
    return ovo::get_cell_indices(*g, x, y, *cell_x, *cell_y) ? 1 : 0;
}

int ovo_get_keypoints_in_cell(const ovs_frame_view* frm, int n_queries,
                              const float* ref_xy, const float* margin,
                              const int32_t* min_level, const int32_t* max_level,
                              uint32_t* out_indices, int cap, uint32_t* out_offsets) {
    const auto cells = ovo::assign_keypoints_to_grid(*frm);
    uint32_t total = 0;
    out_offsets[0] = 0;
    for (int q = 0; q < n_queries; ++q) {
        const auto idx = ovo::get_keypoints_in_cell(*frm, cells, ref_xy[2 * q], ref_xy[2 * q + 1], margin[q],
                                                    min_level ? min_level[q] : -1, max_level ? max_level[q] : -1);
        for (const auto i : idx) {
            if (static_cast<int>(total) < cap) {
                out_indices[total] = i;
            }
            ++total;
        }
        out_offsets[q + 1] = total;
    }
    if (static_cast<int>(total) > cap) {
        return ovo::fail(OVS_ERR_CAPACITY, "index buffer too small");
    }
    return OVS_OK;
}

// ---- match/projection.cc ---------------------------------------------------------------------
// matched_query[i]: >= 0 the query whose landmark frm.landmarks_[i] holds after the call,
// -1 slot untouched, -2 slot written and then reset to nullptr by the orientation check.

int ovo_match_current_and_last_frames(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                      const uint8_t* curr_occupied, float margin, int direction,
                                      int last_num_scale_levels, float /*lowe_ratio*/, int check_orientation,
                                      int32_t* matched_query, uint32_t* num_matches_out) {
    // projection.cc:84-209
    const auto& curr_frm = *curr;
    unsigned int num_matches = 0;
    ovo::angle_checker angle_checker;
    const auto cells = ovo::assign_keypoints_to_grid(curr_frm);
    std::fill(matched_query, matched_query + curr_frm.num_keypts, -1);

    const bool assume_forward = direction == OVS_DIR_FORWARD;
    const bool assume_backward = direction == OVS_DIR_BACKWARD;

    auto has_observed_landmark = [&](unsigned int idx) {
        // curr_frm.landmarks_.at(idx) && curr_frm.landmarks_[idx]->has_observation()
        if (0 <= matched_query[idx]) {
            return q->has_observation ? q->has_observation[matched_query[idx]] != 0 : true;
        }
        return curr_occupied ? curr_occupied[idx] != 0 : false;
    };

    for (int idx_last = 0; idx_last < q->n; ++idx_last) {
        if (!q->valid[idx_last]) {
            continue; // no landmark / outlier / reprojected outside of the image
        }
        const float reproj_x = q->reproj_xy[2 * idx_last], reproj_y = q->reproj_xy[2 * idx_last + 1];
        const float x_right = q->x_right ? q->x_right[idx_last] : -1.0f;
        const int last_scale_level = q->scale_level[idx_last];
        if (last_scale_level < 0 || curr_frm.num_scale_levels <= last_scale_level) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "scale level out of range");
        }

        std::vector<unsigned int> indices;
        if (assume_forward) {
            indices = ovo::get_keypoints_in_cell(curr_frm, cells, reproj_x, reproj_y,
                                                 margin * curr_frm.scale_factors[last_scale_level],
                                                 last_scale_level, last_num_scale_levels - 1);
        }
        else if (assume_backward) {
            indices = ovo::get_keypoints_in_cell(curr_frm, cells, reproj_x, reproj_y,
                                                 margin * curr_frm.scale_factors[last_scale_level],
                                                 0, last_scale_level);
        }
        else {
            indices = ovo::get_keypoints_in_cell(curr_frm, cells, reproj_x, reproj_y,
                                                 margin * curr_frm.scale_factors[last_scale_level],
                                                 last_scale_level - 1, last_scale_level + 1);
        }
        if (indices.empty()) {
            continue;
        }

        const uint8_t* lm_desc = q->desc + 32 * static_cast<size_t>(idx_last);

        unsigned int best_hamm_dist = MAX_HAMMING_DIST;
        int best_idx = -1;

        for (const auto curr_idx : indices) {
            if (has_observed_landmark(curr_idx)) {
                continue;
            }
            if (ovo::stereo_x_right_of(curr_frm, curr_idx) > 0) {
                const float reproj_error = std::fabs(x_right - ovo::stereo_x_right_of(curr_frm, curr_idx));
                if (margin * curr_frm.scale_factors[last_scale_level] < reproj_error) {
                    continue;
                }
            }
            const auto hamm_dist = ovo::hamming_32(lm_desc, curr_frm.descriptors + 32 * static_cast<size_t>(curr_idx));
            if (hamm_dist < best_hamm_dist) {
                best_hamm_dist = hamm_dist;
                best_idx = curr_idx;
            }
        }

        if (HAMMING_DIST_THR_HIGH < best_hamm_dist) {
            continue;
        }

        matched_query[best_idx] = idx_last;
        ++num_matches;

        if (check_orientation) {
            const auto delta_angle = q->angle[idx_last] - curr_frm.undist_keypts[best_idx].angle;
            angle_checker.append_delta_angle(delta_angle, best_idx);
        }
    }

    if (check_orientation) {
        const auto invalid_matches = angle_checker.get_invalid_matches();
        for (const auto invalid_idx : invalid_matches) {
            matched_query[invalid_idx] = -2;
            --num_matches;
        }
    }

    *num_matches_out = num_matches;
    return OVS_OK;
}

int ovo_match_frame_and_landmarks(const ovs_frame_view* frm_view, const ovs_proj_queries* q,
                                  const uint8_t* frm_occupied, float margin, float lowe_ratio_,
                                  int32_t* matched_query, uint32_t* num_matches_out) {
    // projection.cc:11-82
    const auto& frm = *frm_view;
    unsigned int num_matches = 0;
    const auto cells = ovo::assign_keypoints_to_grid(frm);
    std::fill(matched_query, matched_query + frm.num_keypts, -1);

    auto has_observed_landmark = [&](unsigned int idx) {
        if (0 <= matched_query[idx]) {
            return q->has_observation ? q->has_observation[matched_query[idx]] != 0 : true;
        }
        return frm_occupied ? frm_occupied[idx] != 0 : false;
    };

    for (int i = 0; i < q->n; ++i) {
        if (!q->valid[i]) {
            continue; // !is_observable_in_tracking_ || will_be_erased()
        }
        const int pred_scale_level = q->scale_level[i];
        if (pred_scale_level < 0 || frm.num_scale_levels <= pred_scale_level) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "scale level out of range");
        }

        const auto indices_in_cell = ovo::get_keypoints_in_cell(frm, cells, q->reproj_xy[2 * i], q->reproj_xy[2 * i + 1],
                                                                margin * frm.scale_factors[pred_scale_level],
                                                                pred_scale_level - 1, pred_scale_level);
        if (indices_in_cell.empty()) {
            continue;
        }

        const uint8_t* lm_desc = q->desc + 32 * static_cast<size_t>(i);

        unsigned int best_hamm_dist = MAX_HAMMING_DIST;
        int best_scale_level = -1;
        unsigned int second_best_hamm_dist = MAX_HAMMING_DIST;
        int second_best_scale_level = -1;
        int best_idx = -1;

        for (const auto idx : indices_in_cell) {
            if (has_observed_landmark(idx)) {
                continue;
            }
            if (0 < ovo::stereo_x_right_of(frm, idx)) {
                const auto reproj_error = std::abs((q->x_right ? q->x_right[i] : -1.0f) - ovo::stereo_x_right_of(frm, idx));
                if (margin * frm.scale_factors[pred_scale_level] < reproj_error) {
                    continue;
                }
            }
            const auto dist = ovo::hamming_32(lm_desc, frm.descriptors + 32 * static_cast<size_t>(idx));
            if (dist < best_hamm_dist) {
                second_best_hamm_dist = best_hamm_dist;
                best_hamm_dist = dist;
                second_best_scale_level = best_scale_level;
                best_scale_level = frm.undist_keypts[idx].octave;
                best_idx = idx;
            }
            else if (dist < second_best_hamm_dist) {
                second_best_scale_level = frm.undist_keypts[idx].octave;
                second_best_hamm_dist = dist;
            }
        }

        if (best_hamm_dist <= HAMMING_DIST_THR_HIGH) {
            if (best_scale_level == second_best_scale_level && best_hamm_dist > lowe_ratio_ * second_best_hamm_dist) {
                continue;
            }
            matched_query[best_idx] = i;
            ++num_matches;
        }
    }

    *num_matches_out = num_matches;
    return OVS_OK;
}

int ovo_match_frame_and_keyframe(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                 const uint8_t* curr_occupied, float margin, uint32_t hamm_dist_thr,
                                 int check_orientation,
                                 int32_t* matched_query, uint32_t* num_matches_out) {
    // projection.cc:211-313
    const auto& curr_frm = *curr;
    unsigned int num_matches = 0;
    ovo::angle_checker angle_checker;
    const auto cells = ovo::assign_keypoints_to_grid(curr_frm);
    std::fill(matched_query, matched_query + curr_frm.num_keypts, -1);

    for (int idx = 0; idx < q->n; idx++) {
        if (!q->valid[idx]) {
            continue; // null / erased / already matched / not in image / out of the ORB scale range
        }
        const int pred_scale_level = q->scale_level[idx];
        if (pred_scale_level < 0 || curr_frm.num_scale_levels <= pred_scale_level) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "scale level out of range");
        }

        const auto indices = ovo::get_keypoints_in_cell(curr_frm, cells, q->reproj_xy[2 * idx], q->reproj_xy[2 * idx + 1],
                                                        margin * curr_frm.scale_factors[pred_scale_level],
                                                        pred_scale_level - 1, pred_scale_level + 1);
        if (indices.empty()) {
            continue;
        }

        const uint8_t* lm_desc = q->desc + 32 * static_cast<size_t>(idx);

        unsigned int best_hamm_dist = MAX_HAMMING_DIST;
        int best_idx = -1;

        for (unsigned long curr_idx : indices) {
            // curr_frm.landmarks_.at(curr_idx) non-null: initially, or written earlier in this call
            if ((curr_occupied && curr_occupied[curr_idx]) || 0 <= matched_query[curr_idx]) {
                continue;
            }
            const auto hamm_dist = ovo::hamming_32(lm_desc, curr_frm.descriptors + 32 * static_cast<size_t>(curr_idx));
            if (hamm_dist < best_hamm_dist) {
                best_hamm_dist = hamm_dist;
                best_idx = curr_idx;
            }
        }

        if (hamm_dist_thr < best_hamm_dist) {
            continue;
        }

        matched_query[best_idx] = idx;
        num_matches++;

        if (check_orientation) {
            const auto delta_angle = q->angle[idx] - curr_frm.undist_keypts[best_idx].angle;
            angle_checker.append_delta_angle(delta_angle, best_idx);
        }
    }

    if (check_orientation) {
        const auto invalid_matches = angle_checker.get_invalid_matches();
        for (const auto invalid_idx : invalid_matches) {
            matched_query[invalid_idx] = -2;
            --num_matches;
        }
    }

    *num_matches_out = num_matches;
    return OVS_OK;
}

int ovo_match_by_sim3_transform(const ovs_frame_view* keyfrm_view, const ovs_proj_queries* q,
                                const uint8_t* keyfrm_occupied, float margin,
                                int32_t* matched_query, uint32_t* num_matches_out) {
    // projection.cc:315-413
    const auto& keyfrm = *keyfrm_view;
    unsigned int num_matches = 0;
    const auto cells = ovo::assign_keypoints_to_grid(keyfrm);
    std::fill(matched_query, matched_query + keyfrm.num_keypts, -1);

    for (int i = 0; i < q->n; ++i) {
        if (!q->valid[i]) {
            continue;
        }
        const unsigned int pred_scale_level = static_cast<unsigned int>(q->scale_level[i]);
        if (static_cast<unsigned int>(keyfrm.num_scale_levels) <= pred_scale_level) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "scale level out of range");
        }

        const auto indices = ovo::get_keypoints_in_cell(keyfrm, cells, q->reproj_xy[2 * i], q->reproj_xy[2 * i + 1],
                                                        margin * keyfrm.scale_factors[pred_scale_level]);
        if (indices.empty()) {
            continue;
        }

        const uint8_t* lm_desc = q->desc + 32 * static_cast<size_t>(i);

        unsigned int best_dist = MAX_HAMMING_DIST;
        int best_idx = -1;

        for (const auto idx : indices) {
            if ((keyfrm_occupied && keyfrm_occupied[idx]) || 0 <= matched_query[idx]) {
                continue; // matched_lms_in_keyfrm.at(idx) non-null
            }
            const auto scale_level = static_cast<unsigned int>(keyfrm.undist_keypts[idx].octave);
            // unsigned arithmetic as written (projection.cc:390): pred == 0 rejects everything
            if (scale_level < pred_scale_level - 1 || pred_scale_level < scale_level) {
                continue;
            }
            const auto hamm_dist = ovo::hamming_32(lm_desc, keyfrm.descriptors + 32 * static_cast<size_t>(idx));
            if (hamm_dist < best_dist) {
                best_dist = hamm_dist;
                best_idx = idx;
            }
        }

        if (HAMMING_DIST_THR_LOW < best_dist) {
            continue;
        }

        matched_query[best_idx] = i;
        ++num_matches;
    }

    *num_matches_out = num_matches;
    return OVS_OK;
}

int ovo_match_keyframe_one_way(const ovs_frame_view* keyfrm_view, const ovs_proj_queries* q, float margin,
                               int32_t* best_idx_of_query) {
    // one of the two blocks of projection.cc:457-533 / :540-616
    const auto& keyfrm = *keyfrm_view;
    const auto cells = ovo::assign_keypoints_to_grid(keyfrm);
    std::fill(best_idx_of_query, best_idx_of_query + q->n, -1);

    for (int i = 0; i < q->n; ++i) {
        if (!q->valid[i]) {
            continue;
        }
        const unsigned int pred_scale_level = static_cast<unsigned int>(q->scale_level[i]);
        if (static_cast<unsigned int>(keyfrm.num_scale_levels) <= pred_scale_level) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "scale level out of range");
        }
        const auto indices = ovo::get_keypoints_in_cell(keyfrm, cells, q->reproj_xy[2 * i], q->reproj_xy[2 * i + 1],
                                                        margin * keyfrm.scale_factors[pred_scale_level]);
        if (indices.empty()) {
            continue;
        }
        const uint8_t* lm_desc = q->desc + 32 * static_cast<size_t>(i);
        unsigned int best_hamm_dist = MAX_HAMMING_DIST;
        int best_idx = -1;
        for (const auto idx : indices) {
            const auto scale_level = static_cast<unsigned int>(keyfrm.undist_keypts[idx].octave);
            if (scale_level < pred_scale_level - 1 || pred_scale_level < scale_level) {
                continue;
            }
            const auto hamm_dist = ovo::hamming_32(lm_desc, keyfrm.descriptors + 32 * static_cast<size_t>(idx));
            if (hamm_dist < best_hamm_dist) {
                best_hamm_dist = hamm_dist;
                best_idx = idx;
            }
        }
        if (best_hamm_dist <= HAMMING_DIST_THR_HIGH) {
            best_idx_of_query[i] = best_idx;
        }
    }
    return OVS_OK;
}

int ovo_cross_check(const int32_t* idx2_of_1, int n1, const int32_t* idx1_of_2, int n2,
                    int32_t* out_idx2_of_1, uint32_t* num_matches_out) {
    // projection.cc:618-633
    unsigned int num_matches = 0;
    std::fill(out_idx2_of_1, out_idx2_of_1 + n1, -1);
    for (int i = 0; i < n1; ++i) {
        const auto idx_2 = idx2_of_1[i];
        if (idx_2 < 0) {
            continue;
        }
        if (n2 <= idx_2) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "index out of range");
        }
        const auto idx_1 = idx1_of_2[idx_2];
        if (idx_1 == i) {
            out_idx2_of_1[idx_1] = idx_2;
            ++num_matches;
        }
    }
    *num_matches_out = num_matches;
    return OVS_OK;
}

// ---- match/stereo.cc -------------------------------------------------------------------------
namespace {

struct stereo_ctx {
    const ovs_image_view* left_image_pyramid_;
    const ovs_image_view* right_image_pyramid_;
    int num_levels_;
    unsigned int num_keypts_;
    const ovs_keypoint* keypts_left_;
    const ovs_keypoint* keypts_right_;
    unsigned int num_keypts_right_;
    const uint8_t* descs_left_;
    const uint8_t* descs_right_;
    const float* scale_factors_;
    const float* inv_scale_factors_;
    float focal_x_baseline_, true_baseline_;
    float min_disp_, max_disp_;
    static constexpr unsigned int hamm_dist_thr_ = (HAMMING_DIST_THR_HIGH + HAMMING_DIST_THR_LOW) / 2; // stereo.h:99

    std::vector<std::vector<unsigned int>> get_right_keypoint_indices_in_each_row(const float margin) const {
        // stereo.cc:115-142
        const unsigned int num_img_rows = left_image_pyramid_[0].height;
        std::vector<std::vector<unsigned int>> indices_right_in_row(num_img_rows, std::vector<unsigned int>());
        for (unsigned int idx_right = 0; idx_right < num_keypts_right_; ++idx_right) {
            const auto& keypt_right = keypts_right_[idx_right];
            const float y_right = keypt_right.y;
            const float r = margin * scale_factors_[keypt_right.octave];
            const int max_r = ovo::cv_ceil(y_right + r);
            const int min_r = ovo::cv_floor(y_right - r);
            for (int row_right = min_r; row_right <= max_r; ++row_right) {
                indices_right_in_row.at(row_right).push_back(idx_right);
            }
        }
        return indices_right_in_row;
    }

    void find_closest_keypoints_in_stereo(const unsigned int idx_left, const int scale_level_left,
                                          const std::vector<unsigned int>& candidate_indices_right,
                                          const float min_x_right, const float max_x_right,
                                          unsigned int& best_idx_right, unsigned int& best_hamm_dist) const {
        // stereo.cc:144-177
        best_idx_right = 0;
        best_hamm_dist = hamm_dist_thr_;
        const uint8_t* desc_left = descs_left_ + 32 * static_cast<size_t>(idx_left);
        for (const auto idx_right : candidate_indices_right) {
            const auto& keypt_right = keypts_right_[idx_right];
            if (keypt_right.octave < scale_level_left - 1 || keypt_right.octave > scale_level_left + 1) {
                continue;
            }
            const float x_right = keypt_right.x;
            if (x_right < min_x_right || max_x_right < x_right) {
                continue;
            }
            const unsigned int hamm_dist = ovo::hamming_32(desc_left, descs_right_ + 32 * static_cast<size_t>(idx_right));
            if (hamm_dist < best_hamm_dist) {
                best_idx_right = idx_right;
                best_hamm_dist = hamm_dist;
            }
        }
    }

    bool compute_subpixel_disparity(const ovs_keypoint& keypt_left, const ovs_keypoint& keypt_right,
                                    float& best_x_right, float& best_disp, float& best_correlation) const {
        // stereo.cc:179-250
        const float x_right = keypt_right.x;
        const float inv_scale_factor = inv_scale_factors_[keypt_left.octave];
        const int scaled_x_left = ovo::cv_round(keypt_left.x * inv_scale_factor);
        const int scaled_y_left = ovo::cv_round(keypt_left.y * inv_scale_factor);
        const int scaled_x_right = ovo::cv_round(x_right * inv_scale_factor);

        constexpr int win_size = 5;
        constexpr int slide_width = 5;
        const int ini_x = scaled_x_right + slide_width - win_size;
        const int end_x = scaled_x_right + slide_width + win_size + 1;
        const ovs_image_view& left = left_image_pyramid_[keypt_left.octave];
        const ovs_image_view& right = right_image_pyramid_[keypt_left.octave];
        if (ini_x < 0 || right.width <= end_x) {
            return false;
        }

        best_correlation = UINT_MAX;
        int best_offset = 0;
        std::vector<float> correlations(2 * slide_width + 1, -1);

        // rowRange / colRange would assert on an out-of-image patch
        if (scaled_y_left - win_size < 0 || left.height < scaled_y_left + win_size + 1
            || scaled_x_left - win_size < 0 || left.width < scaled_x_left + win_size + 1
            || right.height < scaled_y_left + win_size + 1) {
            throw std::out_of_range("stereo patch outside of the image");
        }

        float patch_left[11][11];
        for (int r = 0; r < 11; ++r) {
            for (int c = 0; c < 11; ++c) {
                patch_left[r][c] = static_cast<float>(left.data[static_cast<size_t>(scaled_y_left - win_size + r) * left.stride + (scaled_x_left - win_size + c)]);
            }
        }
        {
            const float center = patch_left[win_size][win_size];
            for (auto& row : patch_left) {
                for (auto& v : row) {
                    v -= center * 1.0f;
                }
            }
        }

        for (int offset = -slide_width; offset <= +slide_width; ++offset) {
            float patch_right[11][11];
            for (int r = 0; r < 11; ++r) {
                for (int c = 0; c < 11; ++c) {
                    patch_right[r][c] = static_cast<float>(right.data[static_cast<size_t>(scaled_y_left - win_size + r) * right.stride + (scaled_x_right + offset - win_size + c)]);
                }
            }
            const float center = patch_right[win_size][win_size];
            double l1 = 0.0; // cv::norm(NORM_L1) accumulates in double
            for (int r = 0; r < 11; ++r) {
                for (int c = 0; c < 11; ++c) {
                    const float pr = patch_right[r][c] - center * 1.0f;
                    l1 += std::abs(patch_left[r][c] - pr);
                }
            }
            const float correlation = l1;
            if (correlation < best_correlation) {
                best_correlation = correlation;
                best_offset = offset;
            }
            correlations.at(slide_width + offset) = correlation;
        }

        if (best_offset == -slide_width || best_offset == slide_width) {
            return false;
        }

        const float correlation_1 = correlations.at(slide_width + best_offset - 1);
        const float correlation_2 = correlations.at(slide_width + best_offset);
        const float correlation_3 = correlations.at(slide_width + best_offset + 1);

        const float x_delta = (correlation_1 - correlation_3) / (2.0 * (correlation_1 + correlation_3) - 4.0 * correlation_2);

        if (x_delta < -1.0 || 1.0 < x_delta) {
            return false;
        }

        best_x_right = scale_factors_[keypt_left.octave] * (scaled_x_right + best_offset + x_delta);
        best_disp = keypt_left.x - best_x_right;
        return true;
    }

    void compute(float* stereo_x_right, float* depths) const {
        // stereo.cc:18-113
        const auto indices_right_in_row = get_right_keypoint_indices_in_each_row(2.0);

        std::fill(stereo_x_right, stereo_x_right + num_keypts_, -1.0f);
        std::fill(depths, depths + num_keypts_, -1.0f);
        std::vector<std::pair<int, int>> correlation_and_idx_left;
        correlation_and_idx_left.reserve(num_keypts_);

        for (unsigned int idx_left = 0; idx_left < num_keypts_; ++idx_left) {
            const auto& keypt_left = keypts_left_[idx_left];
            const auto scale_level_left = keypt_left.octave;
            const float y_left = keypt_left.y;
            const float x_left = keypt_left.x;

            const auto& candidate_indices_right = indices_right_in_row.at(y_left);
            if (candidate_indices_right.empty()) {
                continue;
            }

            const float min_x_right = x_left - max_disp_;
            const float max_x_right = x_left - min_disp_;
            if (max_x_right < 0) {
                continue;
            }

            unsigned int best_idx_right = 0;
            unsigned int best_hamm_dist = hamm_dist_thr_;
            find_closest_keypoints_in_stereo(idx_left, scale_level_left, candidate_indices_right,
                                             min_x_right, max_x_right, best_idx_right, best_hamm_dist);
            if (hamm_dist_thr_ <= best_hamm_dist) {
                continue;
            }
            const auto& keypt_right = keypts_right_[best_idx_right];

            float best_x_right = -1.0f;
            float best_disp = -1.0f;
            float best_correlation = UINT_MAX;
            const auto is_valid = compute_subpixel_disparity(keypt_left, keypt_right, best_x_right, best_disp, best_correlation);
            if (!is_valid) {
                continue;
            }
            if (best_disp < min_disp_ || max_disp_ <= best_disp) {
                continue;
            }
            if (best_disp <= 0.0f) {
                best_disp = 0.01f;
                best_x_right = x_left - best_disp;
            }

            depths[idx_left] = focal_x_baseline_ / best_disp;
            stereo_x_right[idx_left] = best_x_right;
            correlation_and_idx_left.emplace_back(std::make_pair(best_correlation, idx_left));
        }

        std::sort(correlation_and_idx_left.begin(), correlation_and_idx_left.end());
        const auto median_i = correlation_and_idx_left.size() / 2;
        const float median_correlation = correlation_and_idx_left.empty()
                                             ? 0.0f
                                             : correlation_and_idx_left.at(median_i).first;
        const float correlation_thr = 2.0 * median_correlation;

        for (unsigned int i = median_i; i < correlation_and_idx_left.size(); ++i) {
            const auto correlation = correlation_and_idx_left.at(i).first;
            const auto idx_left = correlation_and_idx_left.at(i).second;
            if (correlation_thr < correlation) {
                stereo_x_right[idx_left] = -1;
                depths[idx_left] = -1;
            }
        }
    }
};

} // namespace

int ovo_stereo_compute(const ovs_image_view* left_pyr, const ovs_image_view* right_pyr, int num_levels,
                       const ovs_keypoint* keypts_left, int n_left,
                       const ovs_keypoint* keypts_right, int n_right,
                       const uint8_t* descs_left, const uint8_t* descs_right,
                       const float* scale_factors, const float* inv_scale_factors,
                       float focal_x_baseline, float true_baseline,
                       float* stereo_x_right, float* depths) {
    stereo_ctx s;
    s.left_image_pyramid_ = left_pyr;
    s.right_image_pyramid_ = right_pyr;
    s.num_levels_ = num_levels;
    s.num_keypts_ = n_left;
    s.keypts_left_ = keypts_left;
    s.keypts_right_ = keypts_right;
    s.num_keypts_right_ = n_right;
    s.descs_left_ = descs_left;
    s.descs_right_ = descs_right;
    s.scale_factors_ = scale_factors;
    s.inv_scale_factors_ = inv_scale_factors;
    s.focal_x_baseline_ = focal_x_baseline;
    s.true_baseline_ = true_baseline;
    s.min_disp_ = 0.0f;
    s.max_disp_ = focal_x_baseline / true_baseline; // stereo.cc:16
    for (int i = 0; i < n_left; ++i) {
        if (keypts_left[i].octave < 0 || num_levels <= keypts_left[i].octave) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "left keypoint octave out of range");
        }
    }
    for (int i = 0; i < n_right; ++i) {
        if (keypts_right[i].octave < 0 || num_levels <= keypts_right[i].octave) {
            return ovo::fail(OVS_ERR_OUT_OF_RANGE, "right keypoint octave out of range");
        }
    }
    try {
        s.compute(stereo_x_right, depths);
    }
    catch (const std::out_of_range& e) {
        return ovo::fail(OVS_ERR_OUT_OF_RANGE, e.what());
    }
    return OVS_OK;
}

// ---- match/robust.cc:209-315 -----------------------------------------------------------------
int ovo_brute_force_match(const uint8_t* descs_1, const float* angles_1, int n1,
                          const uint8_t* descs_2, const float* angles_2, const uint8_t* valid_2, int n2,
                          float lowe_ratio_, int check_orientation,
                          int32_t* matches, uint32_t* num_matches_out) {
    unsigned int num_matches = 0;
    ovo::angle_checker angle_checker;

    const unsigned int num_keypts_1 = n1;
    const unsigned int num_keypts_2 = n2;

    auto matched_indices_2_in_1 = std::vector<int>(num_keypts_1, -1);
    std::vector<bool> already_matched_indices_1(num_keypts_1, false);

    for (unsigned int idx_2 = 0; idx_2 < num_keypts_2; ++idx_2) {
        if (valid_2 && !valid_2[idx_2]) {
            continue; // !lm_2 || lm_2->will_be_erased()
        }
        const uint8_t* desc_2 = descs_2 + 32 * static_cast<size_t>(idx_2);

        unsigned int best_hamm_dist = MAX_HAMMING_DIST;
        int best_idx_1 = -1;
        unsigned int second_best_hamm_dist = MAX_HAMMING_DIST;

        for (unsigned int idx_1 = 0; idx_1 < num_keypts_1; ++idx_1) {
            if (already_matched_indices_1[idx_1]) {
                continue;
            }
            const auto hamm_dist = ovo::hamming_32(desc_2, descs_1 + 32 * static_cast<size_t>(idx_1));
            if (hamm_dist < best_hamm_dist) {
                second_best_hamm_dist = best_hamm_dist;
                best_hamm_dist = hamm_dist;
                best_idx_1 = idx_1;
            }
            else if (hamm_dist < second_best_hamm_dist) {
                second_best_hamm_dist = hamm_dist;
            }
        }

        if (HAMMING_DIST_THR_LOW < best_hamm_dist) {
            continue;
        }
        if (best_idx_1 < 0) {
            continue;
        }
        if (lowe_ratio_ * second_best_hamm_dist < static_cast<float>(best_hamm_dist)) {
            continue;
        }

        matched_indices_2_in_1.at(best_idx_1) = idx_2;
        already_matched_indices_1[best_idx_1] = true;

        if (check_orientation) {
            const auto delta_angle = angles_1[best_idx_1] - angles_2[idx_2];
            angle_checker.append_delta_angle(delta_angle, best_idx_1);
        }
        ++num_matches;
    }

    if (check_orientation) {
        const auto invalid_matches = angle_checker.get_invalid_matches();
        for (const auto invalid_idx_1 : invalid_matches) {
            matched_indices_2_in_1.at(invalid_idx_1) = -1;
            --num_matches;
        }
    }

    unsigned int k = 0;
    for (unsigned int idx_1 = 0; idx_1 < matched_indices_2_in_1.size(); ++idx_1) {
        const auto idx_2 = matched_indices_2_in_1.at(idx_1);
        if (idx_2 < 0) {
            continue;
        }
        matches[2 * k] = idx_1;
        matches[2 * k + 1] = idx_2;
        ++k;
    }
    *num_matches_out = num_matches;
    return OVS_OK;
}

// ---- match/bow_tree.cc -----------------------------------------------------------------------
namespace {

// lower_bound on the ascending node ids of a flattened feature vector
inline int fv_lower_bound(const ovs_bow_feat_vec* v, uint32_t id) {
    return static_cast<int>(std::lower_bound(v->node_ids, v->node_ids + v->num_nodes, id) - v->node_ids);
}

} // namespace

int ovo_bow_match_frame_and_keyframe(const ovs_bow_feat_vec* kf_vec, const uint8_t* kf_descs,
                                     const float* kf_angles, const uint8_t* valid_kf, int n_kf,
                                     const ovs_bow_feat_vec* frm_vec, const uint8_t* frm_descs,
                                     const float* frm_angles, int n_frm,
                                     float lowe_ratio_, int check_orientation,
                                     int32_t* matched_kf_idx_in_frm, uint32_t* num_matches_out) {
    // bow_tree.cc:16-122
    unsigned int num_matches = 0;
    ovo::angle_checker angle_checker;
    std::fill(matched_kf_idx_in_frm, matched_kf_idx_in_frm + n_frm, -1);

    int keyfrm_itr = 0, frm_itr = 0;
    const int keyfrm_end = kf_vec->num_nodes, frm_end = frm_vec->num_nodes;

    while (keyfrm_itr != keyfrm_end && frm_itr != frm_end) {
        if (kf_vec->node_ids[keyfrm_itr] == frm_vec->node_ids[frm_itr]) {
            for (uint32_t a = kf_vec->offsets[keyfrm_itr]; a < kf_vec->offsets[keyfrm_itr + 1]; ++a) {
                const auto keyfrm_idx = kf_vec->indices[a];
                if (keyfrm_idx >= static_cast<unsigned int>(n_kf)) {
                    return ovo::fail(OVS_ERR_OUT_OF_RANGE, "keyframe index out of range");
                }
                if (valid_kf && !valid_kf[keyfrm_idx]) {
                    continue; // !lm || lm->will_be_erased()
                }
                const uint8_t* keyfrm_desc = kf_descs + 32 * static_cast<size_t>(keyfrm_idx);

                unsigned int best_hamm_dist = MAX_HAMMING_DIST;
                int best_frm_idx = -1;
                unsigned int second_best_hamm_dist = MAX_HAMMING_DIST;

                for (uint32_t b = frm_vec->offsets[frm_itr]; b < frm_vec->offsets[frm_itr + 1]; ++b) {
                    const auto frm_idx = frm_vec->indices[b];
                    if (frm_idx >= static_cast<unsigned int>(n_frm)) {
                        return ovo::fail(OVS_ERR_OUT_OF_RANGE, "frame index out of range");
                    }
                    if (0 <= matched_kf_idx_in_frm[frm_idx]) {
                        continue;
                    }
                    const auto hamm_dist = ovo::hamming_32(keyfrm_desc, frm_descs + 32 * static_cast<size_t>(frm_idx));
                    if (hamm_dist < best_hamm_dist) {
                        second_best_hamm_dist = best_hamm_dist;
                        best_hamm_dist = hamm_dist;
                        best_frm_idx = frm_idx;
                    }
                    else if (hamm_dist < second_best_hamm_dist) {
                        second_best_hamm_dist = hamm_dist;
                    }
                }

                if (HAMMING_DIST_THR_LOW < best_hamm_dist) {
                    continue;
                }
                if (lowe_ratio_ * second_best_hamm_dist < static_cast<float>(best_hamm_dist)) {
                    continue;
                }

                matched_kf_idx_in_frm[best_frm_idx] = keyfrm_idx;

                if (check_orientation) {
                    const auto delta_angle = kf_angles[keyfrm_idx] - frm_angles[best_frm_idx];
                    angle_checker.append_delta_angle(delta_angle, best_frm_idx);
                }
                ++num_matches;
            }
            ++keyfrm_itr;
            ++frm_itr;
        }
        else if (kf_vec->node_ids[keyfrm_itr] < frm_vec->node_ids[frm_itr]) {
            keyfrm_itr = fv_lower_bound(kf_vec, frm_vec->node_ids[frm_itr]);
        }
        else {
            frm_itr = fv_lower_bound(frm_vec, kf_vec->node_ids[keyfrm_itr]);
        }
    }

    if (check_orientation) {
        const auto invalid_matches = angle_checker.get_invalid_matches();
        for (const auto invalid_idx : invalid_matches) {
            matched_kf_idx_in_frm[invalid_idx] = -1;
            --num_matches;
        }
    }
    *num_matches_out = num_matches;
    return OVS_OK;
}

int ovo_bow_match_keyframes(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1, const float* angles_1,
                            const uint8_t* valid_1, int n1,
                            const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2, const float* angles_2,
                            const uint8_t* valid_2, int n2,
                            float lowe_ratio_, int check_orientation,
                            int32_t* matched_idx2_in_kf1, uint32_t* num_matches_out) {
    // bow_tree.cc:124-246
    unsigned int num_matches = 0;
    ovo::angle_checker angle_checker;
    std::fill(matched_idx2_in_kf1, matched_idx2_in_kf1 + n1, -1);
    std::vector<bool> is_already_matched_in_keyfrm_2(n2, false);

    int itr_1 = 0, itr_2 = 0;
    const int itr_1_end = vec_1->num_nodes, itr_2_end = vec_2->num_nodes;

    while (itr_1 != itr_1_end && itr_2 != itr_2_end) {
        if (vec_1->node_ids[itr_1] == vec_2->node_ids[itr_2]) {
            for (uint32_t a = vec_1->offsets[itr_1]; a < vec_1->offsets[itr_1 + 1]; ++a) {
                const auto idx_1 = vec_1->indices[a];
                if (idx_1 >= static_cast<unsigned int>(n1)) {
                    return ovo::fail(OVS_ERR_OUT_OF_RANGE, "keyframe 1 index out of range");
                }
                if (valid_1 && !valid_1[idx_1]) {
                    continue;
                }
                const uint8_t* desc_1 = descs_1 + 32 * static_cast<size_t>(idx_1);

                unsigned int best_hamm_dist = MAX_HAMMING_DIST;
                int best_idx_2 = -1;
                unsigned int second_best_hamm_dist = MAX_HAMMING_DIST;

                for (uint32_t b = vec_2->offsets[itr_2]; b < vec_2->offsets[itr_2 + 1]; ++b) {
                    const auto idx_2 = vec_2->indices[b];
                    if (idx_2 >= static_cast<unsigned int>(n2)) {
                        return ovo::fail(OVS_ERR_OUT_OF_RANGE, "keyframe 2 index out of range");
                    }
                    if (valid_2 && !valid_2[idx_2]) {
                        continue;
                    }
                    if (is_already_matched_in_keyfrm_2.at(idx_2)) {
                        continue;
                    }
                    const auto hamm_dist = ovo::hamming_32(desc_1, descs_2 + 32 * static_cast<size_t>(idx_2));
                    if (hamm_dist < best_hamm_dist) {
                        second_best_hamm_dist = best_hamm_dist;
                        best_hamm_dist = hamm_dist;
                        best_idx_2 = idx_2;
                    }
                    else if (hamm_dist < second_best_hamm_dist) {
                        second_best_hamm_dist = hamm_dist;
                    }
                }

                if (HAMMING_DIST_THR_LOW < best_hamm_dist) {
                    continue;
                }
                if (lowe_ratio_ * second_best_hamm_dist < static_cast<float>(best_hamm_dist)) {
                    continue;
                }

                matched_idx2_in_kf1[idx_1] = best_idx_2;
                is_already_matched_in_keyfrm_2.at(best_idx_2) = true;
                num_matches++;

                if (check_orientation) {
                    const auto delta_angle = angles_1[idx_1] - angles_2[best_idx_2];
                    angle_checker.append_delta_angle(delta_angle, idx_1);
                }
            }
            ++itr_1;
            ++itr_2;
        }
        else if (vec_1->node_ids[itr_1] < vec_2->node_ids[itr_2]) {
            itr_1 = fv_lower_bound(vec_1, vec_2->node_ids[itr_2]);
        }
        else {
            itr_2 = fv_lower_bound(vec_2, vec_1->node_ids[itr_1]);
        }
    }

    if (check_orientation) {
        const auto invalid_matches = angle_checker.get_invalid_matches();
        for (const auto invalid_idx : invalid_matches) {
            matched_idx2_in_kf1[invalid_idx] = -1;
            --num_matches;
        }
    }
    *num_matches_out = num_matches;
    return OVS_OK;
}

// ---- match/robust.cc:18-172, :317-335 --------------------------------------------------------
namespace {

inline bool check_epipolar_constraint(const double* bearing_1, const double* bearing_2,
                                      const double* E_12, const float bearing_1_scale_factor) {
    // epiplane_in_1 = E_12 * bearing_2
    double e[3];
    for (int r = 0; r < 3; ++r) {
        e[r] = E_12[3 * r] * bearing_2[0] + E_12[3 * r + 1] * bearing_2[1] + E_12[3 * r + 2] * bearing_2[2];
    }
    const double dot = e[0] * bearing_1[0] + e[1] * bearing_1[1] + e[2] * bearing_1[2];
    const double norm = std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    const auto cos_residual = dot / norm;
    const auto residual_rad = M_PI / 2.0 - std::abs(std::acos(cos_residual));
    constexpr double residual_deg_thr = 0.2;
    constexpr double residual_rad_thr = residual_deg_thr * M_PI / 180.0;
    return residual_rad < residual_rad_thr * bearing_1_scale_factor;
}

} // namespace

int ovo_match_for_triangulation(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1,
                                const ovs_keypoint* undist_keypts_1, const double* bearings_1,
                                const float* stereo_x_right_1, const uint8_t* has_lm_1, int n1,
                                const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2,
                                const ovs_keypoint* undist_keypts_2, const double* bearings_2,
                                const float* stereo_x_right_2, const uint8_t* has_lm_2, int n2,
                                const double* E_12, const double* epipole_in_2,
                                const float* scale_factors_1, int num_scale_levels_1,
                                int check_orientation,
                                uint32_t* pairs, uint32_t* num_matches_out) {
    unsigned int num_matches = 0;
    ovo::angle_checker angle_checker;

    std::vector<bool> is_already_matched_in_keyfrm_2(n2, false);
    std::vector<int> matched_indices_2_in_keyfrm_1(n1, -1);

    int itr_1 = 0, itr_2 = 0;
    const int itr_1_end = vec_1->num_nodes, itr_2_end = vec_2->num_nodes;

    while (itr_1 != itr_1_end && itr_2 != itr_2_end) {
        if (vec_1->node_ids[itr_1] == vec_2->node_ids[itr_2]) {
            for (uint32_t a = vec_1->offsets[itr_1]; a < vec_1->offsets[itr_1 + 1]; ++a) {
                const auto idx_1 = vec_1->indices[a];
                if (idx_1 >= static_cast<unsigned int>(n1)) {
                    return ovo::fail(OVS_ERR_OUT_OF_RANGE, "keyframe 1 index out of range");
                }
                if (has_lm_1 && has_lm_1[idx_1]) {
                    continue;
                }
                const bool is_stereo_keypt_1 = 0 <= (stereo_x_right_1 ? stereo_x_right_1[idx_1] : -1.0f);
                const auto& keypt_1 = undist_keypts_1[idx_1];
                if (keypt_1.octave < 0 || num_scale_levels_1 <= keypt_1.octave) {
                    return ovo::fail(OVS_ERR_OUT_OF_RANGE, "octave out of range");
                }
                const double* bearing_1 = bearings_1 + 3 * static_cast<size_t>(idx_1);
                const uint8_t* desc_1 = descs_1 + 32 * static_cast<size_t>(idx_1);

                unsigned int best_hamm_dist = HAMMING_DIST_THR_LOW;
                int best_idx_2 = -1;

                for (uint32_t b = vec_2->offsets[itr_2]; b < vec_2->offsets[itr_2 + 1]; ++b) {
                    const auto idx_2 = vec_2->indices[b];
                    if (idx_2 >= static_cast<unsigned int>(n2)) {
                        return ovo::fail(OVS_ERR_OUT_OF_RANGE, "keyframe 2 index out of range");
                    }
                    if (has_lm_2 && has_lm_2[idx_2]) {
                        continue;
                    }
                    if (is_already_matched_in_keyfrm_2.at(idx_2)) {
                        continue;
                    }
                    const bool is_stereo_keypt_2 = 0 <= (stereo_x_right_2 ? stereo_x_right_2[idx_2] : -1.0f);
                    const double* bearing_2 = bearings_2 + 3 * static_cast<size_t>(idx_2);
                    const auto hamm_dist = ovo::hamming_32(desc_1, descs_2 + 32 * static_cast<size_t>(idx_2));

                    if (HAMMING_DIST_THR_LOW < hamm_dist || best_hamm_dist < hamm_dist) {
                        continue;
                    }

                    if (!is_stereo_keypt_1 && !is_stereo_keypt_2) {
                        const auto cos_dist = epipole_in_2[0] * bearing_2[0] + epipole_in_2[1] * bearing_2[1] + epipole_in_2[2] * bearing_2[2];
                        constexpr double cos_dist_thr = 0.99862953475;
                        if (cos_dist_thr < cos_dist) {
                            continue;
                        }
                    }

                    const bool is_inlier = check_epipolar_constraint(bearing_1, bearing_2, E_12, scale_factors_1[keypt_1.octave]);
                    if (is_inlier) {
                        best_idx_2 = idx_2;
                        best_hamm_dist = hamm_dist;
                    }
                }

                if (best_idx_2 < 0) {
                    continue;
                }

                is_already_matched_in_keyfrm_2.at(best_idx_2) = true;
                matched_indices_2_in_keyfrm_1.at(idx_1) = best_idx_2;
                ++num_matches;

                if (check_orientation) {
                    const auto delta_angle = keypt_1.angle - undist_keypts_2[best_idx_2].angle;
                    angle_checker.append_delta_angle(delta_angle, idx_1);
                }
            }
            ++itr_1;
            ++itr_2;
        }
        else if (vec_1->node_ids[itr_1] < vec_2->node_ids[itr_2]) {
            itr_1 = fv_lower_bound(vec_1, vec_2->node_ids[itr_2]);
        }
        else {
            itr_2 = fv_lower_bound(vec_2, vec_1->node_ids[itr_1]);
        }
    }

    if (check_orientation) {
        const auto invalid_matches = angle_checker.get_invalid_matches();
        for (const auto invalid_idx : invalid_matches) {
            matched_indices_2_in_keyfrm_1.at(invalid_idx) = -1;
            --num_matches;
        }
    }

    unsigned int k = 0;
    for (unsigned int idx_1 = 0; idx_1 < matched_indices_2_in_keyfrm_1.size(); ++idx_1) {
        if (matched_indices_2_in_keyfrm_1.at(idx_1) < 0) {
            continue;
        }
        pairs[2 * k] = idx_1;
        pairs[2 * k + 1] = matched_indices_2_in_keyfrm_1.at(idx_1);
        ++k;
    }
    *num_matches_out = num_matches;
    return OVS_OK;
}

// ---- database top-2 search -------------------------------------------------------------------
int ovo_knn2_search(const uint8_t* db, int64_t n_rows, int64_t global_row_offset,
                    const uint8_t* queries, int nq, ovs_knn2* out) {
    // the best / second-best update rule of robust.cc:260-267 over all rows, no exclusion
    for (int qi = 0; qi < nq; ++qi) {
        const uint8_t* qd = queries + 32 * static_cast<size_t>(qi);
        unsigned int best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
        int64_t best_row = -1, second_row = -1;
        for (int64_t r = 0; r < n_rows; ++r) {
            const auto d = ovo::hamming_32(qd, db + 32 * static_cast<size_t>(r));
            if (d < best) {
                second = best;
                second_row = best_row;
                best = d;
                best_row = r;
            }
            else if (d < second) {
                second = d;
                second_row = r;
            }
        }
        out[qi].dist[0] = best;
        out[qi].dist[1] = second;
        out[qi].row[0] = best_row < 0 ? -1 : best_row + global_row_offset;
        out[qi].row[1] = second_row < 0 ? -1 : second_row + global_row_offset;
    }
    return OVS_OK;
}

int ovo_knn2_merge(const ovs_knn2* lists, int n_lists, int nq, ovs_knn2* out) {
    // equivalent to the serial scan over the concatenated shards when every shard reports its
    // own best two: order candidates by (dist, row) and keep the first two
    for (int qi = 0; qi < nq; ++qi) {
        std::vector<std::pair<uint32_t, int64_t>> c;
        for (int s = 0; s < n_lists; ++s) {
            const ovs_knn2& e = lists[static_cast<size_t>(s) * nq + qi];
            for (int k = 0; k < 2; ++k) {
                if (0 <= e.row[k]) {
                    c.emplace_back(e.dist[k], e.row[k]);
                }
            }
        }
        std::sort(c.begin(), c.end());
        ovs_knn2 r;
        for (int k = 0; k < 2; ++k) {
            r.dist[k] = k < static_cast<int>(c.size()) ? c[k].first : MAX_HAMMING_DIST;
            r.row[k] = k < static_cast<int>(c.size()) ? c[k].second : -1;
        }
        out[qi] = r;
    }
    return OVS_OK;
}

// ---- batched consecutive-frame projection matching -------------------------------------------
int ovo_match_batch_current_and_last(const ovs_keypoint* kps, const uint8_t* desc,
                                     const int32_t* counts, int n_frames, int cap,
                                     const float* reproj_xy, const ovs_grid* grid,
                                     const float* scale_factors, int num_scale_levels,
                                     float margin, int check_orientation,
                                     int32_t* matched, uint32_t* num_matches) {
    if (0 < n_frames) {
        num_matches[0] = 0;
        std::fill(matched, matched + cap, -1);
    }
    for (int f = 1; f < n_frames; ++f) {
        const int n_curr = counts[f], n_last = counts[f - 1];
        if (n_curr < 0 || n_last < 0 || cap < n_curr || cap < n_last) {
            return ovo::fail(OVS_ERR_CAPACITY, "frame count exceeds capacity");
        }
        const ovs_keypoint* kp_curr = kps + static_cast<size_t>(f) * cap;
        const ovs_keypoint* kp_last = kps + static_cast<size_t>(f - 1) * cap;

        ovs_frame_view curr;
        curr.num_keypts = n_curr;
        curr.undist_keypts = kp_curr;
        curr.descriptors = desc + static_cast<size_t>(f) * cap * 32;
        curr.stereo_x_right = nullptr;
        curr.scale_factors = scale_factors;
        curr.num_scale_levels = num_scale_levels;
        curr.grid = *grid;

        std::vector<int32_t> scale_level(n_last);
        std::vector<float> angle(n_last);
        std::vector<uint8_t> valid(n_last);
        const float* rp = reproj_xy + static_cast<size_t>(f) * cap * 2;
        for (int i = 0; i < n_last; ++i) {
            scale_level[i] = kp_last[i].octave;
            angle[i] = kp_last[i].angle;
            valid[i] = !(rp[2 * i] < -1e30f);
        }
        ovs_proj_queries q;
        q.n = n_last;
        q.reproj_xy = rp;
        q.x_right = nullptr;
        q.scale_level = scale_level.data();
        q.angle = angle.data();
        q.desc = desc + static_cast<size_t>(f - 1) * cap * 32;
        q.valid = valid.data();
        q.has_observation = nullptr;

        int32_t* m = matched + static_cast<size_t>(f) * cap;
        std::fill(m, m + cap, -1);
        const int rc = ovo_match_current_and_last_frames(&curr, &q, nullptr, margin, OVS_DIR_NONE, num_scale_levels,
                                                         0.0f, check_orientation, m, &num_matches[f]);
        if (rc != OVS_OK) {
            return rc;
        }
    }
    return OVS_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// match::area::match_in_consistent_area (src/openvslam/match/area.cc:8-112): the matcher of the
// monocular initializer.  frm_1 is given by its undistorted keypoints and descriptors, frm_2 as a
// frame view; prev_matched_pts (n1 x 2, in/out) as in the reference.
// ---------------------------------------------------------------------------------------------
extern "C" int ovo_match_in_consistent_area(const ovs_keypoint* undist_keypts_1, const uint8_t* descs_1, int n1,
                                            const ovs_frame_view* frm_2, float* prev_matched_pts, int margin,
                                            float lowe_ratio_, int check_orientation,
                                            int32_t* matched_indices_2_in_frm_1, uint32_t* num_matches_out) {
    unsigned int num_matches = 0;
    ovo::angle_checker angle_checker;
    const auto cells = ovo::assign_keypoints_to_grid(*frm_2);
    const unsigned int n2 = frm_2->num_keypts;
    for (int i = 0; i < n1; ++i) {
        matched_indices_2_in_frm_1[i] = -1;
    }
    std::vector<unsigned int> matched_dists_in_frm_2(n2, MAX_HAMMING_DIST);
    std::vector<int> matched_indices_1_in_frm_2(n2, -1);

    for (int idx_1 = 0; idx_1 < n1; ++idx_1) {
        const auto& undist_keypt_1 = undist_keypts_1[idx_1];
        const auto scale_level_1 = undist_keypt_1.octave;
        if (0 < scale_level_1) {
            continue;
        }
        const auto indices = ovo::get_keypoints_in_cell(*frm_2, cells, prev_matched_pts[2 * idx_1], prev_matched_pts[2 * idx_1 + 1],
                                                        static_cast<float>(margin), scale_level_1, scale_level_1);
        if (indices.empty()) {
            continue;
        }
        const uint8_t* desc_1 = descs_1 + 32 * static_cast<size_t>(idx_1);
        unsigned int best_hamm_dist = MAX_HAMMING_DIST;
        unsigned int second_best_hamm_dist = MAX_HAMMING_DIST;
        int best_idx_2 = -1;
        for (const auto idx_2 : indices) {
            const auto hamm_dist = ovo::hamming_32(desc_1, frm_2->descriptors + 32 * static_cast<size_t>(idx_2));
            if (matched_dists_in_frm_2.at(idx_2) <= hamm_dist) {
                continue;
            }
            if (hamm_dist < best_hamm_dist) {
                second_best_hamm_dist = best_hamm_dist;
                best_hamm_dist = hamm_dist;
                best_idx_2 = idx_2;
            }
            else if (hamm_dist < second_best_hamm_dist) {
                second_best_hamm_dist = hamm_dist;
            }
        }
        if (HAMMING_DIST_THR_LOW < best_hamm_dist) {
            continue;
        }
        if (second_best_hamm_dist * lowe_ratio_ < static_cast<float>(best_hamm_dist)) {
            continue;
        }
        const auto prev_idx_1 = matched_indices_1_in_frm_2.at(best_idx_2);
        if (0 <= prev_idx_1) {
            matched_indices_2_in_frm_1[prev_idx_1] = -1;
            --num_matches;
        }
        matched_indices_2_in_frm_1[idx_1] = best_idx_2;
        matched_indices_1_in_frm_2.at(best_idx_2) = idx_1;
        matched_dists_in_frm_2.at(best_idx_2) = best_hamm_dist;
        ++num_matches;
        if (check_orientation) {
            const auto delta_angle = undist_keypts_1[idx_1].angle - frm_2->undist_keypts[best_idx_2].angle;
            angle_checker.append_delta_angle(delta_angle, idx_1);
        }
    }
    if (check_orientation) {
        const auto invalid_matches = angle_checker.get_invalid_matches();
        for (const auto invalid_idx_1 : invalid_matches) {
            if (0 <= matched_indices_2_in_frm_1[invalid_idx_1]) {
                matched_indices_2_in_frm_1[invalid_idx_1] = -1;
                --num_matches;
            }
        }
    }
    for (int idx_1 = 0; idx_1 < n1; ++idx_1) {
        if (0 <= matched_indices_2_in_frm_1[idx_1]) {
            const auto& pt = frm_2->undist_keypts[matched_indices_2_in_frm_1[idx_1]];
            prev_matched_pts[2 * idx_1] = pt.x;
            prev_matched_pts[2 * idx_1 + 1] = pt.y;
        }
    }
    *num_matches_out = num_matches;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// data::landmark::compute_descriptor (src/openvslam/data/landmark.cc:127-184) for a batch of
// landmarks: landmark l observes descriptors descs[offsets[l] .. offsets[l+1]); best_idx[l] is the
// index (inside its own range) of the descriptor whose median distance to the others is smallest,
// first one on ties; -1 for a landmark without observations (the reference returns early).
// ---------------------------------------------------------------------------------------------
extern "C" int ovo_landmark_compute_descriptors(const uint8_t* descs, const uint32_t* offsets, int n_landmarks, int32_t* best_idx_out) {
    for (int l = 0; l < n_landmarks; ++l) {
        const unsigned int num_descs = offsets[l + 1] - offsets[l];
        if (num_descs == 0) {
            best_idx_out[l] = -1;
            continue;
        }
        const uint8_t* base = descs + 32 * static_cast<size_t>(offsets[l]);
        std::vector<std::vector<unsigned int>> hamm_dists(num_descs, std::vector<unsigned int>(num_descs));
        for (unsigned int i = 0; i < num_descs; ++i) {
            hamm_dists.at(i).at(i) = 0;
            for (unsigned int j = i + 1; j < num_descs; ++j) {
                const auto dist = ovo::hamming_32(base + 32 * static_cast<size_t>(i), base + 32 * static_cast<size_t>(j));
                hamm_dists.at(i).at(j) = dist;
                hamm_dists.at(j).at(i) = dist;
            }
        }
        unsigned int best_median_dist = MAX_HAMMING_DIST;
        unsigned int best_idx = 0;
        for (unsigned idx = 0; idx < num_descs; ++idx) {
            std::vector<unsigned int> partial_hamm_dists(hamm_dists.at(idx).begin(), hamm_dists.at(idx).begin() + num_descs);
            std::sort(partial_hamm_dists.begin(), partial_hamm_dists.end());
            const auto median_dist = partial_hamm_dists.at(static_cast<unsigned int>(0.5 * (num_descs - 1)));
            if (median_dist < best_median_dist) {
                best_median_dist = median_dist;
                best_idx = idx;
            }
        }
        best_idx_out[l] = static_cast<int32_t>(best_idx);
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// The candidate search of fuse::detect_duplication (src/openvslam/match/fuse.cc:67-102) and
// fuse::replace_duplication (fuse.cc:173-236).  The visibility gates before it and the landmark
// bookkeeping after it are the caller's; `valid` marks the landmarks that reach the search.
// ---------------------------------------------------------------------------------------------
extern "C" int ovo_fuse_best_candidates(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, const double* reproj_xy_d,
                                        const float* inv_level_sigma_sq, float margin, int replace_mode, int32_t* best_idx_of_query) {
    const auto cells = ovo::assign_keypoints_to_grid(*keyfrm);
    for (int i = 0; i < q->n; ++i) {
        best_idx_of_query[i] = -1;
        if (!q->valid[i]) {
            continue;
        }
        if (q->scale_level[i] < 0 || keyfrm->num_scale_levels <= q->scale_level[i]) {
            return ovo::fail(-5, "scale level out of range"); // scale_factors_.at(pred_scale_level)
        }
        const uint8_t* lm_desc = q->desc + 32 * static_cast<size_t>(i);
        unsigned int best_dist = MAX_HAMMING_DIST;
        int best_idx = -1;
        if (!replace_mode) {
            const int pred_scale_level = q->scale_level[i];
            const auto indices = ovo::get_keypoints_in_cell(*keyfrm, cells, q->reproj_xy[2 * i], q->reproj_xy[2 * i + 1],
                                                            margin * keyfrm->scale_factors[pred_scale_level]);
            for (const auto idx : indices) {
                const auto scale_level = keyfrm->undist_keypts[idx].octave; // keypts_.at(idx).octave, supplied through the view
                if (scale_level < pred_scale_level - 1 || pred_scale_level < scale_level) {
                    continue;
                }
                const auto hamm_dist = ovo::hamming_32(lm_desc, keyfrm->descriptors + 32 * static_cast<size_t>(idx));
                if (hamm_dist < best_dist) {
                    best_dist = hamm_dist;
                    best_idx = idx;
                }
            }
        }
        else {
            const unsigned int pred_scale_level = static_cast<unsigned int>(q->scale_level[i]);
            const double reproj[2] = {reproj_xy_d[2 * i], reproj_xy_d[2 * i + 1]};
            const float x_right = q->x_right[i];
            // get_keypoints_in_cell takes floats: the double reprojection narrows at the call (fuse.cc:176)
            const auto indices = ovo::get_keypoints_in_cell(*keyfrm, cells, static_cast<float>(reproj[0]), static_cast<float>(reproj[1]),
                                                            margin * keyfrm->scale_factors[pred_scale_level]);
            for (const auto idx : indices) {
                const auto& keypt = keyfrm->undist_keypts[idx];
                const auto scale_level = static_cast<unsigned int>(keypt.octave);
                if (scale_level < pred_scale_level - 1 || pred_scale_level < scale_level) {
                    continue;
                }
                const float sxr = keyfrm->stereo_x_right ? keyfrm->stereo_x_right[idx] : -1.0f;
                if (sxr >= 0) {
                    const auto e_x = reproj[0] - keypt.x;
                    const auto e_y = reproj[1] - keypt.y;
                    const auto e_x_right = x_right - sxr;
                    const auto reproj_error_sq = e_x * e_x + e_y * e_y + e_x_right * e_x_right;
                    constexpr float chi_sq_3D = 7.81473;
                    if (chi_sq_3D < reproj_error_sq * inv_level_sigma_sq[scale_level]) {
                        continue;
                    }
                }
                else {
                    const auto e_x = reproj[0] - keypt.x;
                    const auto e_y = reproj[1] - keypt.y;
                    const auto reproj_error_sq = e_x * e_x + e_y * e_y;
                    constexpr float chi_sq_2D = 5.99146;
                    if (chi_sq_2D < reproj_error_sq * inv_level_sigma_sq[scale_level]) {
                        continue;
                    }
                }
                const auto hamm_dist = ovo::hamming_32(lm_desc, keyfrm->descriptors + 32 * static_cast<size_t>(idx));
                if (hamm_dist < best_dist) {
                    best_dist = hamm_dist;
                    best_idx = idx;
                }
            }
        }
        if (HAMMING_DIST_THR_LOW < best_dist) {
            continue;
        }
        best_idx_of_query[i] = best_idx;
    }
    return 0;
}
