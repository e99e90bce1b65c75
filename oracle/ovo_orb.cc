/*
 * ovo_orb.cc -- CPU ORACLE (test infrastructure): restatement of the reference's ORB extractor,
 * src/openvslam/feature/orb_extractor.cc (all of it), orb_extractor_node.cc, orb_params.cc and
 * util/trigonometric.h, on top of the OpenCV primitive restatements of ovo_cvprim.cc.
 * Parity status: see ovo.h.  Float arithmetic is evaluated in the order written in the reference
 * (this file is built without -ffast-math and with -ffp-contract=off).
 */
#include "ovo_internal.h"
#include "ovo_orb_pattern.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <list>
#include <stdexcept>

namespace ovo {

static thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

// ---- util/trigonometric.h:11-46 --------------------------------------------------------------
static constexpr float _PI = 3.14159265358979f;
static constexpr float _PI_2 = _PI / 2.0f;
static constexpr float _TWO_PI = 2.0f * _PI;
static constexpr float _INV_TWO_PI = 1.0f / _TWO_PI;
static constexpr float _THREE_PI_2 = 3.0f * _PI_2;

static inline float _cos(float v) {
    constexpr float c1 = 0.99940307f;
    constexpr float c2 = -0.49558072f;
    constexpr float c3 = 0.03679168f;
    const float v2 = v * v;
    return c1 + v2 * (c2 + c3 * v2);
}

float trig_cos(float v) {
    v = v - cv_floor(v * _INV_TWO_PI) * _TWO_PI;
    v = (0.0f < v) ? v : -v;
    if (v < _PI_2) {
        return _cos(v);
    }
    else if (v < _PI) {
        return -_cos(_PI - v);
    }
    else if (v < _THREE_PI_2) {
        return -_cos(v - _PI);
    }
    else {
        return _cos(_TWO_PI - v);
    }
}

float trig_sin(float v) {
    return trig_cos(_PI_2 - v);
}

// ---- feature/orb_params.cc:53-87 -------------------------------------------------------------
static std::vector<float> calc_scale_factors(unsigned int num_scale_levels, float scale_factor) {
    std::vector<float> scale_factors(num_scale_levels, 1.0);
    for (unsigned int level = 1; level < num_scale_levels; ++level) {
        scale_factors.at(level) = scale_factor * scale_factors.at(level - 1);
    }
    return scale_factors;
}

static std::vector<float> calc_inv_scale_factors(unsigned int num_scale_levels, float scale_factor) {
    std::vector<float> inv_scale_factors(num_scale_levels, 1.0);
    for (unsigned int level = 1; level < num_scale_levels; ++level) {
        inv_scale_factors.at(level) = (1.0f / scale_factor) * inv_scale_factors.at(level - 1);
    }
    return inv_scale_factors;
}

static std::vector<float> calc_level_sigma_sq(unsigned int num_scale_levels, float scale_factor) {
    float scale_factor_at_level = 1.0;
    std::vector<float> level_sigma_sq(num_scale_levels, 1.0);
    for (unsigned int level = 1; level < num_scale_levels; ++level) {
        scale_factor_at_level = scale_factor * scale_factor_at_level;
        level_sigma_sq.at(level) = scale_factor_at_level * scale_factor_at_level;
    }
    return level_sigma_sq;
}

static std::vector<float> calc_inv_level_sigma_sq(unsigned int num_scale_levels, float scale_factor) {
    float scale_factor_at_level = 1.0;
    std::vector<float> inv_level_sigma_sq(num_scale_levels, 1.0);
    for (unsigned int level = 1; level < num_scale_levels; ++level) {
        scale_factor_at_level = scale_factor * scale_factor_at_level;
        inv_level_sigma_sq.at(level) = 1.0f / (scale_factor_at_level * scale_factor_at_level);
    }
    return inv_level_sigma_sq;
}

// ---- image container ---------------------------------------------------------------------------
struct image {
    int cols = 0, rows = 0;
    size_t step = 0;
    std::vector<uint8_t> data;
    bool empty() const { return cols == 0 || rows == 0; }
    void create(int r, int c) {
        rows = r;
        cols = c;
        step = static_cast<size_t>(c);
        data.assign(static_cast<size_t>(r) * c, 0);
    }
    const uint8_t* ptr(int y) const { return data.data() + static_cast<size_t>(y) * step; }
    uint8_t* ptr(int y) { return data.data() + static_cast<size_t>(y) * step; }
    // cv::Mat::at without bounds check; clamped here so that a bad index cannot crash the oracle
    uint8_t at(int y, int x) const {
        y = std::min(std::max(y, 0), rows - 1);
        x = std::min(std::max(x, 0), cols - 1);
        return ptr(y)[x];
    }
};

// ---- feature/orb_extractor_node.{h,cc} -------------------------------------------------------
struct point2i {
    int x = 0, y = 0;
};

struct extractor_node {
    std::vector<ovs_keypoint> keypts_;
    point2i pt_begin_, pt_end_;
    std::list<extractor_node>::iterator iter_;
    bool is_leaf_node_ = false;
    // creation sequence number: replaces the node's heap address as the tie-breaker of the
    // (count, node*) sort at orb_extractor.cc:452
    int seq_ = 0;

    std::array<extractor_node, 4> divide_node() const {
        // orb_extractor_node.cc:6-50
        const unsigned int half_x = cv_ceil((pt_end_.x - pt_begin_.x) / 2.0);
        const unsigned int half_y = cv_ceil((pt_end_.y - pt_begin_.y) / 2.0);

        std::array<extractor_node, 4> child_nodes;

        const point2i pt_top{static_cast<int>(pt_begin_.x + half_x), pt_begin_.y};
        const point2i pt_left{pt_begin_.x, static_cast<int>(pt_begin_.y + half_y)};
        const point2i pt_center{static_cast<int>(pt_begin_.x + half_x), static_cast<int>(pt_begin_.y + half_y)};
        const point2i pt_right{pt_end_.x, static_cast<int>(pt_begin_.y + half_y)};
        const point2i pt_bottom{static_cast<int>(pt_begin_.x + half_x), pt_end_.y};

        child_nodes.at(0).pt_begin_ = pt_begin_;
        child_nodes.at(0).pt_end_ = pt_center;
        child_nodes.at(1).pt_begin_ = pt_top;
        child_nodes.at(1).pt_end_ = pt_right;
        child_nodes.at(2).pt_begin_ = pt_left;
        child_nodes.at(2).pt_end_ = pt_bottom;
        child_nodes.at(3).pt_begin_ = pt_center;
        child_nodes.at(3).pt_end_ = pt_end_;

        for (const auto& keypt : keypts_) {
            unsigned int idx = 0;
            if (pt_begin_.x + half_x <= keypt.x) {
                idx += 1;
            }
            if (pt_begin_.y + half_y <= keypt.y) {
                idx += 2;
            }
            child_nodes.at(idx).keypts_.push_back(keypt);
        }
        return child_nodes;
    }
};

} // namespace ovo

// ---- feature/orb_extractor.{h,cc} ------------------------------------------------------------
struct ovo_orb {
    // orb_params
    unsigned int max_num_keypts_ = 2000;
    float scale_factor_ = 1.2f;
    unsigned int num_levels_ = 8;
    unsigned int ini_fast_thr_ = 20;
    unsigned int min_fast_thr = 7;
    std::vector<std::vector<float>> mask_rects_;
    int blur_kernel = OVS_BLUR_ROUNDED; // coefficients of cv::GaussianBlur (orb_extractor.cc:133), see ovs.h

    static constexpr unsigned int fast_patch_size_ = 31;
    static constexpr int fast_half_patch_size_ = fast_patch_size_ / 2;
    static constexpr unsigned int orb_patch_radius_ = 19;

    bool mask_is_initialized_ = false;
    ovo::image rect_mask_;

    std::vector<float> scale_factors_, inv_scale_factors_, level_sigma_sq_, inv_level_sigma_sq_;
    std::vector<unsigned int> num_keypts_per_level_;
    std::vector<int> u_max_;

    std::vector<ovo::image> image_pyramid_;
    std::vector<ovo::image> blurred_pyramid_;                     // kept for inspection
    std::vector<std::vector<ovs_keypoint>> candidates_per_level_; // keypts_to_distribute
    int node_seq_ = 0;
    double stage_seconds_[6] = {0, 0, 0, 0, 0, 0};

    void initialize();
    void create_rectangle_mask(unsigned int cols, unsigned int rows);
    void compute_image_pyramid(const ovo::image& image);
    void compute_fast_keypoints(std::vector<std::vector<ovs_keypoint>>& all_keypts, const ovo::image& mask);
    std::vector<ovs_keypoint> distribute_keypoints_via_tree(const std::vector<ovs_keypoint>& keypts_to_distribute,
                                                            int min_x, int max_x, int min_y, int max_y,
                                                            unsigned int num_keypts);
    std::list<ovo::extractor_node> initialize_nodes(const std::vector<ovs_keypoint>& keypts_to_distribute,
                                                    int min_x, int max_x, int min_y, int max_y) const;
    void assign_child_nodes(const std::array<ovo::extractor_node, 4>& child_nodes, std::list<ovo::extractor_node>& nodes,
                            std::vector<std::pair<int, int>>& leaf_nodes, std::vector<ovo::extractor_node*>& node_of_seq);
    std::vector<ovs_keypoint> find_keypoints_with_max_response(std::list<ovo::extractor_node>& nodes) const;
    float ic_angle(const ovo::image& image, float px, float py) const;
    void compute_orb_descriptor(const ovs_keypoint& keypt, const ovo::image& image, uint8_t* desc) const;
    int extract(const ovo::image& image, const ovo::image& image_mask, std::vector<ovs_keypoint>& keypts, std::vector<uint8_t>& descriptors);
};

namespace {

struct stage_timer {
    double& acc;
    std::chrono::steady_clock::time_point t0;
    explicit stage_timer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~stage_timer() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

} // namespace

void ovo_orb::initialize() {
    // orb_extractor.cc:205-239
    scale_factors_ = ovo::calc_scale_factors(num_levels_, scale_factor_);
    inv_scale_factors_ = ovo::calc_inv_scale_factors(num_levels_, scale_factor_);
    level_sigma_sq_ = ovo::calc_level_sigma_sq(num_levels_, scale_factor_);
    inv_level_sigma_sq_ = ovo::calc_inv_level_sigma_sq(num_levels_, scale_factor_);

    image_pyramid_.resize(num_levels_);
    blurred_pyramid_.resize(num_levels_);
    num_keypts_per_level_.resize(num_levels_);

    double desired_num_keypts_per_scale
        = max_num_keypts_ * (1.0 - 1.0 / scale_factor_)
          / (1.0 - std::pow(1.0 / scale_factor_, static_cast<double>(num_levels_)));
    unsigned int total_num_keypts = 0;
    for (unsigned int level = 0; level < num_levels_ - 1; ++level) {
        num_keypts_per_level_.at(level) = std::round(desired_num_keypts_per_scale);
        total_num_keypts += num_keypts_per_level_.at(level);
        desired_num_keypts_per_scale *= 1.0 / scale_factor_;
    }
    num_keypts_per_level_.at(num_levels_ - 1) = std::max(static_cast<int>(max_num_keypts_) - static_cast<int>(total_num_keypts), 0);

    u_max_.resize(fast_half_patch_size_ + 1);
    const unsigned int vmax = std::floor(fast_half_patch_size_ * std::sqrt(2.0) / 2 + 1);
    const unsigned int vmin = std::ceil(fast_half_patch_size_ * std::sqrt(2.0) / 2);
    for (unsigned int v = 0; v <= vmax; ++v) {
        u_max_.at(v) = std::round(std::sqrt(fast_half_patch_size_ * fast_half_patch_size_ - v * v));
    }
    for (unsigned int v = fast_half_patch_size_, v0 = 0; vmin <= v; --v) {
        while (u_max_.at(v0) == u_max_.at(v0 + 1)) {
            ++v0;
        }
        u_max_.at(v) = v0;
        ++v0;
    }
}

void ovo_orb::create_rectangle_mask(const unsigned int cols, const unsigned int rows) {
    // orb_extractor.cc:248-261.  cv::rectangle(..., Scalar(0), -1, LINE_AA) zeroes the closed box
    // [x_min, x_max] x [y_min, y_max] (clipped to the image); the anti-aliased rim it also paints
    // is non-zero and therefore invisible to the `== 0` mask test, so it is not reproduced.
    if (rect_mask_.empty()) {
        rect_mask_.create(rows, cols);
        std::fill(rect_mask_.data.begin(), rect_mask_.data.end(), 255);
    }
    for (const auto& mask_rect : mask_rects_) {
        const unsigned int x_min = std::round(cols * mask_rect.at(0));
        const unsigned int x_max = std::round(cols * mask_rect.at(1));
        const unsigned int y_min = std::round(rows * mask_rect.at(2));
        const unsigned int y_max = std::round(rows * mask_rect.at(3));
        const int x0 = std::max(static_cast<int>(x_min), 0), x1 = std::min(static_cast<int>(x_max), static_cast<int>(cols) - 1);
        const int y0 = std::max(static_cast<int>(y_min), 0), y1 = std::min(static_cast<int>(y_max), static_cast<int>(rows) - 1);
        for (int y = y0; y <= y1; ++y) {
            for (int x = x0; x <= x1; ++x) {
                rect_mask_.ptr(y)[x] = 0;
            }
        }
    }
}

void ovo_orb::compute_image_pyramid(const ovo::image& image) {
    // orb_extractor.cc:263-272
    image_pyramid_.at(0) = image;
    for (unsigned int level = 1; level < num_levels_; ++level) {
        const double scale = scale_factors_.at(level);
        const int w = static_cast<int>(std::round(image.cols * 1.0 / scale));
        const int h = static_cast<int>(std::round(image.rows * 1.0 / scale));
        const auto& src = image_pyramid_.at(level - 1);
        auto& dst = image_pyramid_.at(level);
        dst.create(h, w);
        ovo::resize_linear_u8(src.data.data(), src.cols, src.rows, src.step, dst.data.data(), w, h, dst.step);
    }
}

void ovo_orb::compute_fast_keypoints(std::vector<std::vector<ovs_keypoint>>& all_keypts, const ovo::image& mask) {
    // orb_extractor.cc:274-396
    all_keypts.resize(num_levels_);
    candidates_per_level_.assign(num_levels_, {});

    auto is_in_mask = [&mask](const unsigned int y, const unsigned int x, const float scale_factor) {
        return mask.at(static_cast<int>(y * scale_factor), static_cast<int>(x * scale_factor)) == 0;
    };

    constexpr unsigned int overlap = 6;
    constexpr unsigned int cell_size = 64;

    for (unsigned int level = 0; level < num_levels_; ++level) {
        const float scale_factor = scale_factors_.at(level);
        const auto& img = image_pyramid_.at(level);

        constexpr unsigned int min_border_x = orb_patch_radius_;
        constexpr unsigned int min_border_y = orb_patch_radius_;
        const unsigned int max_border_x = img.cols - orb_patch_radius_;
        const unsigned int max_border_y = img.rows - orb_patch_radius_;

        const unsigned int width = max_border_x - min_border_x;
        const unsigned int height = max_border_y - min_border_y;

        const unsigned int num_cols = std::ceil(width / cell_size) + 1;
        const unsigned int num_rows = std::ceil(height / cell_size) + 1;

        std::vector<ovs_keypoint> keypts_to_distribute;
        keypts_to_distribute.reserve(max_num_keypts_ * 10);

        {
            stage_timer t(stage_seconds_[1]);
            for (unsigned int i = 0; i < num_rows; ++i) {
                const unsigned int min_y = min_border_y + i * cell_size;
                if (max_border_y - overlap <= min_y) {
                    continue;
                }
                unsigned int max_y = min_y + cell_size + overlap;
                if (max_border_y < max_y) {
                    max_y = max_border_y;
                }

                for (unsigned int j = 0; j < num_cols; ++j) {
                    const unsigned int min_x = min_border_x + j * cell_size;
                    if (max_border_x - overlap <= min_x) {
                        continue;
                    }
                    unsigned int max_x = min_x + cell_size + overlap;
                    if (max_border_x < max_x) {
                        max_x = max_border_x;
                    }

                    if (!mask.empty()) {
                        if (is_in_mask(min_y, min_x, scale_factor) || is_in_mask(max_y, min_x, scale_factor)
                            || is_in_mask(min_y, max_x, scale_factor) || is_in_mask(max_y, max_x, scale_factor)) {
                            continue;
                        }
                    }

                    std::vector<ovs_keypoint> keypts_in_cell;
                    const uint8_t* sub = img.ptr(min_y) + min_x;
                    ovo::fast_9_16(sub, max_x - min_x, max_y - min_y, img.step, ini_fast_thr_, keypts_in_cell);

                    if (keypts_in_cell.empty()) {
                        ovo::fast_9_16(sub, max_x - min_x, max_y - min_y, img.step, min_fast_thr, keypts_in_cell);
                    }

                    if (keypts_in_cell.empty()) {
                        continue;
                    }

                    for (auto& keypt : keypts_in_cell) {
                        keypt.x += j * cell_size;
                        keypt.y += i * cell_size;
                        if (!mask.empty() && is_in_mask(min_border_y + keypt.y, min_border_x + keypt.x, scale_factor)) {
                            continue;
                        }
                        keypts_to_distribute.push_back(keypt);
                    }
                }
            }
        }

        candidates_per_level_.at(level) = keypts_to_distribute;

        std::vector<ovs_keypoint>& keypts_at_level = all_keypts.at(level);
        {
            stage_timer t(stage_seconds_[2]);
            keypts_at_level = distribute_keypoints_via_tree(keypts_to_distribute,
                                                            min_border_x, max_border_x, min_border_y, max_border_y,
                                                            num_keypts_per_level_.at(level));
        }

        const unsigned int scaled_patch_size = fast_patch_size_ * scale_factors_.at(level);

        for (auto& keypt : keypts_at_level) {
            keypt.x += min_border_x;
            keypt.y += min_border_y;
            keypt.octave = level;
            keypt.size = scaled_patch_size;
        }
    }

    {
        stage_timer t(stage_seconds_[3]);
        for (unsigned int level = 0; level < num_levels_; ++level) {
            for (auto& keypt : all_keypts.at(level)) {
                keypt.angle = ic_angle(image_pyramid_.at(level), keypt.x, keypt.y);
            }
        }
    }
}

std::vector<ovs_keypoint> ovo_orb::distribute_keypoints_via_tree(const std::vector<ovs_keypoint>& keypts_to_distribute,
                                                                 const int min_x, const int max_x, const int min_y, const int max_y,
                                                                 const unsigned int num_keypts) {
    // orb_extractor.cc:398-475
    auto nodes = initialize_nodes(keypts_to_distribute, min_x, max_x, min_y, max_y);

    // (count, creation sequence) instead of (count, node*); node_of_seq maps back to the node
    std::vector<std::pair<int, int>> leaf_nodes_pool;
    std::vector<ovo::extractor_node*> node_of_seq;
    node_seq_ = 0;

    bool is_filled = false;

    while (true) {
        const unsigned int prev_size = nodes.size();

        auto iter = nodes.begin();
        leaf_nodes_pool.clear();

        while (iter != nodes.end()) {
            if (iter->is_leaf_node_) {
                iter++;
                continue;
            }
            const auto child_nodes = iter->divide_node();
            assign_child_nodes(child_nodes, nodes, leaf_nodes_pool, node_of_seq);
            iter = nodes.erase(iter);
        }

        if (num_keypts <= nodes.size() || nodes.size() == prev_size) {
            is_filled = true;
            break;
        }

        if (num_keypts < nodes.size() + leaf_nodes_pool.size()) {
            is_filled = false;
            break;
        }
    }

    while (!is_filled) {
        const unsigned int prev_size = nodes.size();

        auto prev_leaf_nodes_pool = leaf_nodes_pool;
        leaf_nodes_pool.clear();

        std::sort(prev_leaf_nodes_pool.rbegin(), prev_leaf_nodes_pool.rend());
        for (const auto& prev_leaf_node : prev_leaf_nodes_pool) {
            ovo::extractor_node* node = node_of_seq.at(prev_leaf_node.second);
            const auto child_nodes = node->divide_node();
            assign_child_nodes(child_nodes, nodes, leaf_nodes_pool, node_of_seq);
            nodes.erase(node->iter_);

            if (num_keypts <= nodes.size()) {
                is_filled = true;
                break;
            }
        }

        if (is_filled || num_keypts <= nodes.size() || nodes.size() == prev_size) {
            is_filled = true;
            break;
        }
    }

    return find_keypoints_with_max_response(nodes);
}

std::list<ovo::extractor_node> ovo_orb::initialize_nodes(const std::vector<ovs_keypoint>& keypts_to_distribute,
                                                         const int min_x, const int max_x, const int min_y, const int max_y) const {
    // orb_extractor.cc:477-550 (including `delta_x = max_x - min_y` of the portrait branch, :497)
    const auto ratio = static_cast<double>(max_x - min_x) / (max_y - min_y);
    double delta_x, delta_y;
    unsigned int num_x_grid, num_y_grid;

    if (ratio > 1) {
        num_x_grid = std::round(ratio);
        num_y_grid = 1;
        delta_x = static_cast<double>(max_x - min_x) / num_x_grid;
        delta_y = max_y - min_y;
    }
    else {
        num_x_grid = 1;
        num_y_grid = std::round(1 / ratio);
        delta_x = max_x - min_y;
        delta_y = static_cast<double>(max_y - min_y) / num_y_grid;
    }

    const unsigned int num_initial_nodes = num_x_grid * num_y_grid;

    std::list<ovo::extractor_node> nodes;
    std::vector<ovo::extractor_node*> initial_nodes;
    initial_nodes.resize(num_initial_nodes);

    for (unsigned int i = 0; i < num_initial_nodes; ++i) {
        ovo::extractor_node node;
        const unsigned int ix = i % num_x_grid;
        const unsigned int iy = i / num_x_grid;
        node.pt_begin_ = ovo::point2i{static_cast<int>(delta_x * ix), static_cast<int>(delta_y * iy)};
        node.pt_end_ = ovo::point2i{static_cast<int>(delta_x * (ix + 1)), static_cast<int>(delta_y * (iy + 1))};
        nodes.push_back(node);
        initial_nodes.at(i) = &nodes.back();
    }

    for (const auto& keypt : keypts_to_distribute) {
        const unsigned int ix = keypt.x / delta_x;
        const unsigned int iy = keypt.y / delta_y;
        const unsigned int node_idx = ix + iy * num_x_grid;
        initial_nodes.at(node_idx)->keypts_.push_back(keypt);
    }

    auto iter = nodes.begin();
    while (iter != nodes.end()) {
        if (iter->keypts_.empty()) {
            iter = nodes.erase(iter);
            continue;
        }
        iter->is_leaf_node_ = (iter->keypts_.size() == 1);
        iter++;
    }

    return nodes;
}

void ovo_orb::assign_child_nodes(const std::array<ovo::extractor_node, 4>& child_nodes, std::list<ovo::extractor_node>& nodes,
                                 std::vector<std::pair<int, int>>& leaf_nodes, std::vector<ovo::extractor_node*>& node_of_seq) {
    // orb_extractor.cc:552-566
    for (const auto& child_node : child_nodes) {
        if (child_node.keypts_.empty()) {
            continue;
        }
        nodes.push_front(child_node);
        nodes.front().seq_ = node_seq_++;
        node_of_seq.push_back(&nodes.front());
        if (child_node.keypts_.size() == 1) {
            continue;
        }
        leaf_nodes.emplace_back(std::make_pair(static_cast<int>(child_node.keypts_.size()), nodes.front().seq_));
        nodes.front().iter_ = nodes.begin();
    }
}

std::vector<ovs_keypoint> ovo_orb::find_keypoints_with_max_response(std::list<ovo::extractor_node>& nodes) const {
    // orb_extractor.cc:568-590
    std::vector<ovs_keypoint> result_keypts;
    result_keypts.reserve(nodes.size());
    for (auto& node : nodes) {
        auto& node_keypts = node.keypts_;
        auto& keypt = node_keypts.at(0);
        double max_response = keypt.response;
        for (unsigned int k = 1; k < node_keypts.size(); ++k) {
            if (node_keypts.at(k).response > max_response) {
                keypt = node_keypts.at(k);
                max_response = node_keypts.at(k).response;
            }
        }
        result_keypts.push_back(keypt);
    }
    return result_keypts;
}

float ovo_orb::ic_angle(const ovo::image& image, const float px, const float py) const {
    // orb_extractor.cc:608-631
    int m_01 = 0, m_10 = 0;
    const uint8_t* const center = image.ptr(ovo::cv_round(py)) + ovo::cv_round(px);

    for (int u = -fast_half_patch_size_; u <= fast_half_patch_size_; ++u) {
        m_10 += u * center[u];
    }

    const auto step = static_cast<int>(image.step);
    for (int v = 1; v <= fast_half_patch_size_; ++v) {
        int v_sum = 0;
        const int d = u_max_.at(v);
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * step];
            const int val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }

    return ovo::fast_atan2(static_cast<float>(m_01), static_cast<float>(m_10));
}

void ovo_orb::compute_orb_descriptor(const ovs_keypoint& keypt, const ovo::image& image, uint8_t* desc) const {
    // orb_extractor.cc:641-702, non-SSE branch (USE_SSE_ORB is OFF by default)
    const float angle = keypt.angle * M_PI / 180.0;
    const float cos_angle = ovo::trig_cos(angle);
    const float sin_angle = ovo::trig_sin(angle);

    const uint8_t* const center = image.ptr(ovo::cv_round(keypt.y)) + ovo::cv_round(keypt.x);
    const auto step = static_cast<int>(image.step);

    auto get_value = [&](unsigned int shift) -> int {
        const float px = static_cast<float>(ovo_orb_pattern[shift]);
        const float py = static_cast<float>(ovo_orb_pattern[shift + 1]);
        return center[ovo::cv_round(px * sin_angle + py * cos_angle) * step
                      + ovo::cv_round(px * cos_angle - py * sin_angle)];
    };
    auto compare = [&](unsigned int shift) -> int32_t { return get_value(shift) < get_value(shift + 2); };

    static constexpr unsigned interval = 32;
    for (unsigned int i = 0; i < 1024 / interval; ++i) {
        int32_t val = compare(i * interval);
        val |= compare(i * interval + 4) << 1;
        val |= compare(i * interval + 8) << 2;
        val |= compare(i * interval + 12) << 3;
        val |= compare(i * interval + 16) << 4;
        val |= compare(i * interval + 20) << 5;
        val |= compare(i * interval + 24) << 6;
        val |= compare(i * interval + 28) << 7;
        desc[i] = static_cast<uint8_t>(val);
    }
}

int ovo_orb::extract(const ovo::image& image, const ovo::image& image_mask, std::vector<ovs_keypoint>& keypts, std::vector<uint8_t>& descriptors) {
    // orb_extractor.cc:68-144
    if (image.empty()) {
        return 1; // outputs untouched
    }
    {
        stage_timer t(stage_seconds_[0]);
        compute_image_pyramid(image);
    }

    if (!mask_is_initialized_ && !mask_rects_.empty()) {
        create_rectangle_mask(image.cols, image.rows);
        mask_is_initialized_ = true;
    }

    std::vector<std::vector<ovs_keypoint>> all_keypts;

    if (!image_mask.empty()) {
        compute_fast_keypoints(all_keypts, image_mask);
    }
    else if (!rect_mask_.empty()) {
        compute_fast_keypoints(all_keypts, rect_mask_);
    }
    else {
        compute_fast_keypoints(all_keypts, ovo::image());
    }

    unsigned int num_keypts = 0;
    for (unsigned int level = 0; level < num_levels_; ++level) {
        num_keypts += all_keypts.at(level).size();
    }
    descriptors.assign(static_cast<size_t>(num_keypts) * 32, 0);

    keypts.clear();
    keypts.reserve(num_keypts);

    unsigned int offset = 0;
    for (unsigned int level = 0; level < num_levels_; ++level) {
        auto& keypts_at_level = all_keypts.at(level);
        const auto num_keypts_at_level = keypts_at_level.size();
        blurred_pyramid_.at(level) = ovo::image();
        if (num_keypts_at_level == 0) {
            continue;
        }

        ovo::image blurred_image;
        {
            stage_timer t(stage_seconds_[4]);
            const auto& src = image_pyramid_.at(level);
            blurred_image.create(src.rows, src.cols);
            ovo::gaussian_blur_7x7(src.data.data(), src.cols, src.rows, src.step, blurred_image.data.data(), blurred_image.step, blur_kernel);
        }
        {
            stage_timer t(stage_seconds_[5]);
            for (unsigned int i = 0; i < num_keypts_at_level; ++i) {
                compute_orb_descriptor(keypts_at_level.at(i), blurred_image, descriptors.data() + static_cast<size_t>(offset + i) * 32);
            }
        }
        blurred_pyramid_.at(level) = std::move(blurred_image);

        offset += num_keypts_at_level;

        // correct_keypoint_scale, orb_extractor.cc:598-606
        if (level != 0) {
            const float scale_at_level = scale_factors_.at(level);
            for (auto& keypt_at_level : keypts_at_level) {
                keypt_at_level.x *= scale_at_level;
                keypt_at_level.y *= scale_at_level;
            }
        }

        keypts.insert(keypts.end(), keypts_at_level.begin(), keypts_at_level.end());
    }
    return 0;
}

// ---- C API -----------------------------------------------------------------------------------
extern "C" {

const char* ovo_last_error(void) { return ovo::g_last_error.c_str(); }

float ovo_cos(float v) { return ovo::trig_cos(v); }
float ovo_sin(float v) { return ovo::trig_sin(v); }

void ovo_calc_scale_factors(uint32_t n, float s, float* out) {
    const auto v = ovo::calc_scale_factors(n, s);
    std::copy(v.begin(), v.end(), out);
}
void ovo_calc_inv_scale_factors(uint32_t n, float s, float* out) {
    const auto v = ovo::calc_inv_scale_factors(n, s);
    std::copy(v.begin(), v.end(), out);
}
void ovo_calc_level_sigma_sq(uint32_t n, float s, float* out) {
    const auto v = ovo::calc_level_sigma_sq(n, s);
    std::copy(v.begin(), v.end(), out);
}
void ovo_calc_inv_level_sigma_sq(uint32_t n, float s, float* out) {
    const auto v = ovo::calc_inv_level_sigma_sq(n, s);
    std::copy(v.begin(), v.end(), out);
}

static int validate_rects(const ovs_orb_params* p, std::vector<std::vector<float>>& rects) {
    // orb_params.cc:14-24 (the C ABI fixes four floats per rectangle, so the size check cannot fail)
    for (uint32_t i = 0; i < p->num_mask_rects; ++i) {
        std::vector<float> v(p->mask_rects + 4 * i, p->mask_rects + 4 * i + 4);
        if (v.at(0) >= v.at(1)) {
            return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "x_max must be greater than x_min");
        }
        if (v.at(2) >= v.at(3)) {
            return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "y_max must be greater than x_min");
        }
        rects.push_back(v);
    }
    return OVS_OK;
}

int ovo_orb_create(const ovs_orb_params* params, ovo_orb** out) {
    if (!params || !out) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "null argument");
    }
    if (params->num_levels == 0) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "num_levels must be positive");
    }
    std::vector<std::vector<float>> rects;
    const int rc = validate_rects(params, rects);
    if (rc != OVS_OK) {
        return rc;
    }
    auto* h = new ovo_orb();
    h->max_num_keypts_ = params->max_num_keypts;
    h->scale_factor_ = params->scale_factor;
    h->num_levels_ = params->num_levels;
    h->ini_fast_thr_ = params->ini_fast_thr;
    h->min_fast_thr = params->min_fast_thr;
    h->mask_rects_ = rects;
    h->initialize();
    *out = h;
    return OVS_OK;
}

void ovo_orb_destroy(ovo_orb* h) { delete h; }

int ovo_orb_set_max_num_keypoints(ovo_orb* h, uint32_t v) {
    h->max_num_keypts_ = v;
    h->initialize();
    return OVS_OK;
}
int ovo_orb_set_scale_factor(ovo_orb* h, float v) {
    h->scale_factor_ = v;
    h->initialize();
    return OVS_OK;
}
int ovo_orb_set_num_scale_levels(ovo_orb* h, uint32_t v) {
    if (v == 0) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "num_levels must be positive");
    }
    h->num_levels_ = v;
    h->initialize();
    return OVS_OK;
}
int ovo_orb_set_initial_fast_threshold(ovo_orb* h, uint32_t v) {
    h->ini_fast_thr_ = v;
    return OVS_OK;
}
int ovo_orb_set_blur_kernel(ovo_orb* h, int kernel_variant) {
    if (kernel_variant != OVS_BLUR_ROUNDED && kernel_variant != OVS_BLUR_NORMALISED) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "unknown blur kernel variant");
    }
    h->blur_kernel = kernel_variant;
    return OVS_OK;
}
int ovo_orb_set_minimum_fast_threshold(ovo_orb* h, uint32_t v) {
    h->min_fast_thr = v;
    return OVS_OK;
}
int ovo_orb_get_num_keypts_per_level(const ovo_orb* h, uint32_t* out) {
    std::copy(h->num_keypts_per_level_.begin(), h->num_keypts_per_level_.end(), out);
    return OVS_OK;
}
int ovo_orb_get_u_max(const ovo_orb* h, int32_t* out16) {
    std::copy(h->u_max_.begin(), h->u_max_.end(), out16);
    return OVS_OK;
}

static ovo::image wrap_image(const uint8_t* p, int w, int h, size_t stride) {
    ovo::image im;
    if (!p || w <= 0 || h <= 0) {
        return im;
    }
    im.create(h, w);
    for (int y = 0; y < h; ++y) {
        std::memcpy(im.ptr(y), p + static_cast<size_t>(y) * stride, w);
    }
    return im;
}

int ovo_orb_extract(ovo_orb* h, const uint8_t* img, int width, int height, size_t stride,
                    const uint8_t* mask, size_t mask_stride,
                    ovs_keypoint* kps, uint8_t* desc, int cap, int* n) {
    if (!h) {
        return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "null handle");
    }
    const ovo::image image = wrap_image(img, width, height, stride);
    const ovo::image image_mask = wrap_image(mask, width, height, mask_stride);
    if (!image.empty()) {
        // With a level of <= 38 columns or rows the unsigned border arithmetic of
        // orb_extractor.cc:293-297 wraps around and the reference reads outside the image; both
        // the oracle and the device library reject such inputs instead.
        const double scale = h->scale_factors_.back();
        const int w = static_cast<int>(std::round(width * 1.0 / scale));
        const int hh = static_cast<int>(std::round(height * 1.0 / scale));
        if (w <= 38 || hh <= 38) {
            return ovo::fail(OVS_ERR_INVALID_ARGUMENT, "image too small for the pyramid");
        }
    }
    std::vector<ovs_keypoint> keypts;
    std::vector<uint8_t> descriptors;
    if (h->extract(image, image_mask, keypts, descriptors) != 0) {
        return OVS_OK; // empty image: outputs untouched (orb_extractor.cc:70-72)
    }
    if (n) {
        *n = static_cast<int>(keypts.size());
    }
    if (static_cast<int>(keypts.size()) > cap) {
        return ovo::fail(OVS_ERR_CAPACITY, "keypoint buffer too small");
    }
    if (!keypts.empty()) {
        std::memcpy(kps, keypts.data(), keypts.size() * sizeof(ovs_keypoint));
        std::memcpy(desc, descriptors.data(), descriptors.size());
    }
    return OVS_OK;
}

static int level_view(const std::vector<ovo::image>& pyr, int level, const uint8_t** ptr, int* w, int* hh, size_t* stride) {
    if (level < 0 || level >= static_cast<int>(pyr.size())) {
        return ovo::fail(OVS_ERR_OUT_OF_RANGE, "level out of range");
    }
    const auto& im = pyr.at(level);
    *ptr = im.empty() ? nullptr : im.data.data();
    *w = im.cols;
    *hh = im.rows;
    *stride = im.step;
    return OVS_OK;
}

int ovo_orb_pyramid_level(const ovo_orb* h, int level, const uint8_t** ptr, int* w, int* hh, size_t* stride) {
    return level_view(h->image_pyramid_, level, ptr, w, hh, stride);
}
int ovo_orb_blurred_level(const ovo_orb* h, int level, const uint8_t** ptr, int* w, int* hh, size_t* stride) {
    return level_view(h->blurred_pyramid_, level, ptr, w, hh, stride);
}
int ovo_orb_candidates(const ovo_orb* h, int level, ovs_keypoint* out, int cap) {
    if (level < 0 || level >= static_cast<int>(h->candidates_per_level_.size())) {
        return ovo::fail(OVS_ERR_OUT_OF_RANGE, "level out of range");
    }
    const auto& v = h->candidates_per_level_.at(level);
    for (int i = 0; i < std::min<int>(cap, v.size()); ++i) {
        out[i] = v[i];
    }
    return static_cast<int>(v.size());
}
int ovo_orb_stage_seconds(const ovo_orb* h, double out[6]) {
    std::copy(h->stage_seconds_, h->stage_seconds_ + 6, out);
    return OVS_OK;
}

int ovo_create_rectangle_mask(const ovs_orb_params* params, int cols, int rows, uint8_t* out) {
    ovo_orb tmp;
    std::vector<std::vector<float>> rects;
    const int rc = validate_rects(params, rects);
    if (rc != OVS_OK) {
        return rc;
    }
    tmp.mask_rects_ = rects;
    tmp.create_rectangle_mask(cols, rows);
    std::memcpy(out, tmp.rect_mask_.data.data(), static_cast<size_t>(cols) * rows);
    return OVS_OK;
}

} // extern "C"
