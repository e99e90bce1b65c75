// ovs_shim.hpp -- header-only C++ layer with the class names and member signatures of
// feature::orb_extractor (src/openvslam/feature/orb_extractor.h:18-78) and
// match::{projection, robust, bow_tree, area, fuse, stereo} (src/openvslam/match/*.h), plus the transform of
// data::bow_vocabulary (src/openvslam/data/bow_vocabulary.h:16) that frame::compute_bow calls, on top of the C ABI of
// include/ovs.h.  The classes are templates over the caller's frame / keyframe / landmark /
// matrix types and only use the member names the reference's own implementation uses
// (landmarks_, undist_keypts_, descriptors_, stereo_x_right_, scale_factors_, camera_, ...), so in
// the reference tree they are instantiated with data::frame, data::keyframe, data::landmark and
// cv::Mat; this file itself includes neither OpenCV nor Eigen.  namespace util mirrors
// util/image_converter.h and util/stereo_rectifier.h (the image preparation in front of extract()).
//
// What stays on the host, exactly where the reference has it: camera reprojection (double
// precision), landmark getters (they take mutexes), the final pointer writes.  What moves to the
// GPU: the window queries, every Hamming distance, best / second-best selection, the sequential
// conflict resolution and the rotation histogram.
//
// Errors: the C ABI returns status codes; the shim throws std::runtime_error with
// ovs_last_error(), which is what the reference's own failure modes (std::runtime_error from
// orb_params, std::out_of_range from .at()) look like to a caller.
#ifndef OVS_SHIM_HPP
#define OVS_SHIM_HPP

#include "ovs.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace ovs_shim {

inline void check(int rc) {
    if (rc != OVS_OK) {
        throw std::runtime_error(std::string("ovs: ") + ovs_last_error());
    }
}

// Adapter between the caller's 8-bit matrix type and raw memory.  Specialise for cv::Mat in the
// reference tree (empty / data / cols / rows / step / create_u8 / release over the cv::Mat members
// of the same meaning, about ten lines); the primary template fits any type with
// data / cols / rows / step members and create(rows, cols).
template <class Mat>
struct mat_traits {
    static bool empty(const Mat& m) { return m.data == nullptr || m.cols == 0 || m.rows == 0; }
    static const uint8_t* data(const Mat& m) { return m.data; }
    static uint8_t* data(Mat& m) { return m.data; }
    static int cols(const Mat& m) { return m.cols; }
    static int rows(const Mat& m) { return m.rows; }
    static size_t step(const Mat& m) { return static_cast<size_t>(m.step); }
    static void create_u8(Mat& m, int rows, int cols) { m.create(rows, cols); }
    static void release(Mat& m) { m.release(); }
    // only the image preparation (namespace util below) asks for these two
    static int channels(const Mat& m) { return m.channels(); }
    static int elem_size1(const Mat& m) { return static_cast<int>(m.elemSize1()); } // bytes per channel value
};

inline ovs_grid make_grid(float min_x, float min_y, float max_x, float max_y, double inv_cell_width, double inv_cell_height,
                          unsigned num_cols, unsigned num_rows) {
    ovs_grid g;
    g.min_x = min_x; g.min_y = min_y; g.max_x = max_x; g.max_y = max_y;
    g.inv_cell_width = inv_cell_width; g.inv_cell_height = inv_cell_height;
    g.num_cols = num_cols; g.num_rows = num_rows;
    return g;
}

// camera::base keeps the grid geometry (camera/base.h:117-130)
template <class Camera>
ovs_grid grid_of(const Camera& cam) {
    return make_grid(cam.img_bounds_.min_x_, cam.img_bounds_.min_y_, cam.img_bounds_.max_x_, cam.img_bounds_.max_y_,
                     cam.inv_cell_width_, cam.inv_cell_height_, cam.num_grid_cols_, cam.num_grid_rows_);
}

// Flattened copy of what the matchers read from a frame / keyframe.  cv::KeyPoint has the layout
// of ovs_keypoint (checked at compile time where it is used).
struct frame_arrays {
    std::vector<ovs_keypoint> undist_keypts;
    std::vector<uint8_t> descriptors;
    std::vector<float> stereo_x_right;
    std::vector<float> scale_factors;
    ovs_frame_view view;
};

template <class Frame>
void flatten_frame(const Frame& frm, frame_arrays& out) {
    typedef typename std::remove_reference<decltype(frm.undist_keypts_.at(0))>::type keypoint_t;
    static_assert(sizeof(keypoint_t) == sizeof(ovs_keypoint), "cv::KeyPoint layout expected");
    const size_t n = frm.undist_keypts_.size();
    out.undist_keypts.resize(n);
    if (n) {
        std::memcpy(out.undist_keypts.data(), frm.undist_keypts_.data(), n * sizeof(ovs_keypoint));
    }
    typedef mat_traits<typename std::remove_cv<typename std::remove_reference<decltype(frm.descriptors_)>::type>::type> mt;
    out.descriptors.resize(n * 32);
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(out.descriptors.data() + 32 * i, mt::data(frm.descriptors_) + mt::step(frm.descriptors_) * i, 32);
    }
    out.stereo_x_right.assign(frm.stereo_x_right_.begin(), frm.stereo_x_right_.end());
    out.scale_factors.assign(frm.scale_factors_.begin(), frm.scale_factors_.end());
    out.view.num_keypts = static_cast<int32_t>(n);
    out.view.undist_keypts = out.undist_keypts.data();
    out.view.descriptors = out.descriptors.data();
    out.view.stereo_x_right = out.stereo_x_right.size() == n && n ? out.stereo_x_right.data() : nullptr;
    out.view.scale_factors = out.scale_factors.data();
    out.view.num_scale_levels = static_cast<int32_t>(out.scale_factors.size());
    out.view.grid = grid_of(*frm.camera_);
}

struct query_arrays {
    std::vector<float> reproj_xy, x_right, angle;
    std::vector<int32_t> scale_level;
    std::vector<uint8_t> desc, valid, has_observation;
    ovs_proj_queries q;
    void reserve(size_t n) {
        reproj_xy.reserve(2 * n); x_right.reserve(n); angle.reserve(n); scale_level.reserve(n);
        desc.reserve(32 * n); valid.reserve(n); has_observation.reserve(n);
    }
    template <class Mat>
    void push(float x, float y, float xr, int level, float ang, const Mat& d, bool is_valid, bool has_obs) {
        reproj_xy.push_back(x); reproj_xy.push_back(y); x_right.push_back(xr); scale_level.push_back(level); angle.push_back(ang);
        const size_t o = desc.size();
        desc.resize(o + 32, 0);
        if (is_valid) {
            std::memcpy(desc.data() + o, mat_traits<Mat>::data(d), 32);
        }
        valid.push_back(is_valid ? 1 : 0);
        has_observation.push_back(has_obs ? 1 : 0);
    }
    const ovs_proj_queries* finish() {
        q.n = static_cast<int32_t>(valid.size());
        q.reproj_xy = reproj_xy.data(); q.x_right = x_right.data(); q.scale_level = scale_level.data(); q.angle = angle.data();
        q.desc = desc.data(); q.valid = valid.data(); q.has_observation = has_observation.data();
        return &q;
    }
};

template <class LandmarkPtrVec>
std::vector<uint8_t> occupied_flags(const LandmarkPtrVec& lms) {
    // `landmarks_.at(idx) && landmarks_.at(idx)->has_observation()`, projection.cc:41,164
    std::vector<uint8_t> occ(lms.size(), 0);
    for (size_t i = 0; i < lms.size(); ++i) {
        occ[i] = (lms[i] && lms[i]->has_observation()) ? 1 : 0;
    }
    return occ;
}

namespace feature {

struct orb_params {
    unsigned int max_num_keypts_ = 2000;
    float scale_factor_ = 1.2f;
    unsigned int num_levels_ = 8;
    unsigned int ini_fast_thr_ = 20;
    unsigned int min_fast_thr = 7;
    std::vector<std::vector<float>> mask_rects_;
};

// orb_extractor.h:18-78.  One instance per camera stream and thread, as in the reference.
class orb_extractor {
public:
    orb_extractor() = delete;
    explicit orb_extractor(const orb_params& params, int device = 0) {
        std::vector<float> rects;
        for (const auto& r : params.mask_rects_) {
            if (r.size() != 4) {
                throw std::runtime_error("mask rectangle must contain four parameters");
            }
            rects.insert(rects.end(), r.begin(), r.end());
        }
        ovs_orb_params p;
        p.max_num_keypts = params.max_num_keypts_;
        p.scale_factor = params.scale_factor_;
        p.num_levels = params.num_levels_;
        p.ini_fast_thr = params.ini_fast_thr_;
        p.min_fast_thr = params.min_fast_thr;
        p.num_mask_rects = static_cast<uint32_t>(params.mask_rects_.size());
        p.mask_rects = rects.empty() ? nullptr : rects.data();
        check(ovs_orb_create(&p, device, &h_));
    }
    orb_extractor(unsigned int max_num_keypts, float scale_factor, unsigned int num_levels, unsigned int ini_fast_thr,
                  unsigned int min_fast_thr, const std::vector<std::vector<float>>& mask_rects = {}, int device = 0)
        : orb_extractor(make_params(max_num_keypts, scale_factor, num_levels, ini_fast_thr, min_fast_thr, mask_rects), device) {}
    orb_extractor(const orb_extractor&) = delete;
    orb_extractor& operator=(const orb_extractor&) = delete;
    virtual ~orb_extractor() { ovs_orb_destroy(h_); }

    // which cv::GaussianBlur the surrounding build links (OVS_BLUR_ROUNDED: OpenCV up to 4.1.x, the default;
    // OVS_BLUR_NORMALISED: 3.4.9 / 4.2.0 and newer); not part of the reference's class
    void set_blur_kernel(int kernel_variant) { check(ovs_orb_set_blur_kernel(h_, kernel_variant)); }
    int get_blur_kernel() const { return ovs_orb_get_blur_kernel(h_); }

    // extract(in_image, in_image_mask, keypts, out_descriptors), orb_extractor.cc:68-144
    template <class Mat, class KeyPoint>
    void extract(const Mat& in_image, const Mat& in_image_mask, std::vector<KeyPoint>& keypts, Mat& out_descriptors) {
        static_assert(sizeof(KeyPoint) == sizeof(ovs_keypoint), "cv::KeyPoint layout expected");
        typedef mat_traits<Mat> mt;
        if (mt::empty(in_image)) {
            return;
        }
        const int w = mt::cols(in_image), h = mt::rows(in_image);
        const int cap = ovs_orb_keypoint_capacity(h_, w, h);
        if (cap < 0) {
            check(cap);
        }
        kps_.resize(static_cast<size_t>(cap));
        desc_.resize(static_cast<size_t>(cap) * 32);
        int n = 0;
        const bool has_mask = !mt::empty(in_image_mask);
        check(ovs_orb_extract(h_, mt::data(in_image), w, h, mt::step(in_image), has_mask ? mt::data(in_image_mask) : nullptr,
                              has_mask ? mt::step(in_image_mask) : 0, kps_.data(), desc_.data(), cap, &n));
        // image_pyramid_ is public in the reference and read by match::stereo after extract(): here
        // its entries name the levels that stay on the device inside this extractor
        image_pyramid_.assign(get_num_scale_levels(), plane());
        for (unsigned int lv = 0; lv < image_pyramid_.size(); ++lv) {
            image_pyramid_[lv].owner = this;
            image_pyramid_[lv].level = lv;
            check(ovs_orb_level_size(h_, w, h, static_cast<int>(lv), &image_pyramid_[lv].width, &image_pyramid_[lv].height));
        }
        keypts.resize(static_cast<size_t>(n));
        if (n == 0) {
            mt::release(out_descriptors);
            return;
        }
        std::memcpy(static_cast<void*>(keypts.data()), kps_.data(), static_cast<size_t>(n) * sizeof(ovs_keypoint));
        mt::create_u8(out_descriptors, n, 32);
        for (int i = 0; i < n; ++i) {
            std::memcpy(mt::data(out_descriptors) + mt::step(out_descriptors) * i, desc_.data() + 32 * static_cast<size_t>(i), 32);
        }
        last_w_ = w;
        last_h_ = h;
        last_level0_ = mt::data(in_image);
        last_level0_step_ = mt::step(in_image);
    }

    unsigned int get_max_num_keypoints() const { return ovs_orb_get_max_num_keypoints(h_); }
    void set_max_num_keypoints(unsigned int v) { check(ovs_orb_set_max_num_keypoints(h_, v)); }
    float get_scale_factor() const { return ovs_orb_get_scale_factor(h_); }
    void set_scale_factor(float v) { check(ovs_orb_set_scale_factor(h_, v)); }
    unsigned int get_num_scale_levels() const { return ovs_orb_get_num_scale_levels(h_); }
    void set_num_scale_levels(unsigned int v) { check(ovs_orb_set_num_scale_levels(h_, v)); }
    unsigned int get_initial_fast_threshold() const { return ovs_orb_get_initial_fast_threshold(h_); }
    void set_initial_fast_threshold(unsigned int v) { check(ovs_orb_set_initial_fast_threshold(h_, v)); }
    unsigned int get_minimum_fast_threshold() const { return ovs_orb_get_minimum_fast_threshold(h_); }
    void set_minimum_fast_threshold(unsigned int v) { check(ovs_orb_set_minimum_fast_threshold(h_, v)); }
    std::vector<float> get_scale_factors() const { return table(ovs_orb_get_scale_factors); }
    std::vector<float> get_inv_scale_factors() const { return table(ovs_orb_get_inv_scale_factors); }
    std::vector<float> get_level_sigma_sq() const { return table(ovs_orb_get_level_sigma_sq); }
    std::vector<float> get_inv_level_sigma_sq() const { return table(ovs_orb_get_inv_level_sigma_sq); }

    // One level of image_pyramid_.  The entries of image_pyramid_ itself carry no pixels: they name a
    // level resident on the device (owner, level); image_pyramid_level() returns a host copy with pixels
    // (level 0 is the caller's image).
    struct plane {
        std::vector<uint8_t> pixels;
        int width = 0, height = 0;
        orb_extractor* owner = nullptr;
        unsigned int level = 0;
    };
    plane image_pyramid_level(unsigned int level) {
        plane pl;
        check(ovs_orb_level_size(h_, last_w_, last_h_, static_cast<int>(level), &pl.width, &pl.height));
        pl.pixels.resize(static_cast<size_t>(pl.width) * pl.height);
        if (level == 0) {
            for (int y = 0; y < pl.height; ++y) {
                std::memcpy(pl.pixels.data() + static_cast<size_t>(y) * pl.width, last_level0_ + last_level0_step_ * y, pl.width);
            }
        }
        else {
            check(ovs_orb_download_pyramid_level(h_, 0, static_cast<int>(level), pl.pixels.data(), pl.width));
        }
        return pl;
    }
    std::vector<plane> image_pyramid_; // sized like the reference's member; the pixels stay on the device (image_pyramid_level() copies one level)

    ovs_orb* handle() { return h_; }

private:
    static orb_params make_params(unsigned int n, float s, unsigned int l, unsigned int ini, unsigned int mn,
                                  const std::vector<std::vector<float>>& rects) {
        orb_params p;
        p.max_num_keypts_ = n; p.scale_factor_ = s; p.num_levels_ = l; p.ini_fast_thr_ = ini; p.min_fast_thr = mn; p.mask_rects_ = rects;
        return p;
    }
    std::vector<float> table(int (*fn)(const ovs_orb*, float*)) const {
        std::vector<float> t(get_num_scale_levels());
        check(fn(h_, t.data()));
        return t;
    }
    ovs_orb* h_ = nullptr;
    std::vector<ovs_keypoint> kps_;
    std::vector<uint8_t> desc_;
    int last_w_ = 0, last_h_ = 0;
    const uint8_t* last_level0_ = nullptr;
    size_t last_level0_step_ = 0;
};

} // namespace feature

namespace util {

// util/image_converter.h.  camera::color_order_t: Gray = 0, RGB = 1, BGR = 2 (camera/base.h:32-36).
// image_converter.cc:8-39: 3- and 4-channel images become single-channel in place; other images and
// color order Gray are left as they are.
template <class Mat, class ColorOrder>
void convert_to_grayscale(Mat& img, const ColorOrder in_color_order, int device = 0) {
    typedef mat_traits<Mat> mt;
    const int cn = mt::channels(img), order = static_cast<int>(in_color_order);
    if ((cn != 3 && cn != 4) || order == OVS_COLOR_GRAY || mt::empty(img)) {
        return;
    }
    const int w = mt::cols(img), h = mt::rows(img);
    std::vector<uint8_t> gray(static_cast<size_t>(w) * h);
    check(ovs_convert_to_grayscale(mt::data(img), w, h, cn, mt::step(img), order, gray.data(), static_cast<size_t>(w), device));
    mt::create_u8(img, h, w);
    for (int y = 0; y < h; ++y) {
        std::memcpy(mt::data(img) + mt::step(img) * y, gray.data() + static_cast<size_t>(w) * y, static_cast<size_t>(w));
    }
}

// image_converter.cc:45-61: CV_8UC1 or CV_16UC1 in, CV_8UC1 out, in place
template <class Mat>
void equalize_histogram(Mat& img, int device = 0) {
    typedef mat_traits<Mat> mt;
    const int w = mt::cols(img), h = mt::rows(img), bits = 8 * mt::elem_size1(img);
    std::vector<uint8_t> out(static_cast<size_t>(w) * h);
    check(ovs_equalize_histogram(mt::data(img), w, h, mt::step(img), bits, out.data(), static_cast<size_t>(w), device));
    mt::create_u8(img, h, w);
    for (int y = 0; y < h; ++y) {
        std::memcpy(mt::data(img) + mt::step(img) * y, out.data() + static_cast<size_t>(w) * y, static_cast<size_t>(w));
    }
}

// util/stereo_rectifier.h.  The reference's constructors build the four CV_32FC1 maps with
// cv::initUndistortRectifyMap from the StereoRectifier.* entries of the configuration
// (stereo_rectifier.cc:12-53); that one-time host computation stays with the caller, who passes the maps.
class stereo_rectifier {
public:
    template <class MatF>
    stereo_rectifier(const MatF& undist_map_x_l, const MatF& undist_map_y_l, const MatF& undist_map_x_r, const MatF& undist_map_y_r, int device = 0) {
        l_ = make(undist_map_x_l, undist_map_y_l, device, wl_, hl_);
        try {
            r_ = make(undist_map_x_r, undist_map_y_r, device, wr_, hr_);
        }
        catch (...) {
            ovs_remap_destroy(l_);
            throw;
        }
    }
    stereo_rectifier(const stereo_rectifier&) = delete;
    stereo_rectifier& operator=(const stereo_rectifier&) = delete;
    virtual ~stereo_rectifier() {
        ovs_remap_destroy(l_);
        ovs_remap_destroy(r_);
    }

    // stereo_rectifier.cc:56-60
    template <class Mat>
    void rectify(const Mat& in_img_l, const Mat& in_img_r, Mat& out_img_l, Mat& out_img_r) const {
        apply(l_, wl_, hl_, in_img_l, out_img_l);
        apply(r_, wr_, hr_, in_img_r, out_img_r);
    }

private:
    template <class MatF>
    static ovs_remap* make(const MatF& mx, const MatF& my, int device, int& w, int& h) {
        typedef mat_traits<MatF> mt;
        w = mt::cols(mx);
        h = mt::rows(mx);
        if (mt::cols(my) != w || mt::rows(my) != h || mt::elem_size1(mx) != 4 || mt::elem_size1(my) != 4) {
            throw std::runtime_error("rectification maps must be two CV_32FC1 images of one size");
        }
        std::vector<float> x(static_cast<size_t>(w) * h), y(x.size());
        for (int r = 0; r < h; ++r) {
            std::memcpy(x.data() + static_cast<size_t>(w) * r, mt::data(mx) + mt::step(mx) * r, sizeof(float) * w);
            std::memcpy(y.data() + static_cast<size_t>(w) * r, mt::data(my) + mt::step(my) * r, sizeof(float) * w);
        }
        ovs_remap* out = nullptr;
        check(ovs_remap_create(x.data(), y.data(), w, h, device, &out));
        return out;
    }
    template <class Mat>
    static void apply(ovs_remap* r, int w, int h, const Mat& in, Mat& out) {
        typedef mat_traits<Mat> mt;
        std::vector<uint8_t> buf(static_cast<size_t>(w) * h);
        check(ovs_remap_apply(r, mt::data(in), mt::cols(in), mt::rows(in), mt::step(in), buf.data(), static_cast<size_t>(w)));
        mt::create_u8(out, h, w);
        for (int y = 0; y < h; ++y) {
            std::memcpy(mt::data(out) + mt::step(out) * y, buf.data() + static_cast<size_t>(w) * y, static_cast<size_t>(w));
        }
    }
    ovs_remap* l_ = nullptr;
    ovs_remap* r_ = nullptr;
    int wl_ = 0, hl_ = 0, wr_ = 0, hr_ = 0;
};

} // namespace util

namespace camera {

// camera::perspective::convert_keypoints_to_bearings (camera/perspective.cc:130-138) and
// camera::fisheye::convert_keypoints_to_bearings (camera/fisheye.cc:169-177): `cam` is the caller's
// camera object (fx_, fy_, cx_, cy_), `bearings` the caller's eigen_alloc_vector<Vec3_t>
// (three doubles per element), resized as the reference does.
template <class Camera, class KeyPoint, class BearingVec>
void convert_keypoints_to_bearings(const Camera& cam, const std::vector<KeyPoint>& undist_keypts, BearingVec& bearings,
                                   int model = OVS_CAMERA_PERSPECTIVE, int device = 0) {
    static_assert(sizeof(KeyPoint) == sizeof(ovs_keypoint), "cv::KeyPoint layout");
    static_assert(sizeof(typename BearingVec::value_type) == 3 * sizeof(double), "Vec3_t layout");
    ovs_camera c{};
    c.model = model;
    c.fx = cam.fx_; c.fy = cam.fy_; c.cx = cam.cx_; c.cy = cam.cy_;
    bearings.resize(undist_keypts.size());
    check(ovs_convert_keypoints_to_bearings(&c, reinterpret_cast<const ovs_keypoint*>(undist_keypts.data()), static_cast<int>(undist_keypts.size()),
                                            reinterpret_cast<double*>(bearings.data()), device));
}

// camera::equirectangular::convert_keypoints_to_bearings (camera/equirectangular.cc:45-56): cols_, rows_
template <class Camera, class KeyPoint, class BearingVec>
void convert_keypoints_to_bearings_equirectangular(const Camera& cam, const std::vector<KeyPoint>& undist_keypts, BearingVec& bearings, int device = 0) {
    static_assert(sizeof(KeyPoint) == sizeof(ovs_keypoint), "cv::KeyPoint layout");
    static_assert(sizeof(typename BearingVec::value_type) == 3 * sizeof(double), "Vec3_t layout");
    ovs_camera c{};
    c.model = OVS_CAMERA_EQUIRECTANGULAR;
    c.cols = cam.cols_; c.rows = cam.rows_;
    bearings.resize(undist_keypts.size());
    check(ovs_convert_keypoints_to_bearings(&c, reinterpret_cast<const ovs_keypoint*>(undist_keypts.data()), static_cast<int>(undist_keypts.size()),
                                            reinterpret_cast<double*>(bearings.data()), device));
}

// camera::perspective::undistort_keypoints (camera/perspective.cc:99-128): fx_, fy_, cx_, cy_, k1_, k2_, p1_, p2_, k3_
template <class Camera, class KeyPoint>
void undistort_keypoints_perspective(const Camera& cam, const std::vector<KeyPoint>& dist_keypts, std::vector<KeyPoint>& undist_keypts, int device = 0) {
    static_assert(sizeof(KeyPoint) == sizeof(ovs_keypoint), "cv::KeyPoint layout");
    ovs_camera c{};
    c.model = OVS_CAMERA_PERSPECTIVE;
    c.fx = cam.fx_; c.fy = cam.fy_; c.cx = cam.cx_; c.cy = cam.cy_;
    c.dist[0] = cam.k1_; c.dist[1] = cam.k2_; c.dist[2] = cam.p1_; c.dist[3] = cam.p2_; c.dist[4] = cam.k3_;
    undist_keypts.resize(dist_keypts.size());
    check(ovs_undistort_keypoints(&c, reinterpret_cast<const ovs_keypoint*>(dist_keypts.data()), static_cast<int>(dist_keypts.size()),
                                  reinterpret_cast<ovs_keypoint*>(undist_keypts.data()), device));
}

// camera::fisheye::undistort_keypoints (camera/fisheye.cc:139-167): fx_, fy_, cx_, cy_, k1_, k2_, k3_, k4_
template <class Camera, class KeyPoint>
void undistort_keypoints_fisheye(const Camera& cam, const std::vector<KeyPoint>& dist_keypts, std::vector<KeyPoint>& undist_keypts, int device = 0) {
    static_assert(sizeof(KeyPoint) == sizeof(ovs_keypoint), "cv::KeyPoint layout");
    ovs_camera c{};
    c.model = OVS_CAMERA_FISHEYE;
    c.fx = cam.fx_; c.fy = cam.fy_; c.cx = cam.cx_; c.cy = cam.cy_;
    c.dist[0] = cam.k1_; c.dist[1] = cam.k2_; c.dist[2] = cam.k3_; c.dist[3] = cam.k4_;
    undist_keypts.resize(dist_keypts.size());
    check(ovs_undistort_keypoints(&c, reinterpret_cast<const ovs_keypoint*>(dist_keypts.data()), static_cast<int>(dist_keypts.size()),
                                  reinterpret_cast<ovs_keypoint*>(undist_keypts.data()), device));
}

} // namespace camera

namespace data {

// data::bow_vocabulary (data/bow_vocabulary.h:16, the DBoW2 type) as frame::compute_bow and
// keyframe::compute_bow use it (data/frame.cc:162-170, data/keyframe.cc:181-189):
//     bow_vocab_->transform(util::converter::to_desc_vec(descriptors_), bow_vec_, bow_feat_vec_, 4);
// The tree comes in as arrays (ovs_bow_vocab_view: what DBoW2 keeps in m_nodes, which a caller exports from
// its vocabulary once, e.g. from the nodes of the file it loads); the arrays are copied to the device.
// BowVector / FeatureVector are the caller's DBoW2 types: std::map<unsigned, double> and
// std::map<unsigned, std::vector<unsigned>>.
class bow_vocabulary {
public:
    explicit bow_vocabulary(const ovs_bow_vocab_view& view, int device = 0) { check(ovs_bow_vocab_create(&view, device, &h_)); }
    ~bow_vocabulary() { ovs_bow_vocab_destroy(h_); }
    bow_vocabulary(const bow_vocabulary&) = delete;
    bow_vocabulary& operator=(const bow_vocabulary&) = delete;

    // descriptors as the rows of one 8-bit matrix (frame::descriptors_); to_desc_vec only splits it into rows
    template <class Mat, class BowVector, class FeatureVector>
    void transform(const Mat& descriptors, BowVector& v, FeatureVector& fv, int levelsup) const {
        typedef mat_traits<Mat> T;
        const int n = T::empty(descriptors) ? 0 : T::rows(descriptors);
        std::vector<uint8_t> packed(static_cast<size_t>(n) * 32);
        for (int i = 0; i < n; ++i) {
            std::memcpy(packed.data() + 32 * static_cast<size_t>(i), T::data(descriptors) + T::step(descriptors) * static_cast<size_t>(i), 32);
        }
        transform_rows(packed.data(), n, v, fv, levelsup);
    }
    // descriptors as util::converter::to_desc_vec returns them: one 1 x 32 matrix per keypoint
    template <class Mat, class BowVector, class FeatureVector>
    void transform(const std::vector<Mat>& features, BowVector& v, FeatureVector& fv, int levelsup) const {
        typedef mat_traits<Mat> T;
        std::vector<uint8_t> packed(features.size() * 32);
        for (size_t i = 0; i < features.size(); ++i) {
            std::memcpy(packed.data() + 32 * i, T::data(features[i]), 32);
        }
        transform_rows(packed.data(), static_cast<int>(features.size()), v, fv, levelsup);
    }
    ovs_bow_vocab* handle() const { return h_; }

private:
    template <class BowVector, class FeatureVector>
    void transform_rows(const uint8_t* rows, int n, BowVector& v, FeatureVector& fv, int levelsup) const {
        v.clear();
        fv.clear();
        if (n == 0) {
            return;
        }
        const size_t nn = static_cast<size_t>(n);
        std::vector<uint32_t> word_ids(nn), node_ids(nn), offsets(nn + 1), indices(nn);
        std::vector<double> values(nn);
        int32_t num_words = 0, num_nodes = 0;
        check(ovs_bow_transform(h_, rows, n, levelsup, nullptr, nullptr, word_ids.data(), values.data(), &num_words, node_ids.data(), offsets.data(),
                                indices.data(), &num_nodes));
        for (int32_t k = 0; k < num_words; ++k) {
            v.insert(v.end(), typename BowVector::value_type(word_ids[k], values[k]));
        }
        for (int32_t k = 0; k < num_nodes; ++k) {
            typename FeatureVector::mapped_type idx(indices.begin() + offsets[k], indices.begin() + offsets[k + 1]);
            fv.insert(fv.end(), typename FeatureVector::value_type(node_ids[k], std::move(idx)));
        }
    }
    ovs_bow_vocab* h_ = nullptr;
};

} // namespace data

namespace match {

// match/base.h:65-75
class base {
public:
    base(const float lowe_ratio, const bool check_orientation, int device = 0)
        : lowe_ratio_(lowe_ratio), check_orientation_(check_orientation), device_(device) {}
    virtual ~base() = default;

protected:
    const float lowe_ratio_;
    const bool check_orientation_;
    const int device_;
};

class projection final : public base {
public:
    explicit projection(const float lowe_ratio = 0.6, const bool check_orientation = true, int device = 0)
        : base(lowe_ratio, check_orientation, device) {}

    // projection.cc:11-82
    template <class Frame, class Landmark>
    unsigned int match_frame_and_landmarks(Frame& frm, const std::vector<Landmark*>& local_landmarks, const float margin = 5.0) const {
        frame_arrays fa;
        flatten_frame(frm, fa);
        query_arrays qa;
        qa.reserve(local_landmarks.size());
        for (auto* lm : local_landmarks) {
            const bool valid = lm->is_observable_in_tracking_ && !lm->will_be_erased();
            const auto desc = lm->get_descriptor();
            qa.push(lm->reproj_in_tracking_(0), lm->reproj_in_tracking_(1), lm->x_right_in_tracking_,
                    static_cast<int>(lm->scale_level_in_tracking_), 0.0f, desc, valid, lm->has_observation());
        }
        const std::vector<uint8_t> occ = occupied_flags(frm.landmarks_);
        std::vector<int32_t> matched(frm.landmarks_.size(), -1);
        uint32_t num = 0;
        check(ovs_match_frame_and_landmarks(&fa.view, qa.finish(), occ.data(), margin, lowe_ratio_, matched.data(), &num, device_));
        for (size_t i = 0; i < matched.size(); ++i) {
            if (matched[i] >= 0) {
                frm.landmarks_.at(i) = local_landmarks.at(matched[i]);
            }
        }
        return num;
    }

    // projection.cc:84-209
    template <class Frame>
    unsigned int match_current_and_last_frames(Frame& curr_frm, const Frame& last_frm, const float margin) const {
        const auto rot_cw = curr_frm.cam_pose_cw_.template block<3, 3>(0, 0).eval();
        const auto trans_cw = curr_frm.cam_pose_cw_.template block<3, 1>(0, 3).eval();
        const auto trans_wc = (-rot_cw.transpose() * trans_cw).eval();
        const auto rot_lw = last_frm.cam_pose_cw_.template block<3, 3>(0, 0).eval();
        const auto trans_lw = last_frm.cam_pose_cw_.template block<3, 1>(0, 3).eval();
        const auto trans_lc = (rot_lw * trans_wc + trans_lw).eval();
        // camera::setup_type_t::Monocular == 0 (camera/base.h:16-17)
        const bool mono = static_cast<int>(curr_frm.camera_->setup_type_) == 0;
        const bool assume_forward = mono ? false : trans_lc(2) > curr_frm.camera_->true_baseline_;
        const bool assume_backward = mono ? false : -trans_lc(2) > curr_frm.camera_->true_baseline_;
        const int direction = assume_forward ? OVS_DIR_FORWARD : (assume_backward ? OVS_DIR_BACKWARD : OVS_DIR_NONE);

        frame_arrays fa;
        flatten_frame(curr_frm, fa);
        query_arrays qa;
        qa.reserve(last_frm.num_keypts_);
        typedef typename std::remove_reference<decltype(*last_frm.landmarks_.at(0))>::type landmark_t;
        std::vector<landmark_t*> query_lms;
        query_lms.reserve(last_frm.num_keypts_);
        for (unsigned int idx_last = 0; idx_last < last_frm.num_keypts_; ++idx_last) {
            auto* lm = last_frm.landmarks_.at(idx_last);
            if (!lm || last_frm.outlier_flags_.at(idx_last)) {
                continue;
            }
            const auto pos_w = lm->get_pos_in_world();
            typename std::remove_cv<typename std::remove_reference<decltype(lm->reproj_in_tracking_)>::type>::type reproj; // Vec2_t
            float x_right;
            if (!curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) {
                continue;
            }
            const auto desc = lm->get_descriptor();
            qa.push(reproj(0), reproj(1), x_right, last_frm.keypts_.at(idx_last).octave, last_frm.undist_keypts_.at(idx_last).angle,
                    desc, true, lm->has_observation());
            query_lms.push_back(lm);
        }
        const std::vector<uint8_t> occ = occupied_flags(curr_frm.landmarks_);
        std::vector<int32_t> matched(curr_frm.landmarks_.size(), -1);
        uint32_t num = 0;
        check(ovs_match_current_and_last_frames(&fa.view, qa.finish(), occ.data(), margin, direction,
                                                static_cast<int>(last_frm.num_scale_levels_), lowe_ratio_, check_orientation_ ? 1 : 0,
                                                matched.data(), &num, device_));
        for (size_t i = 0; i < matched.size(); ++i) {
            if (matched[i] >= 0) {
                curr_frm.landmarks_.at(i) = query_lms.at(matched[i]);
            }
            else if (matched[i] == -2) {
                curr_frm.landmarks_.at(i) = nullptr; // written, then cleared by the rotation check (projection.cc:203)
            }
        }
        return num;
    }

    // projection.cc:211-313
    template <class Frame, class Keyframe, class LandmarkSet>
    unsigned int match_frame_and_keyframe(Frame& curr_frm, Keyframe* keyfrm, const LandmarkSet& already_matched_lms,
                                          const float margin, const unsigned int hamm_dist_thr) const {
        const auto rot_cw = curr_frm.cam_pose_cw_.template block<3, 3>(0, 0).eval();
        const auto trans_cw = curr_frm.cam_pose_cw_.template block<3, 1>(0, 3).eval();
        const auto cam_center = (-rot_cw.transpose() * trans_cw).eval();
        const auto landmarks = keyfrm->get_landmarks();
        typedef typename std::remove_reference<decltype(*landmarks.at(0))>::type landmark_t;

        frame_arrays fa;
        flatten_frame(curr_frm, fa);
        query_arrays qa;
        qa.reserve(landmarks.size());
        std::vector<landmark_t*> query_lms;
        for (unsigned int idx = 0; idx < landmarks.size(); idx++) {
            auto* lm = landmarks.at(idx);
            if (!lm || lm->will_be_erased() || already_matched_lms.count(lm)) {
                continue;
            }
            const auto pos_w = lm->get_pos_in_world();
            typename std::remove_cv<typename std::remove_reference<decltype(lm->reproj_in_tracking_)>::type>::type reproj; // Vec2_t
            float x_right;
            if (!curr_frm.camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) {
                continue;
            }
            const auto cam_to_lm_vec = (pos_w - cam_center).eval();
            const auto cam_to_lm_dist = cam_to_lm_vec.norm();
            if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) {
                continue;
            }
            const auto pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, &curr_frm);
            const auto desc = lm->get_descriptor();
            qa.push(reproj(0), reproj(1), x_right, static_cast<int>(pred_scale_level), keyfrm->undist_keypts_.at(idx).angle, desc, true, true);
            query_lms.push_back(lm);
        }
        // candidates are skipped when curr_frm.landmarks_.at(i) is non-null (projection.cc:275)
        std::vector<uint8_t> occ(curr_frm.landmarks_.size(), 0);
        for (size_t i = 0; i < occ.size(); ++i) {
            occ[i] = curr_frm.landmarks_[i] ? 1 : 0;
        }
        std::vector<int32_t> matched(occ.size(), -1);
        uint32_t num = 0;
        check(ovs_match_frame_and_keyframe(&fa.view, qa.finish(), occ.data(), margin, hamm_dist_thr, check_orientation_ ? 1 : 0,
                                           matched.data(), &num, device_));
        for (size_t i = 0; i < matched.size(); ++i) {
            if (matched[i] >= 0) {
                curr_frm.landmarks_.at(i) = query_lms.at(matched[i]);
            }
            else if (matched[i] == -2) {
                curr_frm.landmarks_.at(i) = nullptr;
            }
        }
        return num;
    }

    // projection.cc:315-413
    template <class Keyframe, class Mat44, class Landmark>
    unsigned int match_by_Sim3_transform(Keyframe* keyfrm, const Mat44& Sim3_cw, const std::vector<Landmark*>& landmarks,
                                         std::vector<Landmark*>& matched_lms_in_keyfrm, const float margin) const {
        const auto s_rot_cw = Sim3_cw.template block<3, 3>(0, 0).eval();
        const auto row0 = s_rot_cw.template block<1, 3>(0, 0).eval();
        const auto s_cw = std::sqrt(row0.dot(row0));
        const auto rot_cw = (s_rot_cw / s_cw).eval();
        const auto trans_cw = (Sim3_cw.template block<3, 1>(0, 3) / s_cw).eval();
        const auto cam_center = (-rot_cw.transpose() * trans_cw).eval();

        frame_arrays fa;
        flatten_frame(*keyfrm, fa);
        // the level gate of this matcher reads keypts_ (projection.cc:387), the window undist_keypts_
        for (size_t i = 0; i < fa.undist_keypts.size(); ++i) {
            fa.undist_keypts[i].octave = keyfrm->keypts_.at(i).octave;
        }
        query_arrays qa;
        qa.reserve(landmarks.size());
        std::vector<Landmark*> query_lms;
        std::vector<const Landmark*> already_matched; // std::set of the non-null entries, projection.cc:326-327
        for (auto* m : matched_lms_in_keyfrm) {
            if (m) {
                already_matched.push_back(m);
            }
        }
        std::sort(already_matched.begin(), already_matched.end());
        for (auto* lm : landmarks) {
            if (lm->will_be_erased()) {
                continue;
            }
            if (std::binary_search(already_matched.begin(), already_matched.end(), static_cast<const Landmark*>(lm))) {
                continue;
            }
            const auto pos_w = lm->get_pos_in_world();
            typename std::remove_cv<typename std::remove_reference<decltype(lm->reproj_in_tracking_)>::type>::type reproj; // Vec2_t
            float x_right;
            if (!keyfrm->camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) {
                continue;
            }
            const auto cam_to_lm_vec = (pos_w - cam_center).eval();
            const auto cam_to_lm_dist = cam_to_lm_vec.norm();
            if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) {
                continue;
            }
            if (cam_to_lm_vec.dot(lm->get_obs_mean_normal()) < 0.5 * cam_to_lm_dist) {
                continue;
            }
            const auto pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, keyfrm);
            const auto desc = lm->get_descriptor();
            qa.push(reproj(0), reproj(1), x_right, static_cast<int>(pred_scale_level), 0.0f, desc, true, true);
            query_lms.push_back(lm);
        }
        std::vector<uint8_t> occ(matched_lms_in_keyfrm.size(), 0);
        for (size_t i = 0; i < occ.size(); ++i) {
            occ[i] = matched_lms_in_keyfrm[i] ? 1 : 0;
        }
        std::vector<int32_t> matched(occ.size(), -1);
        uint32_t num = 0;
        check(ovs_match_by_sim3_transform(&fa.view, qa.finish(), occ.data(), margin, matched.data(), &num, device_));
        for (size_t i = 0; i < matched.size(); ++i) {
            if (matched[i] >= 0) {
                matched_lms_in_keyfrm.at(i) = query_lms.at(matched[i]);
            }
        }
        return num;
    }

    // projection.cc:415-634: both directions are searched on the device, the cross-check keeps the mutual pairs
    template <class Keyframe, class Landmark, class Mat33, class Vec3>
    unsigned int match_keyframes_mutually(Keyframe* keyfrm_1, Keyframe* keyfrm_2, std::vector<Landmark*>& matched_lms_in_keyfrm_1,
                                          const float& s_12, const Mat33& rot_12, const Vec3& trans_12, const float margin) const {
        const auto rot_1w = keyfrm_1->get_rotation();
        const auto trans_1w = keyfrm_1->get_translation();
        const auto rot_2w = keyfrm_2->get_rotation();
        const auto trans_2w = keyfrm_2->get_translation();
        const auto s_rot_12 = (s_12 * rot_12).eval();
        const auto s_rot_21 = ((1.0 / s_12) * rot_12.transpose()).eval();
        const auto trans_21 = (-s_rot_21 * trans_12).eval();
        const auto landmarks_1 = keyfrm_1->get_landmarks();
        const auto landmarks_2 = keyfrm_2->get_landmarks();

        std::vector<bool> is_already_matched_in_keyfrm_1(landmarks_1.size(), false);
        std::vector<bool> is_already_matched_in_keyfrm_2(landmarks_2.size(), false);
        for (unsigned int idx_1 = 0; idx_1 < landmarks_1.size(); ++idx_1) {
            auto* lm = matched_lms_in_keyfrm_1.at(idx_1);
            if (!lm) {
                continue;
            }
            const auto idx_2 = lm->get_index_in_keyframe(keyfrm_2);
            if (0 <= idx_2 && idx_2 < static_cast<int>(landmarks_2.size())) {
                is_already_matched_in_keyfrm_1.at(idx_1) = true;
                is_already_matched_in_keyfrm_2.at(idx_2) = true;
            }
        }
        const auto s_rot_21w = (s_rot_21 * rot_1w).eval();
        const auto trans_21w = (s_rot_21 * trans_1w + trans_21).eval();
        const auto s_rot_12w = (s_rot_12 * rot_2w).eval();
        const auto trans_12w = (s_rot_12 * trans_2w + trans_12).eval();
        const std::vector<int32_t> idx2_of_1 = one_way(landmarks_1, is_already_matched_in_keyfrm_1, keyfrm_2, s_rot_21w, trans_21w, margin);
        const std::vector<int32_t> idx1_of_2 = one_way(landmarks_2, is_already_matched_in_keyfrm_2, keyfrm_1, s_rot_12w, trans_12w, margin);
        std::vector<int32_t> mutual(landmarks_1.size(), -1);
        uint32_t num = 0;
        check(ovs_cross_check(idx2_of_1.data(), static_cast<int>(idx2_of_1.size()), idx1_of_2.data(), static_cast<int>(idx1_of_2.size()),
                              mutual.data(), &num));
        for (size_t i = 0; i < mutual.size(); ++i) {
            if (mutual[i] >= 0) {
                matched_lms_in_keyfrm_1.at(i) = landmarks_2.at(mutual[i]);
            }
        }
        return num;
    }

private:
    // one direction (projection.cc:457-533): landmarks of `src` projected into `dst`; returns per
    // source keypoint index the best keypoint of dst or -1
    template <class LandmarkVec, class Keyframe, class Mat33, class Vec3>
    std::vector<int32_t> one_way(const LandmarkVec& src_lms, const std::vector<bool>& already, Keyframe* dst, const Mat33& s_rot, const Vec3& trans,
                                 const float margin) const {
        frame_arrays fa;
        flatten_frame(*dst, fa);
        for (size_t i = 0; i < fa.undist_keypts.size(); ++i) {
            fa.undist_keypts[i].octave = dst->keypts_.at(i).octave; // the level gate reads keypts_ (projection.cc:512)
        }
        query_arrays qa;
        qa.reserve(src_lms.size());
        std::vector<unsigned int> src_idx;
        for (unsigned int idx = 0; idx < src_lms.size(); ++idx) {
            auto* lm = src_lms.at(idx);
            if (!lm || lm->will_be_erased() || already.at(idx)) {
                continue;
            }
            const auto pos_w = lm->get_pos_in_world();
            const auto pos_c = (s_rot * pos_w + trans).eval();
            typename std::remove_cv<typename std::remove_reference<decltype(lm->reproj_in_tracking_)>::type>::type reproj; // Vec2_t
            float x_right;
            if (!dst->camera_->reproject_to_image(s_rot, trans, pos_w, reproj, x_right)) {
                continue;
            }
            const auto cam_to_lm_dist = pos_c.norm();
            if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) {
                continue;
            }
            const auto pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, dst);
            const auto desc = lm->get_descriptor();
            qa.push(reproj(0), reproj(1), x_right, static_cast<int>(pred_scale_level), 0.0f, desc, true, true);
            src_idx.push_back(idx);
        }
        std::vector<int32_t> best(src_idx.size(), -1), out(src_lms.size(), -1);
        if (!src_idx.empty()) {
            check(ovs_match_keyframe_one_way(&fa.view, qa.finish(), margin, best.data(), device_));
        }
        for (size_t k = 0; k < src_idx.size(); ++k) {
            out[src_idx[k]] = best[k];
        }
        return out;
    }
};

// DBoW2::FeatureVector is a std::map<unsigned, std::vector<unsigned>>: flattened in key order
struct bow_arrays {
    std::vector<uint32_t> node_ids, offsets, indices;
    ovs_bow_feat_vec v;
    template <class FeatureVector>
    explicit bow_arrays(const FeatureVector& fv) {
        offsets.push_back(0);
        for (const auto& kv : fv) {
            node_ids.push_back(static_cast<uint32_t>(kv.first));
            indices.insert(indices.end(), kv.second.begin(), kv.second.end());
            offsets.push_back(static_cast<uint32_t>(indices.size()));
        }
        v.num_nodes = static_cast<int32_t>(node_ids.size());
        v.node_ids = node_ids.data(); v.offsets = offsets.data(); v.indices = indices.data();
    }
};

class robust final : public base {
public:
    explicit robust(const float lowe_ratio = 0.75, const bool check_orientation = true, int device = 0)
        : base(lowe_ratio, check_orientation, device) {}

    // robust.cc:209-315
    template <class Frame, class Keyframe>
    unsigned int brute_force_match(Frame& frm, Keyframe* keyfrm, std::vector<std::pair<int, int>>& matches) const {
        frame_arrays f1, f2;
        flatten_frame(frm, f1);
        flatten_frame(*keyfrm, f2);
        // the rotation check of this matcher uses keypts_, not undist_keypts_ (robust.cc:289)
        std::vector<float> a1(frm.keypts_.size()), a2(keyfrm->keypts_.size());
        for (size_t i = 0; i < a1.size(); ++i) a1[i] = frm.keypts_[i].angle;
        for (size_t i = 0; i < a2.size(); ++i) a2[i] = keyfrm->keypts_[i].angle;
        const auto lms_2 = keyfrm->get_landmarks();
        std::vector<uint8_t> valid_2(lms_2.size(), 0);
        for (size_t i = 0; i < lms_2.size(); ++i) {
            valid_2[i] = (lms_2[i] && !lms_2[i]->will_be_erased()) ? 1 : 0;
        }
        std::vector<int32_t> pairs(2 * a1.size() + 2);
        uint32_t num = 0;
        check(ovs_brute_force_match(f1.descriptors.data(), a1.data(), static_cast<int>(a1.size()), f2.descriptors.data(), a2.data(),
                                    valid_2.data(), static_cast<int>(a2.size()), lowe_ratio_, check_orientation_ ? 1 : 0,
                                    pairs.data(), &num, device_));
        matches.clear();
        matches.reserve(num);
        for (uint32_t i = 0; i < num; ++i) {
            matches.emplace_back(pairs[2 * i], pairs[2 * i + 1]);
        }
        return num;
    }

    // robust.cc:18-172
    template <class Keyframe, class Mat33>
    unsigned int match_for_triangulation(Keyframe* keyfrm_1, Keyframe* keyfrm_2, const Mat33& E_12,
                                         std::vector<std::pair<unsigned int, unsigned int>>& matched_idx_pairs) const {
        const auto cam_center_1 = keyfrm_1->get_cam_center();
        const auto rot_2w = keyfrm_2->get_rotation();
        const auto trans_2w = keyfrm_2->get_translation();
        typename std::remove_cv<typename std::remove_reference<decltype(keyfrm_2->bearings_.at(0))>::type>::type epiplane_in_keyfrm_2; // Vec3_t
        keyfrm_2->camera_->reproject_to_bearing(rot_2w, trans_2w, cam_center_1, epiplane_in_keyfrm_2);
        frame_arrays f1, f2;
        flatten_frame(*keyfrm_1, f1);
        flatten_frame(*keyfrm_2, f2);
        const size_t n1 = f1.undist_keypts.size(), n2 = f2.undist_keypts.size();
        std::vector<double> b1(3 * n1), b2(3 * n2);
        for (size_t i = 0; i < n1; ++i) for (int k = 0; k < 3; ++k) b1[3 * i + k] = keyfrm_1->bearings_.at(i)(k);
        for (size_t i = 0; i < n2; ++i) for (int k = 0; k < 3; ++k) b2[3 * i + k] = keyfrm_2->bearings_.at(i)(k);
        const auto lms_1 = keyfrm_1->get_landmarks();
        const auto lms_2 = keyfrm_2->get_landmarks();
        std::vector<uint8_t> has_1(n1, 0), has_2(n2, 0);
        for (size_t i = 0; i < n1; ++i) has_1[i] = lms_1.at(i) ? 1 : 0;
        for (size_t i = 0; i < n2; ++i) has_2[i] = lms_2.at(i) ? 1 : 0;
        double E[9], epipole[3];
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) E[3 * i + j] = E_12(i, j);
            epipole[i] = epiplane_in_keyfrm_2(i);
        }
        const bow_arrays v1(keyfrm_1->bow_feat_vec_), v2(keyfrm_2->bow_feat_vec_);
        std::vector<uint32_t> pairs(2 * n1 + 2);
        uint32_t num = 0;
        check(ovs_match_for_triangulation(&v1.v, f1.descriptors.data(), f1.undist_keypts.data(), b1.data(), f1.view.stereo_x_right, has_1.data(),
                                          static_cast<int>(n1), &v2.v, f2.descriptors.data(), f2.undist_keypts.data(), b2.data(),
                                          f2.view.stereo_x_right, has_2.data(), static_cast<int>(n2), E, epipole, f1.scale_factors.data(),
                                          static_cast<int>(f1.scale_factors.size()), check_orientation_ ? 1 : 0, pairs.data(), &num, device_));
        matched_idx_pairs.clear();
        matched_idx_pairs.reserve(num);
        for (uint32_t i = 0; i < num; ++i) {
            matched_idx_pairs.emplace_back(pairs[2 * i], pairs[2 * i + 1]);
        }
        return num;
    }

    // robust.cc:174-207.  The eight-point RANSAC stays on the host and is the caller's class: in
    // the reference tree EssentialSolver = solve::essential_solver (constructed from the two bearing
    // vectors and the matches, find_via_ransac(50, false), solution_is_valid(), get_inlier_matches()).
    template <class EssentialSolver, class Frame, class Keyframe, class Landmark>
    unsigned int match_frame_and_keyframe(Frame& frm, Keyframe* keyfrm, std::vector<Landmark*>& matched_lms_in_frm) const {
        const auto keyfrm_lms = keyfrm->get_landmarks();
        unsigned int num_inlier_matches = 0;
        matched_lms_in_frm = std::vector<Landmark*>(frm.num_keypts_, nullptr);
        std::vector<std::pair<int, int>> matches;
        brute_force_match(frm, keyfrm, matches);
        EssentialSolver solver(frm.bearings_, keyfrm->bearings_, matches);
        solver.find_via_ransac(50, false);
        if (!solver.solution_is_valid()) {
            return 0;
        }
        const auto is_inlier_matches = solver.get_inlier_matches();
        for (unsigned int i = 0; i < matches.size(); ++i) {
            if (!is_inlier_matches.at(i)) {
                continue;
            }
            matched_lms_in_frm.at(matches.at(i).first) = keyfrm_lms.at(matches.at(i).second);
            ++num_inlier_matches;
        }
        return num_inlier_matches;
    }
};

class bow_tree final : public base {
public:
    explicit bow_tree(const float lowe_ratio = 0.6, const bool check_orientation = true, int device = 0)
        : base(lowe_ratio, check_orientation, device) {}

    // bow_tree.cc:16-122
    template <class Keyframe, class Frame, class Landmark>
    unsigned int match_frame_and_keyframe(Keyframe* keyfrm, Frame& frm, std::vector<Landmark*>& matched_lms_in_frm) const {
        frame_arrays fk, ff;
        flatten_frame(*keyfrm, fk);
        flatten_frame(frm, ff);
        std::vector<float> ak(fk.undist_keypts.size()), af(ff.undist_keypts.size());
        for (size_t i = 0; i < ak.size(); ++i) ak[i] = fk.undist_keypts[i].angle;
        for (size_t i = 0; i < af.size(); ++i) af[i] = ff.undist_keypts[i].angle;
        const auto keyfrm_lms = keyfrm->get_landmarks();
        std::vector<uint8_t> valid(keyfrm_lms.size(), 0);
        for (size_t i = 0; i < keyfrm_lms.size(); ++i) {
            valid[i] = (keyfrm_lms[i] && !keyfrm_lms[i]->will_be_erased()) ? 1 : 0;
        }
        const bow_arrays bk(keyfrm->bow_feat_vec_), bf(frm.bow_feat_vec_);
        std::vector<int32_t> matched(af.size(), -1);
        uint32_t num = 0;
        check(ovs_bow_match_frame_and_keyframe(&bk.v, fk.descriptors.data(), ak.data(), valid.data(), static_cast<int>(ak.size()),
                                               &bf.v, ff.descriptors.data(), af.data(), static_cast<int>(af.size()),
                                               lowe_ratio_, check_orientation_ ? 1 : 0, matched.data(), &num, device_));
        matched_lms_in_frm = std::vector<Landmark*>(af.size(), nullptr);
        for (size_t i = 0; i < matched.size(); ++i) {
            if (matched[i] >= 0) {
                matched_lms_in_frm[i] = keyfrm_lms.at(matched[i]);
            }
        }
        return num;
    }

    // bow_tree.cc:124-246
    template <class Keyframe, class Landmark>
    unsigned int match_keyframes(Keyframe* keyfrm_1, Keyframe* keyfrm_2, std::vector<Landmark*>& matched_lms_in_keyfrm_1) const {
        frame_arrays f1, f2;
        flatten_frame(*keyfrm_1, f1);
        flatten_frame(*keyfrm_2, f2);
        std::vector<float> a1(f1.undist_keypts.size()), a2(f2.undist_keypts.size());
        for (size_t i = 0; i < a1.size(); ++i) a1[i] = f1.undist_keypts[i].angle;
        for (size_t i = 0; i < a2.size(); ++i) a2[i] = f2.undist_keypts[i].angle;
        const auto lms_1 = keyfrm_1->get_landmarks();
        const auto lms_2 = keyfrm_2->get_landmarks();
        std::vector<uint8_t> v1(lms_1.size(), 0), v2(lms_2.size(), 0);
        for (size_t i = 0; i < v1.size(); ++i) v1[i] = (lms_1[i] && !lms_1[i]->will_be_erased()) ? 1 : 0;
        for (size_t i = 0; i < v2.size(); ++i) v2[i] = (lms_2[i] && !lms_2[i]->will_be_erased()) ? 1 : 0;
        const bow_arrays b1(keyfrm_1->bow_feat_vec_), b2(keyfrm_2->bow_feat_vec_);
        std::vector<int32_t> matched(a1.size(), -1);
        uint32_t num = 0;
        check(ovs_bow_match_keyframes(&b1.v, f1.descriptors.data(), a1.data(), v1.data(), static_cast<int>(a1.size()),
                                      &b2.v, f2.descriptors.data(), a2.data(), v2.data(), static_cast<int>(a2.size()),
                                      lowe_ratio_, check_orientation_ ? 1 : 0, matched.data(), &num, device_));
        matched_lms_in_keyfrm_1 = std::vector<Landmark*>(a1.size(), nullptr);
        for (size_t i = 0; i < matched.size(); ++i) {
            if (matched[i] >= 0) {
                matched_lms_in_keyfrm_1[i] = lms_2.at(matched[i]);
            }
        }
        return num;
    }
};

// match/area.h:13-25 (monocular initializer)
class area final : public base {
public:
    area(const float lowe_ratio, const bool check_orientation, int device = 0)
        : base(lowe_ratio, check_orientation, device) {}

    // area.cc:8-112
    template <class Frame, class Point2f>
    unsigned int match_in_consistent_area(Frame& frm_1, Frame& frm_2, std::vector<Point2f>& prev_matched_pts,
                                          std::vector<int>& matched_indices_2_in_frm_1, int margin = 10) const {
        static_assert(sizeof(Point2f) == 2 * sizeof(float), "cv::Point2f layout expected");
        frame_arrays f1, f2;
        flatten_frame(frm_1, f1);
        flatten_frame(frm_2, f2);
        const size_t n1 = f1.undist_keypts.size();
        if (prev_matched_pts.size() < n1) {
            throw std::out_of_range("prev_matched_pts is shorter than the keypoints of frame 1"); // .at() in area.cc:29
        }
        std::vector<float> pts(2 * n1);
        if (n1) {
            std::memcpy(pts.data(), static_cast<const void*>(prev_matched_pts.data()), 2 * n1 * sizeof(float));
        }
        std::vector<int32_t> matched(n1, -1);
        uint32_t num = 0;
        check(ovs_match_in_consistent_area(f1.undist_keypts.data(), f1.descriptors.data(), static_cast<int>(n1), &f2.view, pts.data(), margin,
                                           lowe_ratio_, check_orientation_ ? 1 : 0, matched.data(), &num, device_));
        matched_indices_2_in_frm_1.assign(matched.begin(), matched.end());
        if (n1) {
            std::memcpy(static_cast<void*>(prev_matched_pts.data()), pts.data(), 2 * n1 * sizeof(float));
        }
        return num;
    }
};

// match/fuse.h:17-37.  The candidate search of all landmarks runs on the device in one call; the
// landmark bookkeeping (add_observation / add_landmark / replace) then runs on the host in the
// reference's order, re-evaluating the gates that earlier bookkeeping of the same call can close
// (will_be_erased, is_observed_in_keyframe), so the outcome is that of the sequential loop as long
// as a landmark is listed once.
class fuse final : public base {
public:
    explicit fuse(const float lowe_ratio = 0.6, int device = 0)
        : base(lowe_ratio, true, device) {}

    // fuse.cc:12-123
    template <class Keyframe, class Mat44, class Landmark>
    unsigned int detect_duplication(Keyframe* keyfrm, const Mat44& Sim3_cw, const std::vector<Landmark*>& landmarks_to_check,
                                    const float margin, std::vector<Landmark*>& duplicated_lms_in_keyfrm) const {
        const auto s_rot_cw = Sim3_cw.template block<3, 3>(0, 0).eval();
        const auto row0 = s_rot_cw.template block<1, 3>(0, 0).eval();
        const auto s_cw = std::sqrt(row0.dot(row0));
        const auto rot_cw = (s_rot_cw / s_cw).eval();
        const auto trans_cw = (Sim3_cw.template block<3, 1>(0, 3) / s_cw).eval();
        const auto cam_center = (-rot_cw.transpose() * trans_cw).eval();
        duplicated_lms_in_keyfrm = std::vector<Landmark*>(landmarks_to_check.size(), nullptr);
        const auto valid_lms_in_keyfrm = keyfrm->get_valid_landmarks();

        frame_arrays fa;
        flatten_frame(*keyfrm, fa);
        for (size_t i = 0; i < fa.undist_keypts.size(); ++i) {
            fa.undist_keypts[i].octave = keyfrm->keypts_.at(i).octave; // fuse.cc:83 reads keypts_
        }
        query_arrays qa;
        qa.reserve(landmarks_to_check.size());
        std::vector<unsigned int> query_of;
        for (unsigned int i = 0; i < landmarks_to_check.size(); ++i) {
            auto* lm = landmarks_to_check.at(i);
            if (lm->will_be_erased() || valid_lms_in_keyfrm.count(lm)) {
                continue;
            }
            if (!push_if_visible(keyfrm, lm, rot_cw, trans_cw, cam_center, qa, nullptr)) {
                continue;
            }
            query_of.push_back(i);
        }
        std::vector<int32_t> best(query_of.size(), -1);
        if (!query_of.empty()) {
            check(ovs_fuse_best_candidates(&fa.view, qa.finish(), nullptr, nullptr, margin, 0, best.data(), device_));
        }
        unsigned int num_fused = 0;
        for (size_t k = 0; k < query_of.size(); ++k) {
            if (best[k] < 0) {
                continue;
            }
            auto* lm = landmarks_to_check.at(query_of[k]);
            auto* lm_in_keyfrm = keyfrm->get_landmark(best[k]);
            if (lm_in_keyfrm) {
                if (!lm_in_keyfrm->will_be_erased()) {
                    duplicated_lms_in_keyfrm.at(query_of[k]) = lm_in_keyfrm;
                }
            }
            else {
                lm->add_observation(keyfrm, best[k]);
                keyfrm->add_landmark(lm, best[k]);
            }
            ++num_fused;
        }
        return num_fused;
    }

    // fuse.cc:125-264; T = std::vector<landmark*> or std::unordered_set<landmark*>
    template <class Keyframe, class T>
    unsigned int replace_duplication(Keyframe* keyfrm, const T& landmarks_to_check, const float margin = 3.0) const {
        const auto rot_cw = keyfrm->get_rotation();
        const auto trans_cw = keyfrm->get_translation();
        const auto cam_center = keyfrm->get_cam_center();
        typedef typename std::remove_cv<typename std::remove_reference<decltype(*landmarks_to_check.begin())>::type>::type landmark_ptr;

        frame_arrays fa;
        flatten_frame(*keyfrm, fa);
        std::vector<float> inv_level_sigma_sq(keyfrm->inv_level_sigma_sq_.begin(), keyfrm->inv_level_sigma_sq_.end());
        query_arrays qa;
        qa.reserve(landmarks_to_check.size());
        std::vector<double> reproj_d;
        std::vector<landmark_ptr> query_lms;
        for (const auto lm : landmarks_to_check) {
            if (!lm || lm->will_be_erased() || lm->is_observed_in_keyframe(keyfrm)) {
                continue;
            }
            if (!push_if_visible(keyfrm, lm, rot_cw, trans_cw, cam_center, qa, &reproj_d)) {
                continue;
            }
            query_lms.push_back(lm);
        }
        std::vector<int32_t> best(query_lms.size(), -1);
        if (!query_lms.empty()) {
            check(ovs_fuse_best_candidates(&fa.view, qa.finish(), reproj_d.data(), inv_level_sigma_sq.data(), margin, 1, best.data(), device_));
        }
        unsigned int num_fused = 0;
        for (size_t k = 0; k < query_lms.size(); ++k) {
            auto lm = query_lms[k];
            if (best[k] < 0 || lm->will_be_erased() || lm->is_observed_in_keyframe(keyfrm)) {
                continue;
            }
            auto* lm_in_keyfrm = keyfrm->get_landmark(best[k]);
            if (lm_in_keyfrm) {
                if (!lm_in_keyfrm->will_be_erased()) {
                    if (lm->num_observations() < lm_in_keyfrm->num_observations()) {
                        lm->replace(lm_in_keyfrm);
                    }
                    else {
                        lm_in_keyfrm->replace(lm);
                    }
                }
            }
            else {
                lm->add_observation(keyfrm, best[k]);
                keyfrm->add_landmark(lm, best[k]);
            }
            ++num_fused;
        }
        return num_fused;
    }

private:
    // the visibility gates shared by both functions (fuse.cc:38-65 / 146-173)
    template <class Keyframe, class LandmarkPtr, class Mat33, class Vec3>
    static bool push_if_visible(Keyframe* keyfrm, LandmarkPtr lm, const Mat33& rot_cw, const Vec3& trans_cw, const Vec3& cam_center,
                                query_arrays& qa, std::vector<double>* reproj_d) {
        const auto pos_w = lm->get_pos_in_world();
        typename std::remove_cv<typename std::remove_reference<decltype(lm->reproj_in_tracking_)>::type>::type reproj; // Vec2_t
        float x_right;
        if (!keyfrm->camera_->reproject_to_image(rot_cw, trans_cw, pos_w, reproj, x_right)) {
            return false;
        }
        const auto cam_to_lm_vec = (pos_w - cam_center).eval();
        const auto cam_to_lm_dist = cam_to_lm_vec.norm();
        if (cam_to_lm_dist < lm->get_min_valid_distance() || lm->get_max_valid_distance() < cam_to_lm_dist) {
            return false;
        }
        if (cam_to_lm_vec.dot(lm->get_obs_mean_normal()) < 0.5 * cam_to_lm_dist) {
            return false;
        }
        const int pred_scale_level = static_cast<int>(lm->predict_scale_level(cam_to_lm_dist, keyfrm));
        const auto desc = lm->get_descriptor();
        qa.push(reproj(0), reproj(1), x_right, pred_scale_level, 0.0f, desc, true, true);
        if (reproj_d) {
            reproj_d->push_back(reproj(0));
            reproj_d->push_back(reproj(1));
        }
        return true;
    }
};

// match/stereo.h.  Three ways to name the two image pyramids:
//   * the reference's argument list (stereo.h:18-22) with the extractors' image_pyramid_ members, as
//     data/frame.cc:77-81 writes it: the entries of image_pyramid_ name levels that are still on the
//     device, so nothing but keypoints and descriptors travels (compute() must run before either
//     extractor extracts again, as in the reference's frame constructor);
//   * the same argument list with host images (std::vector of the caller's matrix type, cv::Mat in the
//     reference tree): the pyramids are uploaded by ovs_stereo_compute;
//   * the two extractors themselves.
class stereo {
public:
    stereo() = delete;

    template <class KeyPoint, class Mat>
    stereo(feature::orb_extractor& left, feature::orb_extractor& right, const std::vector<KeyPoint>& keypts_left,
           const std::vector<KeyPoint>& keypts_right, const Mat& descs_left, const Mat& descs_right,
           const std::vector<float>& scale_factors, const std::vector<float>& inv_scale_factors,
           const float focal_x_baseline, const float true_baseline, int device = 0)
        : left_(&left), right_(&right), scale_factors_(scale_factors), inv_scale_factors_(inv_scale_factors),
          focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline), device_(device) {
        take(keypts_left, keypts_right, descs_left, descs_right);
    }

    // stereo.h:18-22 with left_image_pyramid = extractor_left->image_pyramid_ (resident levels), or with
    // planes that hold pixels (image_pyramid_level() copies)
    template <class KeyPoint, class Mat>
    stereo(const std::vector<feature::orb_extractor::plane>& left_image_pyramid, const std::vector<feature::orb_extractor::plane>& right_image_pyramid,
           const std::vector<KeyPoint>& keypts_left, const std::vector<KeyPoint>& keypts_right, const Mat& descs_left, const Mat& descs_right,
           const std::vector<float>& scale_factors, const std::vector<float>& inv_scale_factors,
           const float focal_x_baseline, const float true_baseline, int device = 0)
        : scale_factors_(scale_factors), inv_scale_factors_(inv_scale_factors),
          focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline), device_(device) {
        if (left_image_pyramid.size() != right_image_pyramid.size() || left_image_pyramid.empty()) {
            throw std::runtime_error("match::stereo: the two image pyramids must have the same, non-zero number of levels");
        }
        const bool resident = left_image_pyramid[0].owner && right_image_pyramid[0].owner && left_image_pyramid[0].pixels.empty()
                              && right_image_pyramid[0].pixels.empty();
        if (resident) {
            left_ = left_image_pyramid[0].owner;
            right_ = right_image_pyramid[0].owner;
        }
        else {
            for (size_t lv = 0; lv < left_image_pyramid.size(); ++lv) {
                const feature::orb_extractor::plane &l = left_image_pyramid[lv], &r = right_image_pyramid[lv];
                if (l.pixels.size() != static_cast<size_t>(l.width) * l.height || r.pixels.size() != static_cast<size_t>(r.width) * r.height) {
                    throw std::runtime_error("match::stereo: pyramid level without pixels");
                }
                pyr_l_.push_back(ovs_image_view{l.pixels.data(), l.width, l.height, static_cast<size_t>(l.width)});
                pyr_r_.push_back(ovs_image_view{r.pixels.data(), r.width, r.height, static_cast<size_t>(r.width)});
            }
        }
        take(keypts_left, keypts_right, descs_left, descs_right);
    }

    // stereo.h:18-22 with host images (the caller keeps them alive until compute() returns, as the
    // reference's cv::Mat headers keep theirs)
    template <class Img, class KeyPoint, class Mat>
    stereo(const std::vector<Img>& left_image_pyramid, const std::vector<Img>& right_image_pyramid,
           const std::vector<KeyPoint>& keypts_left, const std::vector<KeyPoint>& keypts_right, const Mat& descs_left, const Mat& descs_right,
           const std::vector<float>& scale_factors, const std::vector<float>& inv_scale_factors,
           const float focal_x_baseline, const float true_baseline, int device = 0)
        : scale_factors_(scale_factors), inv_scale_factors_(inv_scale_factors),
          focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline), device_(device) {
        if (left_image_pyramid.size() != right_image_pyramid.size() || left_image_pyramid.empty()) {
            throw std::runtime_error("match::stereo: the two image pyramids must have the same, non-zero number of levels");
        }
        typedef mat_traits<Img> it;
        for (size_t lv = 0; lv < left_image_pyramid.size(); ++lv) {
            const Img &l = left_image_pyramid[lv], &r = right_image_pyramid[lv];
            pyr_l_.push_back(ovs_image_view{it::data(l), it::cols(l), it::rows(l), it::step(l)});
            pyr_r_.push_back(ovs_image_view{it::data(r), it::cols(r), it::rows(r), it::step(r)});
        }
        take(keypts_left, keypts_right, descs_left, descs_right);
    }

    virtual ~stereo() = default;

    // stereo.cc:18-113
    void compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const {
        stereo_x_right.assign(kl_.size(), -1.0f);
        depths.assign(kl_.size(), -1.0f);
        if (left_ && right_) {
            check(ovs_stereo_compute_resident(left_->handle(), right_->handle(), kl_.data(), static_cast<int>(kl_.size()), kr_.data(),
                                              static_cast<int>(kr_.size()), dl_.data(), dr_.data(), focal_x_baseline_, true_baseline_,
                                              stereo_x_right.data(), depths.data(), device_));
            return;
        }
        if (scale_factors_.size() < pyr_l_.size() || inv_scale_factors_.size() < pyr_l_.size()) {
            throw std::runtime_error("match::stereo: scale factor tables shorter than the image pyramid");
        }
        check(ovs_stereo_compute(pyr_l_.data(), pyr_r_.data(), static_cast<int>(pyr_l_.size()), kl_.data(), static_cast<int>(kl_.size()),
                                 kr_.data(), static_cast<int>(kr_.size()), dl_.data(), dr_.data(), scale_factors_.data(), inv_scale_factors_.data(),
                                 focal_x_baseline_, true_baseline_, stereo_x_right.data(), depths.data(), device_));
    }

private:
    template <class KeyPoint, class Mat>
    void take(const std::vector<KeyPoint>& keypts_left, const std::vector<KeyPoint>& keypts_right, const Mat& descs_left, const Mat& descs_right) {
        static_assert(sizeof(KeyPoint) == sizeof(ovs_keypoint), "cv::KeyPoint layout expected");
        kl_.resize(keypts_left.size());
        kr_.resize(keypts_right.size());
        if (!kl_.empty()) std::memcpy(kl_.data(), keypts_left.data(), kl_.size() * sizeof(ovs_keypoint));
        if (!kr_.empty()) std::memcpy(kr_.data(), keypts_right.data(), kr_.size() * sizeof(ovs_keypoint));
        typedef mat_traits<Mat> mt;
        dl_.resize(kl_.size() * 32);
        dr_.resize(kr_.size() * 32);
        for (size_t i = 0; i < kl_.size(); ++i) std::memcpy(dl_.data() + 32 * i, mt::data(descs_left) + mt::step(descs_left) * i, 32);
        for (size_t i = 0; i < kr_.size(); ++i) std::memcpy(dr_.data() + 32 * i, mt::data(descs_right) + mt::step(descs_right) * i, 32);
    }

    feature::orb_extractor* left_ = nullptr;  // both set: the pyramids are resident in the two extractors
    feature::orb_extractor* right_ = nullptr;
    std::vector<ovs_image_view> pyr_l_, pyr_r_; // host pyramids otherwise
    std::vector<ovs_keypoint> kl_, kr_;
    std::vector<uint8_t> dl_, dr_;
    const std::vector<float> scale_factors_, inv_scale_factors_;
    const float focal_x_baseline_, true_baseline_;
    const int device_;
};

} // namespace match
} // namespace ovs_shim

#endif
