// Compiles the header-only C++ shim against small stand-ins for the CALLER's types (a frame with
// the member names of data::frame, a landmark, a 4x4 pose with block<>(), an 8-bit matrix) and
// links it with libovs_hip.so.  Without a GPU the program only checks that construction fails
// with the library's "no HIP device" error; with a GPU (argv[1] == "run") it extracts two frames
// and runs the matchers, printing counts for the Python side to compare.  argv[1] == "threads" runs the
// extractors and matchers from four std::threads at once, the way the reference's tracking, mapping and
// global optimisation threads and the two extraction threads of a stereo frame do (system.cc:126-127,
// data/frame.cc:64-67), and compares every result with the one the same call gave on one thread.
#include "ovs_shim.hpp"

#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <set>
#include <atomic>
#include <thread>

namespace mock {

struct vec2 {
    double v[2] = {0, 0};
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
};
struct vec3 {
    double v[3] = {0, 0, 0};
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    vec3 eval() const { return *this; }
    vec3 operator-() const { vec3 r; for (int i = 0; i < 3; ++i) r.v[i] = -v[i]; return r; }
    vec3 operator+(const vec3& o) const { vec3 r; for (int i = 0; i < 3; ++i) r.v[i] = v[i] + o.v[i]; return r; }
    vec3 operator-(const vec3& o) const { vec3 r; for (int i = 0; i < 3; ++i) r.v[i] = v[i] - o.v[i]; return r; }
    vec3 operator/(double s) const { vec3 r; for (int i = 0; i < 3; ++i) r.v[i] = v[i] / s; return r; }
    double dot(const vec3& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    double norm() const { return std::sqrt(dot(*this)); }
};
struct mat33 {
    double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    mat33 eval() const { return *this; }
    mat33 transpose() const { mat33 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[j][i]; return r; }
    mat33 operator-() const { mat33 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = -m[i][j]; return r; }
    vec3 operator*(const vec3& x) const { vec3 r; for (int i = 0; i < 3; ++i) r.v[i] = m[i][0] * x.v[0] + m[i][1] * x.v[1] + m[i][2] * x.v[2]; return r; }
    mat33 operator/(double s) const { mat33 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] / s; return r; }
    template <int R, int C> vec3 block(int r0, int) const { vec3 r; for (int j = 0; j < 3; ++j) r.v[j] = m[r0][j]; return r; } // block<1, 3>
    mat33 operator*(const mat33& o) const {
        mat33 r;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { r.m[i][j] = 0; for (int k = 0; k < 3; ++k) r.m[i][j] += m[i][k] * o.m[k][j]; }
        return r;
    }
    double operator()(int i, int j) const { return m[i][j]; }
};
inline mat33 operator*(double s, const mat33& a) { mat33 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = s * a.m[i][j]; return r; }
template <int R, int C> struct block_type;
template <> struct block_type<3, 3> { typedef mat33 type; };
template <> struct block_type<3, 1> { typedef vec3 type; };
struct mat44 {
    double m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    template <int R, int C>
    typename block_type<R, C>::type block(int r0, int c0) const;
};
template <> mat33 mat44::block<3, 3>(int r0, int c0) const { mat33 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[r0 + i][c0 + j]; return r; }
template <> vec3 mat44::block<3, 1>(int r0, int c0) const { vec3 r; for (int i = 0; i < 3; ++i) r.v[i] = m[r0 + i][c0]; return r; }

struct mat_u8 {
    uint8_t* data = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;
    std::shared_ptr<std::vector<uint8_t>> store;
    void create(int r, int c) {
        store = std::make_shared<std::vector<uint8_t>>(static_cast<size_t>(r) * c);
        data = store->data(); rows = r; cols = c; step = c; cn = 1; esz1 = 1;
    }
    void release() { store.reset(); data = nullptr; rows = cols = 0; step = 0; }
    // multi-channel / wider element images of the image preparation
    int cn = 1, esz1 = 1;
    int channels() const { return cn; }
    size_t elemSize1() const { return static_cast<size_t>(esz1); }
    void create_raw(int r, int c, int channels_, int elem_bytes) {
        store = std::make_shared<std::vector<uint8_t>>(static_cast<size_t>(r) * c * channels_ * elem_bytes);
        data = store->data(); rows = r; cols = c; cn = channels_; esz1 = elem_bytes; step = static_cast<size_t>(c) * channels_ * elem_bytes;
    }
};

struct keypoint { float x, y, size, angle, response; int octave, class_id; };

struct bounds { float min_x_, min_y_, max_x_, max_y_; };
struct camera {
    int setup_type_ = 0; // Monocular
    float true_baseline_ = 0.0f;
    bounds img_bounds_;
    double inv_cell_width_, inv_cell_height_;
    unsigned num_grid_cols_ = 64, num_grid_rows_ = 48;
    double fx = 500, fy = 500, cx = 320, cy = 240;
    bool reproject_to_image(const mat33& rot_cw, const vec3& trans_cw, const vec3& pos_w, vec2& reproj, float& x_right) const {
        const vec3 pc = rot_cw * pos_w + trans_cw;
        if (pc(2) <= 0) return false;
        reproj(0) = fx * pc(0) / pc(2) + cx;
        reproj(1) = fy * pc(1) / pc(2) + cy;
        x_right = -1.0f;
        return img_bounds_.min_x_ <= reproj(0) && reproj(0) <= img_bounds_.max_x_ && img_bounds_.min_y_ <= reproj(1) && reproj(1) <= img_bounds_.max_y_;
    }
    bool reproject_to_bearing(const mat33& rot_cw, const vec3& trans_cw, const vec3& pos_w, vec3& reproj) const {
        const vec3 pc = rot_cw * pos_w + trans_cw;
        reproj = pc / pc.norm();
        return true;
    }
};
struct pinhole { double fx_ = 500, fy_ = 250, cx_ = 320, cy_ = 240, k1_ = -0.2, k2_ = 0.05, p1_ = 0, p2_ = 0, k3_ = 0, k4_ = 0; };
struct sphere { unsigned int cols_ = 640, rows_ = 320; };
struct frame;

struct landmark {
    vec3 pos;
    mat_u8 desc;
    vec2 reproj_in_tracking_;
    float x_right_in_tracking_ = -1.0f;
    bool is_observable_in_tracking_ = true;
    int scale_level_in_tracking_ = 0;
    bool observed = true;
    vec3 get_pos_in_world() const { return pos; }
    float get_min_valid_distance() const { return 0.1f; }
    float get_max_valid_distance() const { return 1.0e9f; } // the toy camera has fx = 1: points sit hundreds of units off axis
    vec3 get_obs_mean_normal() const { return pos / pos.norm(); }
    int level = 0;
    template <class F> unsigned int predict_scale_level(double, const F*) const { return static_cast<unsigned int>(level); }
    mat_u8 get_descriptor() const { return desc; }
    bool has_observation() const { return observed; }
    bool erased = false;
    bool will_be_erased() const { return erased; }
    // observation bookkeeping, enough for match::fuse and match_keyframes_mutually
    std::map<const frame*, int> observations;
    int get_index_in_keyframe(const frame* kf) const { auto it = observations.find(kf); return it == observations.end() ? -1 : it->second; }
    bool is_observed_in_keyframe(const frame* kf) const { return observations.count(kf) != 0; }
    void add_observation(const frame* kf, int idx) { observations[kf] = idx; }
    unsigned int num_observations() const { return static_cast<unsigned int>(observations.size()); }
    landmark* replaced_by = nullptr;
    void replace(landmark* lm);
};

struct frame {
    unsigned int num_keypts_ = 0;
    std::vector<keypoint> keypts_, undist_keypts_;
    mat_u8 descriptors_;
    std::vector<float> stereo_x_right_;
    std::vector<float> scale_factors_;
    unsigned int num_scale_levels_ = 8;
    std::vector<landmark*> landmarks_;
    std::vector<bool> outlier_flags_;
    mat44 cam_pose_cw_;
    camera* camera_ = nullptr;
    std::map<unsigned, std::vector<unsigned>> bow_feat_vec_;
    std::vector<landmark*> get_landmarks() const { return landmarks_; }
    // keyframe side
    std::vector<float> inv_level_sigma_sq_;
    std::vector<vec3> bearings_;
    vec3 cam_center_;
    mat33 get_rotation() const { return cam_pose_cw_.block<3, 3>(0, 0); }
    vec3 get_translation() const { return cam_pose_cw_.block<3, 1>(0, 3); }
    vec3 get_cam_center() const { return cam_center_; }
    landmark* get_landmark(int idx) const { return landmarks_.at(idx); }
    void add_landmark(landmark* lm, int idx) { landmarks_.at(idx) = lm; }
    std::set<landmark*> get_valid_landmarks() const {
        std::set<landmark*> out;
        for (auto* lm : landmarks_) if (lm && !lm->will_be_erased()) out.insert(lm);
        return out;
    }
};

// the part of landmark::replace (data/landmark.cc:300-335) that the matchers can observe
inline void landmark::replace(landmark* lm) {
    if (lm == this) return;
    const auto obs = observations;
    observations.clear();
    erased = true;
    replaced_by = lm;
    for (const auto& kv : obs) {
        auto* kf = const_cast<frame*>(kv.first);
        if (!lm->is_observed_in_keyframe(kf)) { kf->landmarks_.at(kv.second) = lm; lm->add_observation(kf, kv.second); }
        else { kf->landmarks_.at(kv.second) = nullptr; }
    }
}

struct point2f { float x, y; };

// stands in for solve::essential_solver: accepts the matches whose rows agree within two pixels
struct row_solver {
    const std::vector<vec3>& b1; const std::vector<vec3>& b2; const std::vector<std::pair<int, int>>& matches;
    std::vector<bool> inliers;
    row_solver(const std::vector<vec3>& a, const std::vector<vec3>& b, const std::vector<std::pair<int, int>>& m) : b1(a), b2(b), matches(m) {}
    void find_via_ransac(unsigned, bool) {
        for (const auto& m : matches) inliers.push_back(std::fabs(b1.at(m.first)(1) / b1.at(m.first)(2) - b2.at(m.second)(1) / b2.at(m.second)(2)) < 2.0 / 500.0);
    }
    bool solution_is_valid() const { return true; }
    std::vector<bool> get_inlier_matches() const { return inliers; }
};

} // namespace mock

static void make_image(mock::mat_u8& img, int w, int h, int shift) {
    img.create(h, w);
    unsigned s = 12345;
    std::vector<uint8_t> canvas(static_cast<size_t>(w + 64) * h);
    for (auto& v : canvas) { s = s * 1664525u + 1013904223u; v = static_cast<uint8_t>(128 + ((s >> 24) & 31)); }
    for (int k = 0; k < 400; ++k) {
        s = s * 1664525u + 1013904223u; const int x0 = (s >> 8) % (w + 30);
        s = s * 1664525u + 1013904223u; const int y0 = (s >> 8) % (h - 20);
        s = s * 1664525u + 1013904223u; const uint8_t g = static_cast<uint8_t>(s >> 24);
        for (int y = y0; y < y0 + 14; ++y) for (int x = x0; x < x0 + 18; ++x) canvas[static_cast<size_t>(y) * (w + 64) + x] = g;
    }
    for (int y = 0; y < h; ++y) std::memcpy(img.data + static_cast<size_t>(y) * w, canvas.data() + static_cast<size_t>(y) * (w + 64) + shift, w);
}

// Four threads, 20 rounds each, all released together: two extractors on two images, the projection matcher,
// brute force + BoW matching.  Each matcher thread works on its own copies of the frames (the matchers write
// landmarks_), as each of the reference's threads works on its own frame / keyframe objects.
static int run_threads(ovs_shim::feature::orb_extractor& ext_l, const ovs_shim::feature::orb_params& prm, const mock::frame (&frm)[2],
                       mock::mat_u8 (&img)[2]) {
    using namespace ovs_shim;
    feature::orb_extractor ext_r(prm);
    mock::mat_u8 no_mask;
    const int rounds = 20;
    // one-thread results
    mock::frame p1 = frm[1], p0 = frm[0];
    const unsigned want_proj = match::projection(0.9f, true).match_current_and_last_frames(p1, p0, 15.0f);
    const std::vector<mock::landmark*> want_lms = p1.landmarks_;
    mock::frame b1 = frm[1], b0 = frm[0];
    std::vector<std::pair<int, int>> want_pairs;
    const unsigned want_bf = match::robust(0.75f, true).brute_force_match(b1, &b0, want_pairs);
    for (unsigned i = 0; i < b0.num_keypts_; ++i) b0.bow_feat_vec_[3 + i % 40].push_back(i);
    for (unsigned i = 0; i < b1.num_keypts_; ++i) { b1.bow_feat_vec_[3 + i % 40].push_back(i); b1.landmarks_[i] = frm[0].landmarks_[i % frm[0].num_keypts_]; }
    std::vector<mock::landmark*> want_bow_matched;
    const unsigned want_bow = match::bow_tree(0.75f, true).match_keyframes(&b0, &b1, want_bow_matched);

    std::atomic<int> waiting(4), failures(0);
    auto gate = [&]() { --waiting; while (waiting.load() > 0) std::this_thread::yield(); };
    auto same_extraction = [](const std::vector<mock::keypoint>& k, const mock::mat_u8& d, const mock::frame& f) {
        return k.size() == f.keypts_.size() && std::memcmp(k.data(), f.keypts_.data(), k.size() * sizeof(mock::keypoint)) == 0
               && std::memcmp(d.data, f.descriptors_.data, 32 * k.size()) == 0;
    };
    auto extractor_thread = [&](feature::orb_extractor& ext, int which) {
        try {
            gate();
            for (int r = 0; r < rounds; ++r) {
                std::vector<mock::keypoint> k;
                mock::mat_u8 d;
                ext.extract(img[which], no_mask, k, d);
                if (!same_extraction(k, d, frm[which])) ++failures;
            }
        }
        catch (const std::exception& e) { std::printf("extractor thread %d: %s\n", which, e.what()); ++failures; }
    };
    std::thread t_left([&]() { extractor_thread(ext_l, 0); });
    std::thread t_right([&]() { extractor_thread(ext_r, 1); });
    std::thread t_proj([&]() {
        try {
            gate();
            for (int r = 0; r < rounds; ++r) {
                mock::frame c1 = frm[1], c0 = frm[0];
                const unsigned n = match::projection(0.9f, true).match_current_and_last_frames(c1, c0, 15.0f);
                if (n != want_proj || c1.landmarks_ != want_lms) ++failures;
            }
        }
        catch (const std::exception& e) { std::printf("projection thread: %s\n", e.what()); ++failures; }
    });
    std::thread t_bf([&]() {
        try {
            gate();
            for (int r = 0; r < rounds; ++r) {
                std::vector<std::pair<int, int>> pairs;
                mock::frame c1 = b1, c0 = b0;
                c1.landmarks_ = frm[1].landmarks_; // brute force wants the frame without landmarks
                const unsigned n = match::robust(0.75f, true).brute_force_match(c1, &c0, pairs);
                if (n != want_bf || pairs != want_pairs) ++failures;
                std::vector<mock::landmark*> matched;
                const unsigned nb = match::bow_tree(0.75f, true).match_keyframes(&b0, &b1, matched);
                if (nb != want_bow || matched != want_bow_matched) ++failures;
            }
        }
        catch (const std::exception& e) { std::printf("brute force / bow thread: %s\n", e.what()); ++failures; }
    });
    t_left.join(); t_right.join(); t_proj.join(); t_bf.join();
    std::printf("%s threads: 4 x %d rounds, projection %u brute_force %u bow %u, %d results differ from the one-thread run\n",
                failures.load() == 0 ? "OK" : "FAIL", rounds, want_proj, want_bf, want_bow, failures.load());
    return failures.load() == 0 && want_proj > 200 && want_bf > 100 && want_bow > 20 ? 0 : 3;
}

int main(int argc, char** argv) {
    using namespace ovs_shim;
    const bool threads = argc > 1 && std::string(argv[1]) == "threads";
    const bool run = threads || (argc > 1 && std::string(argv[1]) == "run");
    if (!run) {
        try {
            feature::orb_extractor ext(feature::orb_params{});
            if (ovs_device_count() <= 0) { std::puts("FAIL: constructed without a GPU"); return 1; }
            std::puts("OK gpu-present");
        }
        catch (const std::runtime_error& e) {
            std::printf("OK no-gpu: %s\n", e.what());
        }
        return 0;
    }
    const int w = 640, h = 480, dx = 4;
    feature::orb_params prm;
    prm.max_num_keypts_ = 1000;
    feature::orb_extractor ext(prm);
    mock::camera cam;
    cam.img_bounds_ = {0.0f, 0.0f, static_cast<float>(w), static_cast<float>(h)};
    cam.inv_cell_width_ = 64.0 / w; cam.inv_cell_height_ = 48.0 / h;
    cam.cx = 0; cam.cy = 0; cam.fx = 1; cam.fy = 1;
    mock::frame frm[2];
    mock::mat_u8 img[2], no_mask;
    for (int f = 0; f < 2; ++f) {
        make_image(img[f], w, h, f * dx);
        ext.extract(img[f], no_mask, frm[f].keypts_, frm[f].descriptors_);
        frm[f].undist_keypts_ = frm[f].keypts_;
        frm[f].num_keypts_ = static_cast<unsigned>(frm[f].keypts_.size());
        frm[f].stereo_x_right_.assign(frm[f].num_keypts_, -1.0f);
        frm[f].scale_factors_ = ext.get_scale_factors();
        frm[f].landmarks_.assign(frm[f].num_keypts_, nullptr);
        frm[f].outlier_flags_.assign(frm[f].num_keypts_, false);
        frm[f].camera_ = &cam;
        frm[f].inv_level_sigma_sq_ = ext.get_inv_level_sigma_sq();
        for (const auto& kp : frm[f].keypts_) {
            mock::vec3 b; b.v[0] = (kp.x - 320.0) / 500.0; b.v[1] = (kp.y - 240.0) / 500.0; b.v[2] = 1.0;
            frm[f].bearings_.push_back(b / b.norm());
        }
    }
    // every keypoint of frame 0 carries a landmark at depth 1 whose projection (fx = fy = 1, pose
    // = identity) into frame 1 is its pixel position shifted by -dx
    std::vector<std::unique_ptr<mock::landmark>> lms;
    for (unsigned i = 0; i < frm[0].num_keypts_; ++i) {
        lms.emplace_back(new mock::landmark);
        lms.back()->pos.v[0] = frm[0].keypts_[i].x - dx; lms.back()->pos.v[1] = frm[0].keypts_[i].y; lms.back()->pos.v[2] = 1.0;
        lms.back()->desc.create(1, 32);
        std::memcpy(lms.back()->desc.data, frm[0].descriptors_.data + 32 * static_cast<size_t>(i), 32);
        lms.back()->level = frm[0].keypts_[i].octave;
        frm[0].landmarks_[i] = lms.back().get();
    }
    if (threads) {
        return run_threads(ext, prm, frm, img);
    }
    const unsigned n_proj = match::projection(0.9f, true).match_current_and_last_frames(frm[1], frm[0], 15.0f);
    unsigned written = 0;
    for (auto* lm : frm[1].landmarks_) written += lm != nullptr;
    std::vector<std::pair<int, int>> pairs;
    const unsigned n_bf = match::robust(0.75f, true).brute_force_match(frm[1], &frm[0], pairs);
    // relocalisation-style projection of the keyframe's landmarks, Sim3 variant with the identity,
    // BoW matching with every keypoint under one vocabulary node
    frm[1].landmarks_.assign(frm[1].num_keypts_, nullptr);
    const unsigned n_kf = match::projection(0.9f, true).match_frame_and_keyframe(frm[1], &frm[0], std::set<mock::landmark*>{}, 15.0f, 100);
    std::vector<mock::landmark*> all_lms, sim3_matched(frm[1].num_keypts_, nullptr);
    for (auto& lm : lms) all_lms.push_back(lm.get());
    const unsigned n_sim3 = match::projection(0.9f, true).match_by_Sim3_transform(&frm[1], mock::mat44{}, all_lms, sim3_matched, 10.0f);
    for (int f = 0; f < 2; ++f) {
        for (unsigned i = 0; i < frm[f].num_keypts_; ++i) frm[f].bow_feat_vec_[7].push_back(i);
    }
    std::vector<mock::landmark*> bow_matched;
    for (unsigned i = 0; i < frm[1].num_keypts_; ++i) frm[1].landmarks_[i] = all_lms[i % all_lms.size()];
    const unsigned n_bow = match::bow_tree(0.75f, true).match_keyframes(&frm[0], &frm[1], bow_matched);
    // data::bow_vocabulary::transform as frame::compute_bow calls it: a two-level binary tree whose four words
    // split the descriptors by their first byte's top bits
    bool bow_ok = false;
    {
        uint8_t nd[7][32];
        std::memset(nd, 0, sizeof(nd));
        std::memset(nd[2], 0xFF, 32);                      // children of the root: all zeros / all ones
        nd[4][0] = 0xFF;                                   // leaves under node 1: node 3 all zeros, node 4 with the first byte set
        std::memset(nd[5], 0xFF, 32); nd[5][0] = 0x00;     // leaves under node 2: node 5 with the first byte clear, node 6 all ones
        std::memset(nd[6], 0xFF, 32);
        const uint32_t child_offsets[8] = {0, 2, 4, 6, 6, 6, 6, 6}, children[6] = {1, 2, 3, 4, 5, 6}, word_ids[7] = {0, 0, 0, 0, 1, 2, 3};
        const double weights[7] = {0, 0, 0, 1.0, 2.0, 3.0, 4.0};
        ovs_bow_vocab_view view{};
        view.branching_factor = 2; view.depth_levels = 2; view.weighting = OVS_BOW_TF_IDF; view.scoring = OVS_BOW_L1_NORM; view.num_nodes = 7;
        view.node_descs = &nd[0][0]; view.child_offsets = child_offsets; view.children = children; view.word_ids = word_ids; view.weights = weights;
        data::bow_vocabulary vocab(view);
        std::map<unsigned, double> bow_vec;
        std::map<unsigned, std::vector<unsigned>> feat_vec;
        vocab.transform(frm[0].descriptors_, bow_vec, feat_vec, 1);
        double sum = 0; size_t members = 0; bool ascending = true;
        for (const auto& kv : bow_vec) sum += kv.second;
        for (const auto& kv : feat_vec) {
            members += kv.second.size();
            for (size_t i = 1; i < kv.second.size(); ++i) ascending = ascending && kv.second[i - 1] < kv.second[i];
        }
        std::vector<mock::mat_u8> rows(frm[0].num_keypts_);
        for (unsigned i = 0; i < frm[0].num_keypts_; ++i) { rows[i].create(1, 32); std::memcpy(rows[i].data, frm[0].descriptors_.data + 32 * static_cast<size_t>(i), 32); }
        std::map<unsigned, double> bow_vec2;
        std::map<unsigned, std::vector<unsigned>> feat_vec2;
        vocab.transform(rows, bow_vec2, feat_vec2, 1);
        bow_ok = bow_vec.size() >= 2 && bow_vec.size() <= 4 && std::fabs(sum - 1.0) < 1e-12 && members == frm[0].num_keypts_ && ascending
                 && feat_vec.size() == 2 && feat_vec.count(1) && feat_vec.count(2) && bow_vec2 == bow_vec && feat_vec2 == feat_vec;
    }
    // image preparation: BGR -> gray of a colour copy of frame 0 (equal channels stay), identity
    // rectification, histogram stretch
    mock::mat_u8 bgr;
    bgr.create_raw(h, w, 3, 1);
    for (int i = 0; i < w * h; ++i) for (int ch = 0; ch < 3; ++ch) bgr.data[3 * i + ch] = img[0].data[i];
    util::convert_to_grayscale(bgr, OVS_COLOR_BGR);
    const bool gray_ok = bgr.channels() == 1 && std::memcmp(bgr.data, img[0].data, static_cast<size_t>(w) * h) == 0;
    mock::mat_u8 map_x, map_y;
    map_x.create_raw(h, w, 1, 4);
    map_y.create_raw(h, w, 1, 4);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        reinterpret_cast<float*>(map_x.data)[y * w + x] = static_cast<float>(x);
        reinterpret_cast<float*>(map_y.data)[y * w + x] = static_cast<float>(y);
    }
    mock::mat_u8 rect_l, rect_r;
    {
        util::stereo_rectifier rectifier(map_x, map_y, map_x, map_y);
        rectifier.rectify(img[0], img[1], rect_l, rect_r);
    }
    const bool rect_ok = std::memcmp(rect_l.data, img[0].data, static_cast<size_t>(w) * h) == 0 && std::memcmp(rect_r.data, img[1].data, static_cast<size_t>(w) * h) == 0;
    util::equalize_histogram(rect_l);
    int eq_min = 255, eq_max = 0;
    for (int i = 0; i < w * h; ++i) { eq_min = std::min<int>(eq_min, rect_l.data[i]); eq_max = std::max<int>(eq_max, rect_l.data[i]); }
    const bool pre_ok = gray_ok && rect_ok && eq_min == 0 && eq_max == 255;
    // bearings: the principal point looks along +z, the image centre of a panorama as well
    std::vector<mock::keypoint> at_centre(3);
    at_centre[0].x = 320; at_centre[0].y = 240;
    at_centre[1].x = 820; at_centre[1].y = 240;      // x_n = 1
    at_centre[2].x = 320; at_centre[2].y = 160;      // panorama centre
    std::vector<mock::vec3> bearings;
    camera::convert_keypoints_to_bearings(mock::pinhole{}, at_centre, bearings);
    bool bearings_ok = bearings.size() == 3 && bearings[0](0) == 0.0 && bearings[0](1) == 0.0 && bearings[0](2) == 1.0
                       && bearings[1](0) == 1.0 / std::sqrt(2.0) && bearings[1](2) == 1.0 / std::sqrt(2.0);
    camera::convert_keypoints_to_bearings_equirectangular(mock::sphere{}, at_centre, bearings);
    bearings_ok = bearings_ok && std::fabs(bearings[2](0)) < 1e-15 && std::fabs(bearings[2](1)) < 1e-15 && std::fabs(bearings[2](2) - 1.0) < 1e-15;
    // undistortion: the principal point stays, a point right of it moves outwards under barrel distortion (k1 < 0)
    std::vector<mock::keypoint> undist;
    at_centre[1].angle = 33.0f; at_centre[1].octave = 2; at_centre[1].size = 9.0f; at_centre[1].response = 50.0f;
    camera::undistort_keypoints_perspective(mock::pinhole{}, at_centre, undist);
    bearings_ok = bearings_ok && undist.size() == 3 && undist[0].x == 320.0f && undist[0].y == 240.0f && undist[1].x > 820.0f && undist[1].y == 240.0f
                  && undist[1].angle == 33.0f && undist[1].octave == 2 && undist[1].size == 9.0f && undist[1].response == 0.0f && undist[1].class_id == -1;
    camera::undistort_keypoints_fisheye(mock::pinhole{}, at_centre, undist);
    bearings_ok = bearings_ok && undist.size() == 3 && undist[0].x == 320.0f && undist[0].y == 240.0f && undist[1].x > 820.0f;
    // stereo: frame 1 is frame 0 seen dx pixels to the left, i.e. a right camera image with disparity dx
    feature::orb_extractor ext_r(prm);
    std::vector<mock::keypoint> kps_r;
    mock::mat_u8 desc_r;
    ext.extract(img[0], no_mask, frm[0].keypts_, frm[0].descriptors_); // the left pyramid must be the last one in `ext`
    ext_r.extract(img[1], no_mask, kps_r, desc_r);
    std::vector<float> sxr, sdepth;
    match::stereo(ext, ext_r, frm[0].keypts_, kps_r, frm[0].descriptors_, desc_r, ext.get_scale_factors(), ext.get_inv_scale_factors(),
                  40.0f, 0.5f).compute(sxr, sdepth);
    unsigned n_stereo = 0, n_stereo_ok = 0;
    for (size_t i = 0; i < sxr.size(); ++i) {
        if (sxr[i] >= 0) {
            ++n_stereo;
            n_stereo_ok += std::fabs((frm[0].keypts_[i].x - sxr[i]) - dx) < 1.0f && std::fabs(sdepth[i] - 40.0f / (frm[0].keypts_[i].x - sxr[i])) < 1e-3f;
        }
    }
    bool stereo_ok = n_stereo > 200 && n_stereo_ok * 10 > n_stereo * 9;
    {
        // the reference's own argument list (match/stereo.h:18-22), as data/frame.cc:77-81 writes it: with the
        // extractors' image_pyramid_ members (levels resident on the device) ...
        std::vector<float> sxr2, sdepth2;
        match::stereo(ext.image_pyramid_, ext_r.image_pyramid_, frm[0].keypts_, kps_r, frm[0].descriptors_, desc_r, ext.get_scale_factors(),
                      ext.get_inv_scale_factors(), 40.0f, 0.5f).compute(sxr2, sdepth2);
        stereo_ok = stereo_ok && sxr2 == sxr && sdepth2 == sdepth;
        // ... and with host images: the pyramids travel with the call
        std::vector<mock::mat_u8> pyr_l(ext.get_num_scale_levels()), pyr_r(ext.get_num_scale_levels());
        for (unsigned int lv = 0; lv < pyr_l.size(); ++lv) {
            const auto pl = ext.image_pyramid_level(lv), pr = ext_r.image_pyramid_level(lv);
            pyr_l[lv].create(pl.height, pl.width);
            pyr_r[lv].create(pr.height, pr.width);
            std::memcpy(pyr_l[lv].data, pl.pixels.data(), pl.pixels.size());
            std::memcpy(pyr_r[lv].data, pr.pixels.data(), pr.pixels.size());
        }
        std::vector<float> sxr3, sdepth3;
        match::stereo(pyr_l, pyr_r, frm[0].keypts_, kps_r, frm[0].descriptors_, desc_r, ext.get_scale_factors(), ext.get_inv_scale_factors(),
                      40.0f, 0.5f).compute(sxr3, sdepth3);
        stereo_ok = stereo_ok && sxr3 == sxr && sdepth3 == sdepth;
    }
    // brute force + (stand-in) essential-matrix inlier selection
    std::vector<mock::landmark*> robust_matched;
    const unsigned n_rkf = match::robust(0.75f, true).match_frame_and_keyframe<mock::row_solver>(frm[1], &frm[0], robust_matched);
    // triangulation matches between keypoints without landmarks: frame 1 is frame 0 moved along x
    frm[0].landmarks_.assign(frm[0].num_keypts_, nullptr);
    frm[1].landmarks_.assign(frm[1].num_keypts_, nullptr);
    mock::mat33 E_12;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E_12.m[i][j] = 0.0;
    E_12.m[1][2] = -1.0; E_12.m[2][1] = 1.0;
    frm[0].cam_center_.v[0] = 1.0;
    std::vector<std::pair<unsigned int, unsigned int>> tri_pairs;
    const unsigned n_tri = match::robust(0.75f, true).match_for_triangulation(&frm[0], &frm[1], E_12, tri_pairs);
    frm[0].cam_center_.v[0] = 0.0;
    // initializer-style area matching around the previous positions
    std::vector<mock::point2f> prev_pts;
    for (const auto& kp : frm[0].undist_keypts_) prev_pts.push_back({kp.x, kp.y});
    std::vector<int> area_matched;
    const unsigned n_area = match::area(0.9f, true).match_in_consistent_area(frm[0], frm[1], prev_pts, area_matched, 20);
    unsigned area_moved = 0;
    for (size_t i = 0; i < area_matched.size(); ++i) {
        if (area_matched[i] >= 0) area_moved += prev_pts[i].x == frm[1].undist_keypts_[area_matched[i]].x && prev_pts[i].y == frm[1].undist_keypts_[area_matched[i]].y;
    }
    // landmarks of both keyframes: frame 0's project into frame 1 at x - dx, frame 1's into frame 0 at x + dx
    std::vector<std::unique_ptr<mock::landmark>> lms2;
    for (unsigned i = 0; i < frm[0].num_keypts_; ++i) { frm[0].landmarks_[i] = lms[i].get(); lms[i]->add_observation(&frm[0], static_cast<int>(i)); }
    for (unsigned i = 0; i < frm[1].num_keypts_; ++i) {
        lms2.emplace_back(new mock::landmark);
        lms2.back()->pos.v[0] = frm[1].keypts_[i].x + dx; lms2.back()->pos.v[1] = frm[1].keypts_[i].y; lms2.back()->pos.v[2] = 1.0;
        lms2.back()->desc.create(1, 32);
        std::memcpy(lms2.back()->desc.data, frm[1].descriptors_.data + 32 * static_cast<size_t>(i), 32);
        lms2.back()->level = frm[1].keypts_[i].octave;
        lms2.back()->add_observation(&frm[1], static_cast<int>(i));
        frm[1].landmarks_[i] = lms2.back().get();
    }
    std::vector<mock::landmark*> mutual(frm[0].num_keypts_, nullptr);
    const float s_12 = 1.0f;
    const unsigned n_mutual = match::projection(0.9f, true).match_keyframes_mutually(&frm[0], &frm[1], mutual, s_12, mock::mat33{}, mock::vec3{}, 7.5f);
    unsigned mutual_written = 0;
    for (auto* lm : mutual) mutual_written += lm != nullptr;
    // duplicate detection, then replacement, of frame 0's landmarks inside keyframe 1
    std::vector<mock::landmark*> duplicated;
    const unsigned n_detect = match::fuse(0.6f).detect_duplication(&frm[1], mock::mat44{}, all_lms, 4.0f, duplicated);
    unsigned dup_written = 0;
    for (auto* lm : duplicated) dup_written += lm != nullptr;
    const unsigned n_replace = match::fuse(0.6f).replace_duplication(&frm[1], all_lms, 4.0f);
    unsigned replaced = 0, adopted = 0;
    for (auto& lm : lms2) replaced += lm->erased;
    for (auto& lm : lms) { replaced += lm->erased; adopted += lm->is_observed_in_keyframe(&frm[1]); }
    std::printf("OK keypts %u %u projection %u written %u brute_force %u pairs %zu frame_and_keyframe %u sim3 %u bow_keyframes %u "
                "robust_keyframe %u triangulation %u area %u mutual %u fuse_detect %u fuse_replace %u replaced %u adopted %u preparation %d bearings %d bow %d stereo %u/%u\n",
                frm[0].num_keypts_, frm[1].num_keypts_, n_proj, written, n_bf, pairs.size(), n_kf, n_sim3, n_bow, n_rkf, n_tri, n_area, n_mutual,
                n_detect, n_replace, replaced, adopted, pre_ok ? 1 : 0, bearings_ok ? 1 : 0, bow_ok ? 1 : 0, n_stereo_ok, n_stereo);
    const bool ok = n_proj == written && n_proj > 200 && n_bf == pairs.size() && n_bf > 100 && n_kf > 200 && n_sim3 > 100 && n_bow > 100
                    && n_rkf > 50 && n_rkf <= n_bf && n_tri == tri_pairs.size() && n_tri > 50 && n_area > 50 && area_moved == n_area
                    && n_mutual == mutual_written && n_mutual > 100 && n_detect == dup_written && n_detect > 100 && n_replace == replaced
                    && n_replace > 100 && adopted > 50 && pre_ok && bearings_ok && stereo_ok && bow_ok;
    return ok ? 0 : 2;
}
