/*
 * ovo.h -- CPU ORACLE for the ORB extraction + descriptor matching path.  TEST INFRASTRUCTURE.
 *
 * This is a plain C++ restatement of the reference's CPU algorithm
 * (src/openvslam/feature/orb_extractor.cc, orb_extractor_node.cc, orb_params.cc,
 * util/trigonometric.h, match/{base.h,angle_checker.h,projection.cc,stereo.cc,robust.cc,
 * bow_tree.cc}, data/common.{h,cc}) plus restatements of the OpenCV primitives that path calls
 * (cv::resize INTER_LINEAR 8U, cv::FAST TYPE_9_16 + NMS, cv::GaussianBlur 7x7 sigma 2 8U,
 * cv::fastAtan2, cvRound/cvFloor/cvCeil).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product library never does.
 *
 * PARITY STATUS
 *  - pinned by the reference's own known-answer tests: Hamming distance
 *    (test/openvslam/match/base.cc), angle_checker partitions
 *    (test/openvslam/match/angle_checker.cc), scale tables (test/openvslam/feature/orb_params.cc),
 *    grid cell indices (test/openvslam/data/common_get_cell_indices.cc), cos/sin tolerance
 *    (test/openvslam/util/trigonometric.cc), extractor property tests
 *    (test/openvslam/feature/orb_extractor.cc);
 *  - PARITY UNPINNED for everything that goes through OpenCV (3.3.1 or later, not present in
 *    this build environment, no golden pixels/keypoints/descriptors in the reference's tests):
 *    pyramid pixels, FAST keypoints and scores, blurred pixels, angles, descriptors; and for BoW
 *    node assignment (DBoW2 + vocabulary file, absent).  The OpenCV restatements target the
 *    generic (non-IPP, scalar) code paths of OpenCV 3.3.1 as documented in ovo_cvprim.cc.
 *  - where the reference is itself implementation-defined the oracle fixes one rule:
 *    the quadtree's (count, node address) sort (orb_extractor.cc:452) breaks count ties by node
 *    creation sequence (later created first), which is what monotonically increasing heap
 *    addresses give; angle_checker's std::sort of 30 bin indices (angle_checker.h:129) is kept as
 *    libstdc++'s std::sort.
 *
 * The API mirrors include/ovs.h one to one (ovo_* instead of ovs_*, no device/stream arguments)
 * so that parity tests call both with the same buffers.
 */
#ifndef OVO_H
#define OVO_H

#include "../include/ovs.h"

#ifdef __cplusplus
extern "C" {
#endif

const char* ovo_last_error(void);

/* ---- OpenCV primitive restatements (ovo_cvprim.cc) ---- */
int ovo_cv_round_f(float v);
int ovo_cv_round_d(double v);
int ovo_cv_floor_f(float v);
int ovo_cv_ceil_f(float v);
int ovo_cv_ceil_d(double v);
float ovo_fast_atan2(float y, float x);
/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR) for CV_8UC1 */
int ovo_resize_linear_u8(const uint8_t* src, int sw, int sh, size_t sstride,
                         uint8_t* dst, int dw, int dh, size_t dstride);
/* cv::FAST(img, kps, threshold, true) (TYPE_9_16); returns count, writes up to cap keypoints */
int ovo_fast_9_16(const uint8_t* img, int w, int h, size_t stride, int threshold,
                  ovs_keypoint* kps, int cap);
/* cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1 */
int ovo_gaussian_blur_7x7(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride);
int ovo_gaussian_blur_7x7_variant(const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst, size_t dstride, int kernel_variant);
void ovo_gaussian_kernel_7_q8_normalised(int32_t out[7]);
/* the integer kernel cvRound(getGaussianKernel(7, 2) * 256) */
void ovo_gaussian_kernel_7_q8(int32_t out[7]);

/* ---- util/trigonometric.h ---- */
float ovo_cos(float v);
float ovo_sin(float v);

/* ---- feature/orb_params.cc ---- */
void ovo_calc_scale_factors(uint32_t num_levels, float scale_factor, float* out);
void ovo_calc_inv_scale_factors(uint32_t num_levels, float scale_factor, float* out);
void ovo_calc_level_sigma_sq(uint32_t num_levels, float scale_factor, float* out);
void ovo_calc_inv_level_sigma_sq(uint32_t num_levels, float scale_factor, float* out);

/* ---- feature/orb_extractor.cc ---- */
typedef struct ovo_orb ovo_orb;
int ovo_orb_create(const ovs_orb_params* params, ovo_orb** out);
void ovo_orb_destroy(ovo_orb* h);
int ovo_orb_set_max_num_keypoints(ovo_orb* h, uint32_t v);
int ovo_orb_set_scale_factor(ovo_orb* h, float v);
int ovo_orb_set_num_scale_levels(ovo_orb* h, uint32_t v);
int ovo_orb_set_initial_fast_threshold(ovo_orb* h, uint32_t v);
int ovo_orb_set_minimum_fast_threshold(ovo_orb* h, uint32_t v);
int ovo_orb_set_blur_kernel(ovo_orb* h, int kernel_variant);
int ovo_orb_get_num_keypts_per_level(const ovo_orb* h, uint32_t* out);
int ovo_orb_get_u_max(const ovo_orb* h, int32_t* out16);
int ovo_orb_extract(ovo_orb* h, const uint8_t* img, int width, int height, size_t stride,
                    const uint8_t* mask, size_t mask_stride,
                    ovs_keypoint* kps, uint8_t* desc, int cap, int* n);
/* views into the state left by the last extract (valid until the next call on the handle) */
int ovo_orb_pyramid_level(const ovo_orb* h, int level, const uint8_t** ptr, int* w, int* hh, size_t* stride);
int ovo_orb_blurred_level(const ovo_orb* h, int level, const uint8_t** ptr, int* w, int* hh, size_t* stride);
/* FAST candidates handed to the quadtree at `level` (keypts_to_distribute, orb_extractor.cc:302-369) */
int ovo_orb_candidates(const ovo_orb* h, int level, ovs_keypoint* out, int cap);
/* accumulated seconds per stage since creation: pyramid, fast, quadtree, orientation, blur, descriptors */
int ovo_orb_stage_seconds(const ovo_orb* h, double out[6]);
/* the rectangle mask of orb_extractor.cc:248-261 for a cols x rows image (255 = keep, 0 = masked) */
int ovo_create_rectangle_mask(const ovs_orb_params* params, int cols, int rows, uint8_t* out);

/* ---- match/base.h ---- */
uint32_t ovo_hamming_32(const uint8_t* a, const uint8_t* b);
uint32_t ovo_hamming_64(const uint8_t* a, const uint8_t* b);
int ovo_hamming_pairs(const uint8_t* a, const uint8_t* b, int n, uint32_t* out);
int ovo_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out);

/* ---- match/angle_checker.h (T = int) ---- */
typedef struct ovo_angle_checker ovo_angle_checker;
ovo_angle_checker* ovo_angle_checker_create(uint32_t histogram_length, uint32_t num_bins_thr);
void ovo_angle_checker_destroy(ovo_angle_checker* c);
void ovo_angle_checker_append(ovo_angle_checker* c, float delta_angle, int match);
int ovo_angle_checker_valid(const ovo_angle_checker* c, int* out, int cap);
int ovo_angle_checker_invalid(const ovo_angle_checker* c, int* out, int cap);
int ovo_angle_checker_partition(const float* delta_angles, int n, int num_bins_thr, uint8_t* is_valid);

/* ---- data/common.{h,cc} ---- */
int ovo_get_cell_indices(const ovs_grid* g, float x, float y, int* cell_x, int* cell_y);
int ovo_get_keypoints_in_cell(const ovs_frame_view* frm, int n_queries,
                              const float* ref_xy, const float* margin,
                              const int32_t* min_level, const int32_t* max_level,
                              uint32_t* out_indices, int cap, uint32_t* out_offsets);

/* ---- match/projection.cc ---- */
int ovo_match_current_and_last_frames(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                      const uint8_t* curr_occupied, float margin, int direction,
                                      int last_num_scale_levels, float lowe_ratio, int check_orientation,
                                      int32_t* matched_query, uint32_t* num_matches);
int ovo_match_frame_and_landmarks(const ovs_frame_view* frm, const ovs_proj_queries* q,
                                  const uint8_t* frm_occupied, float margin, float lowe_ratio,
                                  int32_t* matched_query, uint32_t* num_matches);
int ovo_match_frame_and_keyframe(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                 const uint8_t* curr_occupied, float margin, uint32_t hamm_dist_thr,
                                 int check_orientation,
                                 int32_t* matched_query, uint32_t* num_matches);
int ovo_match_by_sim3_transform(const ovs_frame_view* keyfrm, const ovs_proj_queries* q,
                                const uint8_t* keyfrm_occupied, float margin,
                                int32_t* matched_query, uint32_t* num_matches);
int ovo_match_keyframe_one_way(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, float margin,
                               int32_t* best_idx_of_query);
int ovo_cross_check(const int32_t* idx2_of_1, int n1, const int32_t* idx1_of_2, int n2,
                    int32_t* out_idx2_of_1, uint32_t* num_matches);

/* ---- match/stereo.cc ---- */
int ovo_stereo_compute(const ovs_image_view* left_pyr, const ovs_image_view* right_pyr, int num_levels,
                       const ovs_keypoint* keypts_left, int n_left,
                       const ovs_keypoint* keypts_right, int n_right,
                       const uint8_t* descs_left, const uint8_t* descs_right,
                       const float* scale_factors, const float* inv_scale_factors,
                       float focal_x_baseline, float true_baseline,
                       float* stereo_x_right, float* depths);

/* ---- match/robust.cc, match/bow_tree.cc ---- */
int ovo_brute_force_match(const uint8_t* descs_1, const float* angles_1, int n1,
                          const uint8_t* descs_2, const float* angles_2, const uint8_t* valid_2, int n2,
                          float lowe_ratio, int check_orientation,
                          int32_t* matches, uint32_t* num_matches);
int ovo_bow_match_frame_and_keyframe(const ovs_bow_feat_vec* kf_vec, const uint8_t* kf_descs,
                                     const float* kf_angles, const uint8_t* valid_kf, int n_kf,
                                     const ovs_bow_feat_vec* frm_vec, const uint8_t* frm_descs,
                                     const float* frm_angles, int n_frm,
                                     float lowe_ratio, int check_orientation,
                                     int32_t* matched_kf_idx_in_frm, uint32_t* num_matches);
int ovo_bow_match_keyframes(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1, const float* angles_1,
                            const uint8_t* valid_1, int n1,
                            const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2, const float* angles_2,
                            const uint8_t* valid_2, int n2,
                            float lowe_ratio, int check_orientation,
                            int32_t* matched_idx2_in_kf1, uint32_t* num_matches);
/* frame::compute_bow (src/openvslam/data/frame.cc:162-170): DBoW2 TemplatedVocabulary::transform(features, v, fv,
 * levelsup), restated from the published algorithm (DBoW2 is not in the reference tree: parity unpinned) */
int ovo_bow_transform(const ovs_bow_vocab_view* voc, const uint8_t* descs, int n, int levelsup, uint32_t* word_of_feature,
                      uint32_t* node_of_feature, uint32_t* bow_word_ids, double* bow_values, int32_t* num_words,
                      uint32_t* fv_node_ids, uint32_t* fv_offsets, uint32_t* fv_indices, int32_t* fv_num_nodes);
/* candidate search of fuse::detect_duplication / replace_duplication (src/openvslam/match/fuse.cc) */
int ovo_fuse_best_candidates(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, const double* reproj_xy_d,
                             const float* inv_level_sigma_sq, float margin, int replace_mode, int32_t* best_idx_of_query);

/* util::convert_to_grayscale / equalize_histogram (src/openvslam/util/image_converter.cc:8-61) and one
 * cv::remap of util::stereo_rectifier::rectify (src/openvslam/util/stereo_rectifier.cc:56-60) */
int ovo_convert_to_grayscale(const uint8_t* src, int width, int height, int channels, size_t src_stride, int color_order,
                             uint8_t* dst, size_t dst_stride);
int ovo_equalize_histogram(const void* src, int width, int height, size_t src_stride, int bits, uint8_t* dst, size_t dst_stride);
int ovo_remap_linear(const uint8_t* src, int src_width, int src_height, size_t src_stride, const float* map_x, const float* map_y,
                     int width, int height, uint8_t* dst, size_t dst_stride);
void ovo_remap_linear_table(int16_t* out);

/* data::landmark::compute_descriptor (src/openvslam/data/landmark.cc:127-184), batched */
int ovo_landmark_compute_descriptors(const uint8_t* descs, const uint32_t* offsets, int n_landmarks, int32_t* best_idx);

/* match::area::match_in_consistent_area (src/openvslam/match/area.cc:8-112) */
int ovo_match_in_consistent_area(const ovs_keypoint* undist_keypts_1, const uint8_t* descs_1, int n1,
                                 const ovs_frame_view* frm_2, float* prev_matched_pts, int margin,
                                 float lowe_ratio, int check_orientation,
                                 int32_t* matched_indices_2_in_frm_1, uint32_t* num_matches);

int ovo_match_for_triangulation(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1,
                                const ovs_keypoint* undist_keypts_1, const double* bearings_1,
                                const float* stereo_x_right_1, const uint8_t* has_lm_1, int n1,
                                const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2,
                                const ovs_keypoint* undist_keypts_2, const double* bearings_2,
                                const float* stereo_x_right_2, const uint8_t* has_lm_2, int n2,
                                const double* E_12, const double* epipole_in_2,
                                const float* scale_factors_1, int num_scale_levels_1,
                                int check_orientation,
                                uint32_t* pairs, uint32_t* num_matches);

/* ---- database top-2 search (serial strict-< scan, robust.cc:250-268 without the exclusion) ---- */
int ovo_knn2_search(const uint8_t* db, int64_t n_rows, int64_t global_row_offset,
                    const uint8_t* queries, int nq, ovs_knn2* out);
int ovo_knn2_merge(const ovs_knn2* lists, int n_lists, int nq, ovs_knn2* out);

/* ---- batched projection matching of consecutive frames (same contract as
 *      ovs_match_batch_current_and_last_dev, host buffers) ---- */
int ovo_match_batch_current_and_last(const ovs_keypoint* kps, const uint8_t* desc,
                                     const int32_t* counts, int n_frames, int cap,
                                     const float* reproj_xy, const ovs_grid* grid,
                                     const float* scale_factors, int num_scale_levels,
                                     float margin, int check_orientation,
                                     int32_t* matched, uint32_t* num_matches);

/* camera::*::convert_keypoints_to_bearings (camera/perspective.cc:130-138, fisheye.cc:169-177,
 * equirectangular.cc:45-56) */
int ovo_convert_keypoints_to_bearings(const ovs_camera* cam, const ovs_keypoint* undist_keypts, int n, double* bearings);
/* camera::*::undistort_keypoints (camera/perspective.cc:99-128, fisheye.cc:139-167, equirectangular.cc:41-43) */
int ovo_undistort_keypoints(const ovs_camera* cam, const ovs_keypoint* dist_keypts, int n, ovs_keypoint* undist_keypts);

#ifdef __cplusplus
}
#endif
#endif /* OVO_H */
