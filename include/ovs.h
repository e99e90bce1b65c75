/*
 * ovs.h -- C ABI of the MI355X (gfx950) ORB extraction + binary-descriptor matching library.
 *
 * This is the drop-in boundary for the front-end hot path of OpenVSLAM
 * (feature::orb_extractor::extract() and match::{projection,stereo,robust,bow_tree}).
 * The reference has no FFI layer for this path: it is ordinary C++ classes
 * (src/openvslam/feature/orb_extractor.h:13-150 and projection.h, stereo.h, robust.h,
 * bow_tree.h under src/openvslam/match), so every entry point below cites the class member it
 * replaces.  A header-only C++ shim that restores the reference's class names on top of these
 * functions lives in openvslam_amd/cpp/.
 *
 * Conventions
 *  - plain pointers and sizes only; nothing here depends on torch, OpenCV or Eigen;
 *  - every function returns an ovs_status (0 = ok, <0 = error); the message of the last error
 *    of the calling thread is available from ovs_last_error();
 *  - "host" entry points take host pointers and copy; "_dev" entry points take device (HBM)
 *    pointers and enqueue work on the given hipStream_t (passed as void*), without syncing;
 *  - the library never keeps a caller pointer after the call returns (the reference deep-copies
 *    descriptors the same way, src/openvslam/data/keyframe.cc:28);
 *  - an extractor handle is stateful and not re-entrant, exactly like the reference object
 *    (image_pyramid_, rect_mask_: src/openvslam/feature/orb_extractor.h:75,136-137); distinct
 *    handles may be used from distinct threads (left/right extraction,
 *    src/openvslam/data/frame.cc:64-67).  Matcher entry points keep no global state.
 */
#ifndef OVS_H
#define OVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum ovs_status {
    OVS_OK = 0,
    OVS_ERR_INVALID_ARGUMENT = -1, /* std::runtime_error / assert in the reference           */
    OVS_ERR_CAPACITY = -2,         /* caller buffer too small for the result                  */
    OVS_ERR_HIP = -3,              /* a HIP runtime call failed                               */
    OVS_ERR_UNSUPPORTED = -4,      /* a configuration the device path does not implement      */
    OVS_ERR_OUT_OF_RANGE = -5      /* std::out_of_range from .at() in the reference           */
} ovs_status;

const char* ovs_last_error(void);
/* Number of visible HIP devices (<=0 if none).  Never falls back to a CPU path. */
int ovs_device_count(void);

/* Binary layout of cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id.
 * extract() fills them as src/openvslam/feature/orb_extractor.cc:380-389,594,598-606 do. */
typedef struct ovs_keypoint {
    float x, y;
    float size;
    float angle;
    float response;
    int32_t octave;
    int32_t class_id;
} ovs_keypoint;

/* ------------------------------------------------------------------------------------------ */
/* ORB extractor                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* feature::orb_params (src/openvslam/feature/orb_params.h, orb_params.cc:8-33). */
typedef struct ovs_orb_params {
    uint32_t max_num_keypts;
    float scale_factor;
    uint32_t num_levels;
    uint32_t ini_fast_thr;
    uint32_t min_fast_thr;
    uint32_t num_mask_rects;
    const float* mask_rects; /* num_mask_rects x {x_min, x_max, y_min, y_max} as image ratios */
} ovs_orb_params;

/* Fills the defaults of orb_params(const YAML::Node&): 2000 / 1.2 / 8 / 20 / 7 / no rects
 * (src/openvslam/feature/orb_params.cc:27-33). */
void ovs_orb_params_default(ovs_orb_params* p);

/* Static scale tables, orb_params::calc_* (src/openvslam/feature/orb_params.cc:53-87). */
void ovs_calc_scale_factors(uint32_t num_levels, float scale_factor, float* out);
void ovs_calc_inv_scale_factors(uint32_t num_levels, float scale_factor, float* out);
void ovs_calc_level_sigma_sq(uint32_t num_levels, float scale_factor, float* out);
void ovs_calc_inv_level_sigma_sq(uint32_t num_levels, float scale_factor, float* out);

typedef struct ovs_orb ovs_orb;

/* orb_extractor::orb_extractor(const orb_params&) (orb_extractor.cc:62-66).  Mask-rectangle
 * validation errors of orb_params.cc:14-24 come back as OVS_ERR_INVALID_ARGUMENT with the
 * reference's message.  `device` is the HIP device ordinal the handle lives on.
 * Limits (OVS_ERR_UNSUPPORTED from the first extract call, which plans for the image size): 16 levels,
 * 4134 pixels per side, and 8187 keypoints per pyramid level (about 37 000 keypoints at eight levels and
 * scale 1.2; levels with more than about 1130 keep their quadtree node records in a device workspace of
 * the handle instead of LDS: 72 bytes x (2 * quota + 8) per such level and frame of a batch). */
int ovs_orb_create(const ovs_orb_params* params, int device, ovs_orb** out);
void ovs_orb_destroy(ovs_orb* h);

/* The coefficients of cv::GaussianBlur(7 x 7, sigma 2) on 8-bit images (orb_extractor.cc:133) depend on the
 * OpenCV the reference is built against (CMakeLists.txt:82-92 accepts 3.3.1 or newer; its Dockerfile builds
 * 4.1.0).  Both forms are the exact integer sum  (sum_j sum_i k[j] k[i] p[y+j][x+i] + 2^15) >> 16, saturated:
 *   OVS_BLUR_ROUNDED     k = round(256 g) = {18, 34, 49, 55, 49, 34, 18} (sum 257): OpenCV 3.3.1 (integer
 *                        separable filter) up to 4.1.x (Q8.8 fixed point with the same rounded coefficients,
 *                        which computes the same integers).  The default.
 *   OVS_BLUR_NORMALISED  k = {18, 34, 48, 56, 48, 34, 18} (sum 256, rounding error diffused into the next
 *                        coefficient, the centre takes the rest): OpenCV 3.4.9 / 4.2.0 and newer.
 * Which release changed what is restated from the OpenCV sources from memory (they are not available to this
 * build): a caller pins the choice by blurring one image with its own OpenCV. */
enum { OVS_BLUR_ROUNDED = 0, OVS_BLUR_NORMALISED = 1 };
int ovs_orb_set_blur_kernel(ovs_orb* h, int kernel_variant);
int ovs_orb_get_blur_kernel(const ovs_orb* h);

/* Getters / setters of orb_extractor.cc:146-203.  As in the reference, changing keypoints /
 * scale factor / levels re-runs initialize(); changing thresholds does not. */
uint32_t ovs_orb_get_max_num_keypoints(const ovs_orb* h);
int ovs_orb_set_max_num_keypoints(ovs_orb* h, uint32_t v);
float ovs_orb_get_scale_factor(const ovs_orb* h);
int ovs_orb_set_scale_factor(ovs_orb* h, float v);
uint32_t ovs_orb_get_num_scale_levels(const ovs_orb* h);
int ovs_orb_set_num_scale_levels(ovs_orb* h, uint32_t v);
uint32_t ovs_orb_get_initial_fast_threshold(const ovs_orb* h);
int ovs_orb_set_initial_fast_threshold(ovs_orb* h, uint32_t v);
uint32_t ovs_orb_get_minimum_fast_threshold(const ovs_orb* h);
int ovs_orb_set_minimum_fast_threshold(ovs_orb* h, uint32_t v);
int ovs_orb_get_scale_factors(const ovs_orb* h, float* out);
int ovs_orb_get_inv_scale_factors(const ovs_orb* h, float* out);
int ovs_orb_get_level_sigma_sq(const ovs_orb* h, float* out);
int ovs_orb_get_inv_level_sigma_sq(const ovs_orb* h, float* out);
/* num_keypts_per_level_ of orb_extractor.cc:214-223 (num_levels values). */
int ovs_orb_get_num_keypts_per_level(const ovs_orb* h, uint32_t* out);

/* Upper bound on the number of keypoints extract() can return for a width x height image.
 * The quadtree of orb_extractor.cc:398-475 may exceed the per-level quota (a sweep stops once
 * `quota <= nodes.size()`), so this is larger than max_num_keypts; size output buffers with it. */
int ovs_orb_keypoint_capacity(const ovs_orb* h, int width, int height);

/* orb_extractor::extract() (orb_extractor.cc:68-144), one frame, host buffers.
 *   img          CV_8UC1 image, `stride` bytes per row; width==0 or height==0 or img==NULL is
 *                the reference's "empty image": returns OVS_OK without touching the outputs.
 *   mask         optional CV_8UC1 image mask of the same size (0 = masked); NULL selects the
 *                rectangle mask of the params if any, else no mask (orb_extractor.cc:90-104).
 *   kps, desc    caller buffers of `cap` keypoints / cap x 32 bytes; *n receives the count.
 * Synchronous. */
int ovs_orb_extract(ovs_orb* h, const uint8_t* img, int width, int height, size_t stride,
                    const uint8_t* mask, size_t mask_stride,
                    ovs_keypoint* kps, uint8_t* desc, int cap, int* n);

/* The same for a batch of equally sized frames already resident in HBM (offline batches).
 *   d_imgs       n_frames images, row_stride bytes per row, frame_stride bytes per frame
 *   d_mask       optional level-0 mask shared by all frames (NULL: rectangle mask / none)
 *   d_kps        n_frames x cap keypoints; d_desc n_frames x cap x 32 bytes; d_counts n_frames
 *   stream       hipStream_t; work is enqueued, not synchronised.  The blur runs on a second stream
 *                owned by the handle, forked from `stream` after the pyramid and joined before the
 *                descriptors with events (legal under stream capture); everything the call enqueues
 *                is complete when `stream` reaches the point after the call.  OVS_SERIAL_EXTRACT=1 in
 *                the environment keeps all kernels on `stream`.
 * A frame whose keypoint count exceeds `cap` gets d_counts[f] = -(count) and truncated output. */
int ovs_orb_extract_batch_dev(ovs_orb* h, const uint8_t* d_imgs, int n_frames, int width, int height,
                              size_t row_stride, size_t frame_stride,
                              const uint8_t* d_mask, size_t mask_stride,
                              ovs_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts, int cap,
                              void* stream);

/* image_pyramid_ (orb_extractor.h:75) of the last extract call, read by match::stereo
 * (src/openvslam/data/frame.cc:77).  Level 0 is the caller's image and is not retained by the
 * library; levels >= 1 stay resident on the device until the next extract on this handle. */
int ovs_orb_pyramid_level_dev(ovs_orb* h, int frame, int level,
                              const uint8_t** d_ptr, int* width, int* height, size_t* stride);
/* Host copy of one pyramid level (levels >= 1) into `out` with `out_stride` bytes per row. */
int ovs_orb_download_pyramid_level(ovs_orb* h, int frame, int level, uint8_t* out, size_t out_stride);
/* Host copy of the Gaussian-blurred level used for the descriptors (orb_extractor.cc:132-133);
 * a debugging / parity hook, the reference keeps this image local. */
int ovs_orb_download_blurred_level(ovs_orb* h, int frame, int level, uint8_t* out, size_t out_stride);
/* Size of pyramid level `level` for a width x height input (orb_extractor.cc:267-268). */
int ovs_orb_level_size(const ovs_orb* h, int width, int height, int level, int* w, int* hh);

/* Per-kernel device timing of the extract/match launches of this thread: when enabled every
 * launch is bracketed by hipEvents on its stream; query returns accumulated milliseconds and the
 * launch count for a kernel name since the last reset.  Diagnostics only. */
void ovs_timing_enable(int on);
void ovs_timing_reset(void);
int ovs_timing_query(const char* kernel_name, double* total_ms, int* launches);
/* Writes up to `cap` names (each <= 63 chars + NUL, 64-byte slots) and returns the count. */
int ovs_timing_names(char* names, int cap);

/* ------------------------------------------------------------------------------------------ */
/* Descriptor distance                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* match::compute_descriptor_distance_32 (src/openvslam/match/base.h:18-39) for n pairs of
 * 32-byte descriptors: out[i] = hamming(a[i], b[i]).  Host buffers, computed on the device. */
int ovs_hamming_pairs(const uint8_t* a, const uint8_t* b, int n, uint32_t* out, int device);
/* All-pairs distance matrix out[i*nb + j] = hamming(a[i], b[j]) (host buffers). */
int ovs_hamming_matrix(const uint8_t* a, int na, const uint8_t* b, int nb, uint16_t* out, int device);

/* ------------------------------------------------------------------------------------------ */
/* Frame views used by the matchers                                                           */
/* ------------------------------------------------------------------------------------------ */

/* Grid geometry of camera::base (src/openvslam/camera/base.h:117-130,
 * src/openvslam/camera/perspective.cc:27-30): image bounds and inverse cell sizes. */
typedef struct ovs_grid {
    float min_x, min_y, max_x, max_y;
    double inv_cell_width, inv_cell_height;
    uint32_t num_cols, num_rows; /* 64 x 48 in the reference (camera/base.cc) */
} ovs_grid;

/* The fields of data::frame / data::keyframe the matchers read, flattened by the caller
 * (landmark getters take mutexes, src/openvslam/data/landmark.cc:30-33, and stay on the host). */
typedef struct ovs_frame_view {
    int32_t num_keypts;
    const ovs_keypoint* undist_keypts; /* undist_keypts_: pt, octave, angle are read        */
    const uint8_t* descriptors;        /* num_keypts x 32                                    */
    const float* stereo_x_right;       /* num_keypts, or NULL meaning all -1 (monocular)     */
    const float* scale_factors;        /* num_scale_levels                                   */
    int32_t num_scale_levels;
    ovs_grid grid;
} ovs_frame_view;

/* data::get_cell_indices (src/openvslam/data/common.h:59-64): returns 1 if inside the grid. */
int ovs_get_cell_indices(const ovs_grid* g, float x, float y, int* cell_x, int* cell_y);

/* data::assign_keypoints_to_grid + data::get_keypoints_in_cell
 * (src/openvslam/data/common.cc:97-186) for n_queries windows against one frame: query q
 * writes its indices (reference order: cell-x major, cell-y, insertion order) to
 * out_indices[out_offsets[q] .. out_offsets[q+1]).  out_offsets has n_queries+1 entries.
 * Host buffers, computed on the device. */
int ovs_get_keypoints_in_cell(const ovs_frame_view* frm, int n_queries,
                              const float* ref_xy, const float* margin,
                              const int32_t* min_level, const int32_t* max_level,
                              uint32_t* out_indices, int cap, uint32_t* out_offsets, int device);

/* ------------------------------------------------------------------------------------------ */
/* match::projection                                                                          */
/* ------------------------------------------------------------------------------------------ */

/* One entry per 3-D point projected into the frame being searched.  The projection itself
 * (camera_->reproject_to_image, double precision, src/openvslam/match/projection.cc:127) is
 * done by the caller; these are its outputs plus the landmark fields the loop body reads. */
typedef struct ovs_proj_queries {
    int32_t n;
    const float* reproj_xy;         /* n x 2: reproj(0), reproj(1)                             */
    const float* x_right;           /* n: x_right of the reprojection (NULL: unused)           */
    const int32_t* scale_level;     /* n: last_frm octave / predicted scale level              */
    const float* angle;             /* n: undistorted keypoint angle of the source (or NULL)   */
    const uint8_t* desc;            /* n x 32: lm->get_descriptor()                            */
    const uint8_t* valid;           /* n: 0 = the reference `continue`s before the search      */
    const uint8_t* has_observation; /* n: lm->has_observation() (NULL: all 1)                  */
} ovs_proj_queries;

enum { OVS_DIR_NONE = 0, OVS_DIR_FORWARD = 1, OVS_DIR_BACKWARD = 2 };

/* projection::match_current_and_last_frames (src/openvslam/match/projection.cc:84-209).
 *   curr_occupied   n_curr bytes, 1 where curr.landmarks_[i] && has_observation() on entry
 *                   (projection.cc:164); NULL = none
 *   direction       assume_forward / assume_backward / neither (projection.cc:101-107,139-153)
 *   matched_query   out, n_curr: index of the query whose landmark ends up in
 *                   curr.landmarks_[i] through this call, or -1
 * *num_matches is the function's return value. */
int ovs_match_current_and_last_frames(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                      const uint8_t* curr_occupied, float margin, int direction,
                                      int last_num_scale_levels, float lowe_ratio, int check_orientation,
                                      int32_t* matched_query, uint32_t* num_matches, int device);

/* projection::match_frame_and_landmarks (projection.cc:11-82).  Queries = local landmarks with
 * reproj_in_tracking_, x_right_in_tracking_, scale_level_in_tracking_; `valid` folds
 * is_observable_in_tracking_ && !will_be_erased().  Levels searched: [pred-1, pred]. */
int ovs_match_frame_and_landmarks(const ovs_frame_view* frm, const ovs_proj_queries* q,
                                  const uint8_t* frm_occupied, float margin, float lowe_ratio,
                                  int32_t* matched_query, uint32_t* num_matches, int device);

/* projection::match_frame_and_keyframe (projection.cc:211-313): queries = keyframe landmarks
 * that pass the visibility / distance gates (folded into `valid`), levels [pred-1, pred+1],
 * candidates skipped when curr.landmarks_[i] is non-null (`curr_occupied`), threshold
 * hamm_dist_thr, rotation check against the keyframe keypoint angles. */
int ovs_match_frame_and_keyframe(const ovs_frame_view* curr, const ovs_proj_queries* q,
                                 const uint8_t* curr_occupied, float margin, uint32_t hamm_dist_thr,
                                 int check_orientation,
                                 int32_t* matched_query, uint32_t* num_matches, int device);

/* projection::match_by_Sim3_transform (projection.cc:315-413): window without level gate in
 * get_keypoints_in_cell, then the unsigned `scale_level < pred - 1 || pred < scale_level` gate
 * of projection.cc:390 (pred == 0 rejects every candidate), threshold HAMMING_DIST_THR_LOW.
 * keyfrm_occupied = matched_lms_in_keyfrm[i] != nullptr on entry. */
int ovs_match_by_sim3_transform(const ovs_frame_view* keyfrm, const ovs_proj_queries* q,
                                const uint8_t* keyfrm_occupied, float margin,
                                int32_t* matched_query, uint32_t* num_matches, int device);

/* One direction of projection::match_keyframes_mutually (projection.cc:457-533 / 540-616):
 * per query the best keypoint index (<= HAMMING_DIST_THR_HIGH, same unsigned level gate, no
 * exclusion between queries) or -1.  The cross-check of projection.cc:619-631 is
 * ovs_cross_check below. */
int ovs_match_keyframe_one_way(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, float margin,
                               int32_t* best_idx_of_query, int device);
/* projection.cc:619-631: keeps i -> idx2 only if idx2 -> i.  out_idx2_of_1[i] = idx2 or -1. */
int ovs_cross_check(const int32_t* idx2_of_1, int n1, const int32_t* idx1_of_2, int n2,
                    int32_t* out_idx2_of_1, uint32_t* num_matches);

/* match::angle_checker<T>(30, num_bins_thr) (src/openvslam/match/angle_checker.h:60-134) on its own:
 * append_delta_angle for each of the n delta angles (degrees), then is_valid[i] = 1 when entry i is
 * among get_valid_matches() (its bin is one of the num_bins_thr fullest of the 30 bins, equally full
 * bins ordered as std::sort of libstdc++ orders them), 0 when among get_invalid_matches().  Every
 * matcher above runs this check inside its own kernels; this entry point exposes the class itself.
 * A delta angle outside (-360, 720) gives OVS_ERR_OUT_OF_RANGE (angle_histogram_.at() throws). */
int ovs_angle_checker_partition(const float* delta_angles, int n, int num_bins_thr, uint8_t* is_valid, int device);

/* ------------------------------------------------------------------------------------------ */
/* match::stereo                                                                              */
/* ------------------------------------------------------------------------------------------ */

typedef struct ovs_image_view {
    const uint8_t* data;
    int32_t width, height;
    size_t stride;
} ovs_image_view;

/* match::stereo::compute (src/openvslam/match/stereo.cc:18-113) including
 * get_right_keypoint_indices_in_each_row (:115-142), find_closest_keypoints_in_stereo
 * (:144-177), compute_subpixel_disparity (:179-250) and the median rejection (:96-112).
 *   left_pyr / right_pyr   num_levels host images each (image_pyramid_ of the two extractors)
 *   keypts / descs         distorted keypoints (keypts_, not undistorted) and descriptors
 *   stereo_x_right, depths out, n_left floats each, -1 = no match. */
int ovs_stereo_compute(const ovs_image_view* left_pyr, const ovs_image_view* right_pyr, int num_levels,
                       const ovs_keypoint* keypts_left, int n_left,
                       const ovs_keypoint* keypts_right, int n_right,
                       const uint8_t* descs_left, const uint8_t* descs_right,
                       const float* scale_factors, const float* inv_scale_factors,
                       float focal_x_baseline, float true_baseline,
                       float* stereo_x_right, float* depths, int device);

/* The same on device-resident data: the two pyramids are the ones the extractor handles keep
 * after ovs_orb_extract_batch_dev (frame indices select the batch entries), keypoints / descriptors
 * / counts are the device outputs of those calls.  Enqueued on `stream`, not synchronised.
 * *d_status (device int32, may be NULL) becomes an OVS_ERR_* code when a keypoint's row, octave or
 * patch falls outside the image (where the reference's cv::Mat range checks throw), else 0;
 * with NULL such keypoints are skipped silently. */
int ovs_stereo_compute_dev(ovs_orb* left, int frame_left, ovs_orb* right, int frame_right,
                           const ovs_keypoint* d_kps_left, const uint8_t* d_desc_left, const int32_t* d_n_left,
                           const ovs_keypoint* d_kps_right, const uint8_t* d_desc_right, const int32_t* d_n_right, int cap,
                           float focal_x_baseline, float true_baseline,
                           float* d_stereo_x_right, float* d_depths, int32_t* d_status, int device, void* stream);

/* stereo::compute for the live path: the two pyramids are the ones the extractor handles still
 * hold on the device after their last ovs_orb_extract (data/frame.cc:64-77 runs the two
 * extractions and then match::stereo on their image_pyramid_), so only keypoints and descriptors
 * travel to the device and two float arrays come back.  Host buffers, synchronous. */
int ovs_stereo_compute_resident(ovs_orb* left, ovs_orb* right,
                                const ovs_keypoint* keypts_left, int n_left,
                                const ovs_keypoint* keypts_right, int n_right,
                                const uint8_t* descs_left, const uint8_t* descs_right,
                                float focal_x_baseline, float true_baseline,
                                float* stereo_x_right, float* depths, int device);

/* stereo::compute for a batch of pairs in one set of launches: pair p is frame p of the last
 * ovs_orb_extract_batch_dev on `left` and on `right` (both with >= n_pairs frames), keypoint /
 * descriptor / count arrays in that call's layout (n_pairs x cap, counts per frame), outputs
 * n_pairs x cap floats.  *d_status (device int32) becomes an OVS_ERR_* code when a keypoint's row,
 * octave or patch falls outside the image (where the reference throws), else 0.  Enqueued on
 * `stream`, not synchronised; n_pairs <= 65535. */
int ovs_stereo_compute_batch_dev(ovs_orb* left, ovs_orb* right, int n_pairs,
                                 const ovs_keypoint* d_kps_left, const uint8_t* d_desc_left, const int32_t* d_counts_left,
                                 const ovs_keypoint* d_kps_right, const uint8_t* d_desc_right, const int32_t* d_counts_right, int cap,
                                 float focal_x_baseline, float true_baseline,
                                 float* d_stereo_x_right, float* d_depths, int32_t* d_status, int device, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* match::robust, match::bow_tree                                                             */
/* ------------------------------------------------------------------------------------------ */

/* robust::brute_force_match (src/openvslam/match/robust.cc:209-315).
 *   descs_1/angles_1  frame keypoints (n1); descs_2/angles_2 keyframe keypoints (n2)
 *   valid_2           n2 bytes: lms_2[i] && !will_be_erased() (robust.cc:234-240)
 *   matches           out, pairs (idx_1, idx_2) ascending in idx_1, capacity n1 pairs
 * *num_matches is the function's return value (== number of pairs). */
int ovs_brute_force_match(const uint8_t* descs_1, const float* angles_1, int n1,
                          const uint8_t* descs_2, const float* angles_2, const uint8_t* valid_2, int n2,
                          float lowe_ratio, int check_orientation,
                          int32_t* matches, uint32_t* num_matches, int device);

/* BoW feature vectors (DBoW2::FeatureVector = std::map<node id, std::vector<keypoint idx>>,
 * src/openvslam/match/bow_tree.cc:25-35) flattened in ascending node-id order:
 * node n holds indices[offsets[n] .. offsets[n+1]). */
typedef struct ovs_bow_feat_vec {
    int32_t num_nodes;
    const uint32_t* node_ids; /* ascending */
    const uint32_t* offsets;  /* num_nodes + 1 */
    const uint32_t* indices;
} ovs_bow_feat_vec;

/* bow_tree::match_frame_and_keyframe (src/openvslam/match/bow_tree.cc:16-122).
 *   valid_kf   keyfrm_lms[i] && !will_be_erased();  matched_kf_idx_in_frm out, n_frm: keyframe
 *   keypoint index whose landmark lands in matched_lms_in_frm[i], or -1. */
int ovs_bow_match_frame_and_keyframe(const ovs_bow_feat_vec* kf_vec, const uint8_t* kf_descs,
                                     const float* kf_angles, const uint8_t* valid_kf, int n_kf,
                                     const ovs_bow_feat_vec* frm_vec, const uint8_t* frm_descs,
                                     const float* frm_angles, int n_frm,
                                     float lowe_ratio, int check_orientation,
                                     int32_t* matched_kf_idx_in_frm, uint32_t* num_matches, int device);

/* bow_tree::match_keyframes (bow_tree.cc:124-246): matched_idx2_in_kf1 out, n1: index in
 * keyframe 2 whose landmark lands in matched_lms_in_keyfrm_1[i], or -1. */
int ovs_bow_match_keyframes(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1, const float* angles_1,
                            const uint8_t* valid_1, int n1,
                            const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2, const float* angles_2,
                            const uint8_t* valid_2, int n2,
                            float lowe_ratio, int check_orientation,
                            int32_t* matched_idx2_in_kf1, uint32_t* num_matches, int device);

/* robust::match_for_triangulation (src/openvslam/match/robust.cc:18-172).  Distances, the epipole
 * gate (robust.cc:106-116), the epipolar residual gate (robust.cc:317-335) and the sequential
 * "already matched" rule run on the device, the gates in double.  The residual gate compares
 * cosines (asin(c) < T  <=>  c < sin T); a pair whose cosine lies within 1e-13 of the threshold is
 * re-evaluated on the host with the reference's expression (acos of the host libm) and, should the
 * two disagree, the match is repeated with that decision, so the result is the reference's.
 * Every keypoint index appears in at most one node of its feature vector (as in DBoW2).
 *   bearings_1/2     n x 3 doubles; E_12 row-major 3x3; epipole_in_2 = reproject_to_bearing of
 *                    camera centre 1 into keyframe 2 (robust.cc:25-29)
 *   has_lm_1/2       assoc landmark non-null (those keypoints are skipped, robust.cc:64,83)
 *   stereo_x_right_1/2  may be NULL (monocular: all < 0)
 *   pairs            out, (idx_1, idx_2) ascending idx_1, capacity n1 pairs. */
int ovs_match_for_triangulation(const ovs_bow_feat_vec* vec_1, const uint8_t* descs_1,
                                const ovs_keypoint* undist_keypts_1, const double* bearings_1,
                                const float* stereo_x_right_1, const uint8_t* has_lm_1, int n1,
                                const ovs_bow_feat_vec* vec_2, const uint8_t* descs_2,
                                const ovs_keypoint* undist_keypts_2, const double* bearings_2,
                                const float* stereo_x_right_2, const uint8_t* has_lm_2, int n2,
                                const double* E_12, const double* epipole_in_2,
                                const float* scale_factors_1, int num_scale_levels_1,
                                int check_orientation,
                                uint32_t* pairs, uint32_t* num_matches, int device);

/* ------------------------------------------------------------------------------------------ */
/* BoW transform: descriptors -> BowVector + FeatureVector                                     */
/* ------------------------------------------------------------------------------------------ */

/* frame::compute_bow / keyframe::compute_bow (src/openvslam/data/frame.cc:162-170,
 * data/keyframe.cc:181-189): bow_vocab_->transform(to_desc_vec(descriptors_), bow_vec_, bow_feat_vec_, 4)
 * with the reference's default vocabulary type, DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>
 * (data/bow_vocabulary.h:16).  DBoW2 itself (shinsumicco/DBoW2@687fcb7, Dockerfile.desktop:144) and the
 * vocabulary file are not part of the reference tree: the vocabulary comes in as arrays (parsing the
 * file stays with the caller), and the transform follows DBoW2's published algorithm:
 *   per descriptor, from the root: among the children of the current node, in their stored order, the one
 *   with the smallest Hamming distance wins (strict <: the first of equally distant children); repeat
 *   until a node without children; its word id and weight are the result.  The node passed on the way at
 *   level L - levelsup (root = level 0) is the descriptor's feature-vector node (the root when
 *   L - levelsup <= 0; the leaf itself when the leaf lies above that level, where DBoW2 leaves the id
 *   unset).  Words of weight <= 0 are dropped ("stopped").
 *   BowVector: std::map<word id, value>; TF / TF_IDF add the weight once per descriptor in descriptor
 *   order (value = ((w + w) + w) ..., in double), IDF / BINARY store it once; then the L1 / L2
 *   normalisation of the scoring type, summed in ascending word order, or for DOT_PRODUCT with TF /
 *   TF_IDF a division by the number of words.
 *   FeatureVector: std::map<node id, std::vector<descriptor index>>, indices ascending.
 * Values are IEEE doubles produced by the same operations in the same order as DBoW2's loops. */
enum {
    OVS_BOW_TF_IDF = 0, OVS_BOW_TF = 1, OVS_BOW_IDF = 2, OVS_BOW_BINARY = 3                  /* DBoW2::WeightingType */
};
enum {
    OVS_BOW_L1_NORM = 0, OVS_BOW_L2_NORM = 1, OVS_BOW_CHI_SQUARE = 2, OVS_BOW_KL = 3,
    OVS_BOW_BHATTACHARYYA = 4, OVS_BOW_DOT_PRODUCT = 5                                       /* DBoW2::ScoringType */
};

/* A DBoW2 vocabulary tree as arrays (m_nodes of TemplatedVocabulary): node 0 is the root; the children of
 * node n are children[child_offsets[n] .. child_offsets[n + 1]) in the order of m_nodes[n].children; a node
 * without children is a leaf (a word) and carries word_ids[n] and weights[n]. */
typedef struct ovs_bow_vocab_view {
    int32_t branching_factor;      /* m_k, informative */
    int32_t depth_levels;          /* m_L */
    int32_t weighting;             /* OVS_BOW_TF_IDF ... */
    int32_t scoring;               /* OVS_BOW_L1_NORM ... */
    int32_t num_nodes;             /* root included */
    const uint8_t* node_descs;     /* num_nodes x 32 bytes (row 0, the root's, is not read) */
    const uint32_t* child_offsets; /* num_nodes + 1, ascending */
    const uint32_t* children;      /* child_offsets[num_nodes] node ids */
    const uint32_t* word_ids;      /* num_nodes, read for leaves; any value but 0xFFFFFFFF */
    const double* weights;         /* num_nodes, read for leaves */
} ovs_bow_vocab_view;

typedef struct ovs_bow_vocab ovs_bow_vocab;

/* Copies the tree to the device.  OVS_ERR_INVALID_ARGUMENT unless the arrays describe a tree rooted at
 * node 0 in which every other node has exactly one parent. */
int ovs_bow_vocab_create(const ovs_bow_vocab_view* view, int device, ovs_bow_vocab** out);
void ovs_bow_vocab_destroy(ovs_bow_vocab* v);

/* TemplatedVocabulary::transform(features, v, fv, levelsup) for n descriptors (host buffers).
 *   word_of_feature / node_of_feature   out, n each, may be NULL: word id and feature-vector node of every
 *                                       descriptor (word 0xFFFFFFFF for a stopped word)
 *   bow_word_ids / bow_values           out, capacity n: the BowVector in ascending word order
 *   fv_node_ids / fv_offsets / fv_indices   out, capacity n / n + 1 / n: the FeatureVector in the layout
 *                                       of ovs_bow_feat_vec
 * An empty vocabulary (the root has no children) gives empty outputs, as DBoW2 does. */
int ovs_bow_transform(const ovs_bow_vocab* v, const uint8_t* descs, int n, int levelsup,
                      uint32_t* word_of_feature, uint32_t* node_of_feature,
                      uint32_t* bow_word_ids, double* bow_values, int32_t* num_words,
                      uint32_t* fv_node_ids, uint32_t* fv_offsets, uint32_t* fv_indices, int32_t* fv_num_nodes);

/* The same for n_frames descriptor sets on the device, frame f at d_descs + f * cap * 32 with d_counts[f]
 * descriptors (the layout ovs_orb_extract_batch_dev writes; d_counts NULL: every frame has `cap`).  Outputs
 * are per frame with strides cap (word / node of feature, BowVector, node ids, indices) and cap + 1
 * (offsets); d_num_words / d_fv_num_nodes hold n_frames counts.  word_of_feature / node_of_feature may
 * be NULL.  Enqueued on `stream`, not synchronised; descriptors 16-byte aligned. */
int ovs_bow_transform_batch_dev(const ovs_bow_vocab* v, const uint8_t* d_descs, const int32_t* d_counts,
                                int n_frames, int cap, int levelsup,
                                uint32_t* d_word_of_feature, uint32_t* d_node_of_feature,
                                uint32_t* d_bow_word_ids, double* d_bow_values, int32_t* d_num_words,
                                uint32_t* d_fv_node_ids, uint32_t* d_fv_offsets, uint32_t* d_fv_indices,
                                int32_t* d_fv_num_nodes, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Keyframe-database brute-force search (relocalisation over a descriptor database)           */
/* ------------------------------------------------------------------------------------------ */

/* A device-resident descriptor database: rows of 32 bytes, row r belongs to keyframe
 * r / keypts_per_keyframe.  No counterpart object in the reference, which only brute-forces one
 * frame against one keyframe (robust.cc:250-268); the selection rule (strict <, first index wins,
 * best and second best) is that loop's. */
typedef struct ovs_desc_db ovs_desc_db;
int ovs_desc_db_create(int device, ovs_desc_db** out);
void ovs_desc_db_destroy(ovs_desc_db* db);
/* Replaces the contents with n_rows host descriptors (copied). global_row_offset is added to
 * every reported row index (shard base when the database is partitioned over processes). */
int ovs_desc_db_upload(ovs_desc_db* db, const uint8_t* descs, int64_t n_rows, int64_t global_row_offset);
/* Device memory the handle itself holds (uploaded rows, the partial-result buffer of ovs_desc_db_search_dev,
 * staging of the host-buffer search); attached rows and the built-in scratch array are not the handle's.
 * Bookkeeping only: no device call. */
int ovs_desc_db_device_bytes(const ovs_desc_db* db, int64_t* bytes);
/* Adopts a device buffer the caller keeps alive (no copy). */
int ovs_desc_db_attach_dev(ovs_desc_db* db, const uint8_t* d_descs, int64_t n_rows, int64_t global_row_offset);

/* One search result: distance and global row of the best and second-best database rows. */
typedef struct ovs_knn2 {
    uint32_t dist[2]; /* 256 where absent */
    int64_t row[2];   /* -1 where absent  */
} ovs_knn2;

/* Top-2 over the database for nq query descriptors (host query/result buffers). */
int ovs_desc_db_search(ovs_desc_db* db, const uint8_t* queries, int nq, ovs_knn2* out);
/* Device buffers, enqueued on `stream`.  The per-slice partial results of a search live in one
 * buffer of the handle: searches of one database must not overlap (one stream per handle at a time;
 * attach the same device rows to a second handle to search them from a second stream). */
int ovs_desc_db_search_dev(ovs_desc_db* db, const uint8_t* d_queries, int nq, ovs_knn2* d_out, void* stream);
/* The search for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * arguments and results as ovs_desc_db_search_dev; the per-slice partial results live in a 4 MiB array
 * that is part of the library's code object (one copy per GPU, held until the process ends) instead of
 * a buffer the call would have to grow.  A search whose slices x nq x 16 bytes exceed 4 MiB returns
 * OVS_ERR_CAPACITY: with 128 to 262 144 queries (matrix-core kernel, at most 196 608 partial entries while
 * the rows are cut into more than one slice) that does not happen; beyond 262 144 queries one slice alone
 * exceeds it; with fewer than 128 queries (popcount kernel, one slice per 4096 rows) it happens when
 * ceil(rows / 4096) x nq exceeds 262 144.  The call makes no hipMalloc / hipFree; the array's address is
 * looked up when the database handle is created.  All calls on one GPU share the array: calls that
 * overlap on one GPU (different streams or threads) race on it. */
int ovs_desc_db_search_noalloc_dev(ovs_desc_db* db, const uint8_t* d_queries, int nq, ovs_knn2* d_out, void* stream);
/* Merges n_lists per-shard results (list-major: lists[s*nq + q]) into out[q] with the
 * lexicographic (dist, row) order, i.e. the serial strict-< scan over the whole database. */
int ovs_knn2_merge(const ovs_knn2* lists, int n_lists, int nq, ovs_knn2* out);
int ovs_knn2_merge_dev(const ovs_knn2* d_lists, int n_lists, int nq, ovs_knn2* d_out, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Batched extract + match on resident frames (offline sequences)                             */
/* ------------------------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------------------------ */
/* match::fuse                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* The candidate search of fuse::detect_duplication (src/openvslam/match/fuse.cc:67-102,
 * replace_mode = 0) and fuse::replace_duplication (fuse.cc:173-236, replace_mode = 1): for every
 * landmark that passed the caller's visibility gates (`valid`), the keypoint of the keyframe with
 * the smallest Hamming distance (<= HAMMING_DIST_THR_LOW, first one on ties) inside the window
 * margin * scale_factors[pred level], or -1.  Queries do not exclude each other; what happens to a
 * hit (add the observation, or merge with the landmark already there) is landmark bookkeeping and
 * stays with the caller, in query order.
 *   detect:  level gate `octave < pred - 1 || pred < octave` on SIGNED ints (fuse.cc:68,83-88); put
 *            keypts_[i].octave into the view's keypoints, as the reference reads keypts_ here
 *   replace: the same gate on UNSIGNED values (fuse.cc:191-196) plus the chi-square gate on the
 *            reprojection error (5.99146 for two, 7.81473 for three degrees of freedom when the
 *            keypoint has a stereo x_right >= 0, fuse.cc:198-222); reproj_xy_d holds the
 *            double-precision reprojection the reference computes (n x 2), q->x_right its x_right,
 *            inv_level_sigma_sq the keyframe's table (num_scale_levels floats) */
int ovs_fuse_best_candidates(const ovs_frame_view* keyfrm, const ovs_proj_queries* q, const double* reproj_xy_d,
                             const float* inv_level_sigma_sq, float margin, int replace_mode,
                             int32_t* best_idx_of_query, int device);

/* ------------------------------------------------------------------------------------------ */
/* match::area (monocular initializer)                                                        */
/* ------------------------------------------------------------------------------------------ */

/* area::match_in_consistent_area (src/openvslam/match/area.cc:8-112).
 *   undist_keypts_1 / descs_1   frm_1.undist_keypts_ and descriptors_ (only octave-0 keypoints search)
 *   frm_2                       the frame searched; its grid is queried with (margin, level 0..0)
 *   prev_matched_pts            n1 x 2 floats, in/out: search centres, updated for matched keypoints
 *   matched_indices_2_in_frm_1  out, n1 ints (-1 = none);  *num_matches = the return value
 * Window queries, Hamming distances and the overwrite rule of area.cc:46-49,75-85 (a later keypoint
 * takes over a candidate it is strictly closer to) run on the device.  Up to about 18 000 keypoints in
 * frm_2 the per-keypoint state of that rule lives in LDS; beyond that it lives in device memory and the
 * queries are resolved one per round (slower, same result).  n1 and frm_2->num_keypts < 2^20. */
int ovs_match_in_consistent_area(const ovs_keypoint* undist_keypts_1, const uint8_t* descs_1, int n1,
                                 const ovs_frame_view* frm_2, float* prev_matched_pts, int margin,
                                 float lowe_ratio, int check_orientation,
                                 int32_t* matched_indices_2_in_frm_1, uint32_t* num_matches, int device);

/* ------------------------------------------------------------------------------------------ */
/* data::landmark::compute_descriptor                                                         */
/* ------------------------------------------------------------------------------------------ */

/* landmark::compute_descriptor (src/openvslam/data/landmark.cc:127-184) for a batch of landmarks
 * (every new / merged landmark of a mapping step).  Landmark l observes the descriptors
 * descs[offsets[l] .. offsets[l+1]) (those of keyframes that are not being erased, in the
 * iteration order of its observation map); best_idx[l] is the index inside that range of the
 * descriptor with the smallest median Hamming distance to all of the landmark's descriptors
 * (median = element floor(0.5 * (n - 1)) of the sorted row, self distance included), the first
 * one on ties, or -1 for a landmark without observations.  Host buffers; at most 4096
 * observations per landmark. */
int ovs_landmark_compute_descriptors(const uint8_t* descs, const uint32_t* offsets, int n_landmarks,
                                     int32_t* best_idx, int device);

/* ------------------------------------------------------------------------------------------ */
/* Image preparation in front of the extractor (util::image_converter, util::stereo_rectifier)   */
/* ------------------------------------------------------------------------------------------ */

/* camera::color_order_t (src/openvslam/camera/base.h:29-33) */
enum { OVS_COLOR_GRAY = 0, OVS_COLOR_RGB = 1, OVS_COLOR_BGR = 2 };

/* util::convert_to_grayscale (src/openvslam/util/image_converter.cc:8-39): cv::cvtColor
 * RGB2GRAY / BGR2GRAY / RGBA2GRAY / BGRA2GRAY of 8-bit interleaved images, OpenCV 3.3.1 fixed-point
 * rule (R 4899 + G 9617 + B 1868 + 2^13) >> 14.  channels = 3 or 4.  With color order Gray the
 * reference leaves a multi-channel image untouched: OVS_ERR_UNSUPPORTED.
 * Device variant: n_frames images, enqueued on `stream`, not synchronised. */
int ovs_convert_to_grayscale_dev(const uint8_t* d_src, int n_frames, int width, int height, int channels,
                                 size_t src_row_stride, size_t src_frame_stride, int color_order,
                                 uint8_t* d_dst, size_t dst_row_stride, size_t dst_frame_stride,
                                 int device, void* stream);
int ovs_convert_to_grayscale(const uint8_t* src, int width, int height, int channels, size_t src_stride,
                             int color_order, uint8_t* dst, size_t dst_stride, int device);

/* util::equalize_histogram (src/openvslam/util/image_converter.cc:45-61): l / h = the values at
 * positions (unsigned)(0.05 n) and (unsigned)(0.95 n) of the sorted pixels, then
 * convertTo(CV_8U, 255.0 / (h - l), -255.0 l / (h - l)) (float multiply and add, round half to
 * even, saturate).  8-bit images are stretched in place; 16-bit images give an 8-bit image.
 * A frame with h == l (the reference divides by zero) is left unchanged (8 bit) / written as 0
 * (16 bit) and d_flat_flags[frame] is set to 1, else 0.  Enqueued on `stream`. */
int ovs_equalize_histogram_dev(uint8_t* d_img, int n_frames, int width, int height,
                               size_t row_stride, size_t frame_stride,
                               int32_t* d_flat_flags, int device, void* stream);
int ovs_equalize_histogram_u16_dev(const uint16_t* d_img, int n_frames, int width, int height,
                                   size_t row_stride, size_t frame_stride,
                                   uint8_t* d_dst, size_t dst_row_stride, size_t dst_frame_stride,
                                   int32_t* d_flat_flags, int device, void* stream);
/* Host buffers; bits = 8 or 16 per source pixel; h == l returns OVS_ERR_UNSUPPORTED. */
int ovs_equalize_histogram(const void* src, int width, int height, size_t src_stride, int bits,
                           uint8_t* dst, size_t dst_stride, int device);

/* One of the two cv::remap(..., INTER_LINEAR) calls of util::stereo_rectifier::rectify
 * (src/openvslam/util/stereo_rectifier.cc:56-60; border constant 0): dst(x, y) = bilinear sample of
 * src at (map_x(x, y), map_y(x, y)) with OpenCV's fixed-point rule (coordinates rounded to 1/32
 * pixel, 15-bit weights).  The maps are the CV_32FC1 pair of cv::initUndistortRectifyMap
 * (stereo_rectifier.cc:37-38,42-43), computed once by the caller; create converts them to fixed
 * point and keeps them on the device (width x height = size of the rectified image). */
typedef struct ovs_remap ovs_remap;
int ovs_remap_create(const float* map_x, const float* map_y, int width, int height, int device, ovs_remap** out);
void ovs_remap_destroy(ovs_remap* r);
int ovs_remap_apply_dev(ovs_remap* r, const uint8_t* d_src, int n_frames, int src_width, int src_height,
                        size_t src_row_stride, size_t src_frame_stride,
                        uint8_t* d_dst, size_t dst_row_stride, size_t dst_frame_stride, void* stream);
int ovs_remap_apply(ovs_remap* r, const uint8_t* src, int src_width, int src_height, size_t src_stride,
                    uint8_t* dst, size_t dst_stride);

/* ------------------------------------------------------------------------------------------ */
/* Saved maps as descriptor databases                                                           */
/* ------------------------------------------------------------------------------------------ */

/* The keyframe descriptors of a map written by io::map_database_io::save_message_pack
 * (src/openvslam/io/map_database_io.cc:19-46): root["keyframes"][std::to_string(id)]["descs"] =
 * n_keypts x 8 uint32 words of the descriptor rows (data/common.cc:71-82, data/keyframe.cc:127).
 * Host code (no GPU needed); rows come in ascending keyframe id order and feed
 * ovs_desc_db_upload.  A missing file or a file of another structure gives
 * OVS_ERR_INVALID_ARGUMENT (the reference throws std::runtime_error, map_database_io.cc:60-63). */
typedef struct ovs_map_file ovs_map_file;
int ovs_map_open(const char* path, ovs_map_file** out);
void ovs_map_close(ovs_map_file* m);
int ovs_map_num_keyframes(const ovs_map_file* m);
int64_t ovs_map_num_descriptors(const ovs_map_file* m);
/* ids: num_keyframes entries; row_offsets: num_keyframes + 1 entries (rows of keyframe i are
 * [row_offsets[i], row_offsets[i + 1])) */
int ovs_map_keyframe_rows(const ovs_map_file* m, int32_t* ids, int64_t* row_offsets);
/* descs: num_descriptors x 32 bytes */
int ovs_map_copy_descriptors(const ovs_map_file* m, uint8_t* descs);

/* ------------------------------------------------------------------------------------------ */
/* Keypoints to bearing vectors (the step of data::frame construction between undistortion and  */
/* grid assignment, src/openvslam/data/frame.cc:36-51)                                          */
/* ------------------------------------------------------------------------------------------ */

/* camera::model_type_t (src/openvslam/camera/base.h:22-26) */
enum { OVS_CAMERA_PERSPECTIVE = 0, OVS_CAMERA_FISHEYE = 1, OVS_CAMERA_EQUIRECTANGULAR = 2 };

/* The camera parameters the two conversions read: fx_, fy_, cx_, cy_ (double) of the perspective
 * and fisheye models and their distortion parameters in the order of cv_dist_params_ -- k1, k2, p1,
 * p2, k3 for perspective (camera/perspective.cc:22), k1, k2, k3, k4 for fisheye (camera/fisheye.cc:22)
 * -- and cols_ / rows_ (unsigned int) of the equirectangular one.  The bearing conversion does not
 * read `dist`. */
typedef struct ovs_camera {
    int32_t model;
    uint32_t cols, rows;
    int32_t reserved;
    double fx, fy, cx, cy;
    double dist[5];
} ovs_camera;

/* camera::perspective::convert_keypoints_to_bearings (src/openvslam/camera/perspective.cc:130-138),
 * camera::fisheye::convert_keypoints_to_bearings (camera/fisheye.cc:169-177),
 * camera::equirectangular::convert_keypoints_to_bearings (camera/equirectangular.cc:45-56).
 *   undist_keypts  undistorted keypoints; bearings out, 3 doubles per keypoint (Vec3_t)
 * Device variant: n_frames x cap keypoints, d_counts (n_frames, or NULL = cap each) limits the
 * entries converted per frame, d_bearings n_frames x cap x 3; enqueued on `stream`. */
int ovs_convert_keypoints_to_bearings_dev(const ovs_camera* cam, const ovs_keypoint* d_undist_keypts,
                                          const int32_t* d_counts, int n_frames, int cap,
                                          double* d_bearings, int device, void* stream);
int ovs_convert_keypoints_to_bearings(const ovs_camera* cam, const ovs_keypoint* undist_keypts, int n,
                                      double* bearings, int device);

/* camera::perspective::undistort_keypoints (camera/perspective.cc:99-128: cv::undistortPoints with
 * R empty, P = the camera matrix and the termination criterion 20 iterations / 1e-6),
 * camera::fisheye::undistort_keypoints (camera/fisheye.cc:139-167: cv::fisheye::undistortPoints),
 * camera::equirectangular::undistort_keypoints (camera/equirectangular.cc:41-43: a copy).
 * The reference hands OpenCV the camera matrix and the distortion parameters as float matrices
 * (perspective.cc:21-22, fisheye.cc:21-22); the narrowing happens inside these entry points.
 *   undist_keypts  pt undistorted; angle, size, octave copied; response 0 and class_id -1 (the
 *                  defaults of cv::KeyPoint) for perspective / fisheye, the whole record copied for
 *                  equirectangular
 * Device variant: as above; d_bearings (n_frames x cap x 3 doubles) may be NULL, otherwise the
 * bearing of every undistorted keypoint is written by the same launch. */
int ovs_undistort_keypoints_dev(const ovs_camera* cam, const ovs_keypoint* d_dist_keypts,
                                const int32_t* d_counts, int n_frames, int cap,
                                ovs_keypoint* d_undist_keypts, double* d_bearings, int device, void* stream);
int ovs_undistort_keypoints(const ovs_camera* cam, const ovs_keypoint* dist_keypts, int n,
                            ovs_keypoint* undist_keypts, int device);

/* Concurrency of the device-pointer entry points: an extractor / database handle serves one call
 * at a time (its buffers belong to the stream of the last call), while the matcher entry points
 * keep their scratch memory per (host thread, device, stream), so independent batches may be in
 * flight on several streams at once. */

/* For every frame f >= 1 of the batch last extracted with ovs_orb_extract_batch_dev (outputs
 * d_kps/d_desc/d_counts as written by that call), runs projection::match_current_and_last_frames
 * (projection.cc:84-209) with frame f as `curr` and frame f-1 as `last`, every last keypoint
 * carrying a landmark whose descriptor is its own and whose reprojection is given by the caller:
 *   d_reproj_xy  n_frames x cap x 2 floats: reprojection of last-frame keypoint i into frame f
 *                stored at [f][i]; entries with x < -1e30 are treated as not in the image
 *   undistorted keypoints are the extracted ones (zero-distortion camera), monocular
 *   d_matched    out, n_frames x cap int32: per curr keypoint the last-frame index or -1
 *   d_num_matches out, n_frames uint32 (entry 0 is 0). */
int ovs_match_batch_current_and_last_dev(const ovs_keypoint* d_kps, const uint8_t* d_desc,
                                         const int32_t* d_counts, int n_frames, int cap,
                                         const float* d_reproj_xy, const ovs_grid* grid,
                                         const float* scale_factors, int num_scale_levels,
                                         float margin, int check_orientation,
                                         int32_t* d_matched, uint32_t* d_num_matches,
                                         int device, void* stream);

/* Dense results for the trip to the host.  The reference returns exactly sized containers (a
 * std::vector<cv::KeyPoint> and a cv::Mat with one row per keypoint, feature/orb_extractor.cc:116-120;
 * landmark vectors of num_keypts_ entries); the batch entry points above write frame f at slot f of
 * capacity-strided arrays because the counts only exist on the device.  This call moves the first
 * min(d_counts[f], cap) records of every frame to d_offsets[f] of dense arrays (d_offsets: n_frames + 1
 * int32, the exclusive prefix sum of the counts, d_offsets[n_frames] = total), so that a download moves the
 * payload and not the capacity.  Any of the three (source, destination) pairs may be NULL; the dense
 * arrays need room for the total (n_frames x cap is always enough).  Enqueued on `stream`. */
int ovs_pack_batch_results_dev(const int32_t* d_counts, int n_frames, int cap,
                               const ovs_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_matched,
                               int32_t* d_offsets, ovs_keypoint* d_packed_kps, uint8_t* d_packed_desc,
                               int32_t* d_packed_matched, int device, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* OVS_H */
