"""Host-side mirrors of match::{projection, robust, bow_tree, stereo} (src/openvslam/match/) over the
C ABI.  The reference's methods take data::frame / data::keyframe objects and mutate landmark
pointers; here they take the flattened views of openvslam_amd._abi and return index arrays:
`matched[i]` = index of the query / keypoint whose landmark ends up in slot i, -1 = untouched,
-2 = written and then cleared by the rotation check.  All of it runs on the GPU."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, lib

HAMMING_DIST_THR_LOW = 50
HAMMING_DIST_THR_HIGH = 100
MAX_HAMMING_DIST = 256


def compute_descriptor_distance_32(desc_1, desc_2, device=0):
    """match::compute_descriptor_distance_32 for one pair or for n pairs (arrays of shape [n, 32])."""
    a = np.ascontiguousarray(desc_1, dtype=np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(desc_2, dtype=np.uint8).reshape(-1, 32)
    assert a.shape == b.shape
    out = np.zeros(len(a), dtype=np.uint32)
    check(lib().ovs_hamming_pairs(A.ptr(a, A.u8p), A.ptr(b, A.u8p), len(a), A.ptr(out, A.u32p), device))
    return int(out[0]) if np.ndim(desc_1) == 1 else out


def descriptor_distance_matrix(descs_a, descs_b, device=0):
    a = np.ascontiguousarray(descs_a, dtype=np.uint8).reshape(-1, 32)
    b = np.ascontiguousarray(descs_b, dtype=np.uint8).reshape(-1, 32)
    out = np.zeros((len(a), len(b)), dtype=np.uint16)
    check(lib().ovs_hamming_matrix(A.ptr(a, A.u8p), len(a), A.ptr(b, A.u8p), len(b), A.ptr(out, A.u16p), device))
    return out


class base:
    def __init__(self, lowe_ratio, check_orientation, device=0):
        self.lowe_ratio_ = lowe_ratio
        self.check_orientation_ = bool(check_orientation)
        self.device = device


class projection(base):
    def __init__(self, lowe_ratio=0.6, check_orientation=True, device=0):
        super().__init__(lowe_ratio, check_orientation, device)

    def _run(self, fn, frame_view, queries, n, *args):
        matched = np.full(n, -1, dtype=np.int32)
        num = C.c_uint32(0)
        check(fn(frame_view.ref(), queries.ref(), *args, A.ptr(matched, A.i32p), C.byref(num), self.device))
        return num.value, matched

    def match_current_and_last_frames(self, curr_view, queries, margin, curr_occupied=None, direction=A.DIR_NONE,
                                      last_num_scale_levels=8):
        occ = None if curr_occupied is None else np.ascontiguousarray(curr_occupied, dtype=np.uint8)
        return self._run(lib().ovs_match_current_and_last_frames, curr_view, queries, curr_view.struct.num_keypts,
                         A.ptr(occ, A.u8p), margin, direction, last_num_scale_levels, self.lowe_ratio_, int(self.check_orientation_))

    def match_frame_and_landmarks(self, frm_view, queries, margin=5.0, frm_occupied=None):
        occ = None if frm_occupied is None else np.ascontiguousarray(frm_occupied, dtype=np.uint8)
        return self._run(lib().ovs_match_frame_and_landmarks, frm_view, queries, frm_view.struct.num_keypts,
                         A.ptr(occ, A.u8p), margin, self.lowe_ratio_)

    def match_frame_and_keyframe(self, curr_view, queries, margin, hamm_dist_thr, curr_occupied=None):
        occ = None if curr_occupied is None else np.ascontiguousarray(curr_occupied, dtype=np.uint8)
        return self._run(lib().ovs_match_frame_and_keyframe, curr_view, queries, curr_view.struct.num_keypts,
                         A.ptr(occ, A.u8p), margin, hamm_dist_thr, int(self.check_orientation_))

    def match_by_Sim3_transform(self, keyfrm_view, queries, margin, keyfrm_occupied=None):
        occ = None if keyfrm_occupied is None else np.ascontiguousarray(keyfrm_occupied, dtype=np.uint8)
        return self._run(lib().ovs_match_by_sim3_transform, keyfrm_view, queries, keyfrm_view.struct.num_keypts,
                         A.ptr(occ, A.u8p), margin)

    def match_keyframes_mutually(self, keyfrm_1_view, queries_2_in_1, keyfrm_2_view, queries_1_in_2, margin):
        """queries_1_in_2: landmarks of keyframe 1 projected into keyframe 2 (one per keyframe-1 keypoint),
        queries_2_in_1 the reverse.  Returns (num_matches, idx2_of_1)."""
        n1, n2 = keyfrm_1_view.struct.num_keypts, keyfrm_2_view.struct.num_keypts
        b12 = np.full(n1, -1, dtype=np.int32)
        b21 = np.full(n2, -1, dtype=np.int32)
        check(lib().ovs_match_keyframe_one_way(keyfrm_2_view.ref(), queries_1_in_2.ref(), margin, A.ptr(b12, A.i32p), self.device))
        check(lib().ovs_match_keyframe_one_way(keyfrm_1_view.ref(), queries_2_in_1.ref(), margin, A.ptr(b21, A.i32p), self.device))
        out = np.full(n1, -1, dtype=np.int32)
        num = C.c_uint32(0)
        check(lib().ovs_cross_check(A.ptr(b12, A.i32p), n1, A.ptr(b21, A.i32p), n2, A.ptr(out, A.i32p), C.byref(num)))
        return num.value, out


class robust(base):
    def __init__(self, lowe_ratio=0.75, check_orientation=True, device=0):
        super().__init__(lowe_ratio, check_orientation, device)

    def brute_force_match(self, descs_1, angles_1, descs_2, angles_2, valid_2=None):
        """Returns the (idx_1, idx_2) pairs of robust::brute_force_match, ascending in idx_1."""
        d1 = np.ascontiguousarray(descs_1, dtype=np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(descs_2, dtype=np.uint8).reshape(-1, 32)
        a1 = np.ascontiguousarray(angles_1, dtype=np.float32)
        a2 = np.ascontiguousarray(angles_2, dtype=np.float32)
        v2 = None if valid_2 is None else np.ascontiguousarray(valid_2, dtype=np.uint8)
        m = np.zeros((max(len(d1), 1), 2), dtype=np.int32)
        num = C.c_uint32(0)
        check(lib().ovs_brute_force_match(A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), len(d1), A.ptr(d2, A.u8p), A.ptr(a2, A.f32p),
                                          A.ptr(v2, A.u8p), len(d2), self.lowe_ratio_, int(self.check_orientation_),
                                          A.ptr(m, A.i32p), C.byref(num), self.device))
        return m[:num.value].copy()


class bow_tree(base):
    def __init__(self, lowe_ratio=0.6, check_orientation=True, device=0):
        super().__init__(lowe_ratio, check_orientation, device)

    def match_frame_and_keyframe(self, kf_vec, kf_descs, kf_angles, valid_kf, frm_vec, frm_descs, frm_angles):
        kd = np.ascontiguousarray(kf_descs, dtype=np.uint8).reshape(-1, 32)
        fd = np.ascontiguousarray(frm_descs, dtype=np.uint8).reshape(-1, 32)
        ka = np.ascontiguousarray(kf_angles, dtype=np.float32)
        fa = np.ascontiguousarray(frm_angles, dtype=np.float32)
        vk = None if valid_kf is None else np.ascontiguousarray(valid_kf, dtype=np.uint8)
        m = np.full(len(fd), -1, dtype=np.int32)
        num = C.c_uint32(0)
        check(lib().ovs_bow_match_frame_and_keyframe(kf_vec.ref(), A.ptr(kd, A.u8p), A.ptr(ka, A.f32p), A.ptr(vk, A.u8p), len(kd),
                                                     frm_vec.ref(), A.ptr(fd, A.u8p), A.ptr(fa, A.f32p), len(fd),
                                                     self.lowe_ratio_, int(self.check_orientation_), A.ptr(m, A.i32p), C.byref(num), self.device))
        return num.value, m

    def match_keyframes(self, vec_1, descs_1, angles_1, valid_1, vec_2, descs_2, angles_2, valid_2):
        d1 = np.ascontiguousarray(descs_1, dtype=np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(descs_2, dtype=np.uint8).reshape(-1, 32)
        a1 = np.ascontiguousarray(angles_1, dtype=np.float32)
        a2 = np.ascontiguousarray(angles_2, dtype=np.float32)
        v1 = None if valid_1 is None else np.ascontiguousarray(valid_1, dtype=np.uint8)
        v2 = None if valid_2 is None else np.ascontiguousarray(valid_2, dtype=np.uint8)
        m = np.full(len(d1), -1, dtype=np.int32)
        num = C.c_uint32(0)
        check(lib().ovs_bow_match_keyframes(vec_1.ref(), A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), A.ptr(v1, A.u8p), len(d1),
                                            vec_2.ref(), A.ptr(d2, A.u8p), A.ptr(a2, A.f32p), A.ptr(v2, A.u8p), len(d2),
                                            self.lowe_ratio_, int(self.check_orientation_), A.ptr(m, A.i32p), C.byref(num), self.device))
        return num.value, m


class stereo:
    """match::stereo: constructed with both pyramids, keypoints and descriptors, compute() returns
    (stereo_x_right, depths)."""

    def __init__(self, left_image_pyramid, right_image_pyramid, keypts_left, keypts_right, descs_left, descs_right,
                 scale_factors, inv_scale_factors, focal_x_baseline, true_baseline, device=0):
        self._l = [np.ascontiguousarray(im, dtype=np.uint8) for im in left_image_pyramid]
        self._r = [np.ascontiguousarray(im, dtype=np.uint8) for im in right_image_pyramid]
        self._kl = np.ascontiguousarray(keypts_left, dtype=A.KEYPOINT_DTYPE)
        self._kr = np.ascontiguousarray(keypts_right, dtype=A.KEYPOINT_DTYPE)
        self._dl = np.ascontiguousarray(descs_left, dtype=np.uint8).reshape(-1, 32)
        self._dr = np.ascontiguousarray(descs_right, dtype=np.uint8).reshape(-1, 32)
        self._sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        self._isf = np.ascontiguousarray(inv_scale_factors, dtype=np.float32)
        self.focal_x_baseline_ = np.float32(focal_x_baseline)
        self.true_baseline_ = np.float32(true_baseline)
        self.device = device

    @staticmethod
    def _views(levels):
        arr = (A.ImageView * len(levels))()
        for i, im in enumerate(levels):
            arr[i].data = A.ptr(im, A.u8p)
            arr[i].width, arr[i].height, arr[i].stride = im.shape[1], im.shape[0], im.shape[1]
        return arr

    def compute(self):
        n = len(self._kl)
        xr = np.full(n, -1.0, dtype=np.float32)
        depths = np.full(n, -1.0, dtype=np.float32)
        check(lib().ovs_stereo_compute(self._views(self._l), self._views(self._r), len(self._l),
                                       A.ptr(self._kl, A.kpp), n, A.ptr(self._kr, A.kpp), len(self._kr),
                                       A.ptr(self._dl, A.u8p), A.ptr(self._dr, A.u8p), A.ptr(self._sf, A.f32p), A.ptr(self._isf, A.f32p),
                                       self.focal_x_baseline_, self.true_baseline_, A.ptr(xr, A.f32p), A.ptr(depths, A.f32p), self.device))
        return xr, depths


class area(base):
    """match::area (src/openvslam/match/area.h): the matcher of the monocular initializer."""

    def __init__(self, lowe_ratio=0.9, check_orientation=True, device=0):
        super().__init__(lowe_ratio, check_orientation, device)

    def match_in_consistent_area(self, undist_keypts_1, descs_1, frm_2_view, prev_matched_pts, margin):
        """Returns (num_matches, matched_indices_2_in_frm_1); prev_matched_pts ([n1, 2] float32) is updated in place."""
        k1 = np.ascontiguousarray(undist_keypts_1, dtype=A.KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(descs_1, dtype=np.uint8).reshape(-1, 32)
        assert prev_matched_pts.dtype == np.float32 and prev_matched_pts.flags["C_CONTIGUOUS"] and prev_matched_pts.shape == (len(k1), 2)
        matched = np.full(len(k1), -1, dtype=np.int32)
        num = C.c_uint32(0)
        check(lib().ovs_match_in_consistent_area(A.ptr(k1, A.kpp), A.ptr(d1, A.u8p), len(k1), frm_2_view.ref(), A.ptr(prev_matched_pts, A.f32p),
                                                 int(margin), self.lowe_ratio_, int(self.check_orientation_), A.ptr(matched, A.i32p),
                                                 C.byref(num), self.device))
        return num.value, matched


def landmark_compute_descriptors(descs, offsets, device=0):
    """data::landmark::compute_descriptor for a batch of landmarks: landmark l observes
    descs[offsets[l]:offsets[l+1]]; returns the index (inside its range) of its representative
    descriptor, -1 where a landmark has no observation."""
    d = np.ascontiguousarray(descs, dtype=np.uint8).reshape(-1, 32)
    off = np.ascontiguousarray(offsets, dtype=np.uint32)
    best = np.full(len(off) - 1, -9, dtype=np.int32)
    check(lib().ovs_landmark_compute_descriptors(A.ptr(d, A.u8p), A.ptr(off, A.u32p), len(off) - 1, A.ptr(best, A.i32p), device))
    return best


class fuse:
    """The device part of match::fuse (src/openvslam/match/fuse.cc): per landmark the best keypoint
    of the keyframe; the caller applies add_observation / replace in query order."""

    def __init__(self, device=0):
        self.device = device

    def best_candidates(self, keyfrm_view, queries, margin, reproj_xy_d=None, inv_level_sigma_sq=None):
        """detect_duplication search when reproj_xy_d is None, replace_duplication search otherwise."""
        n = queries.struct.n
        best = np.full(n, -9, dtype=np.int32)
        rd = None if reproj_xy_d is None else np.ascontiguousarray(reproj_xy_d, dtype=np.float64).reshape(n, 2)
        isg = None if inv_level_sigma_sq is None else np.ascontiguousarray(inv_level_sigma_sq, dtype=np.float32)
        check(lib().ovs_fuse_best_candidates(keyfrm_view.ref(), queries.ref(), A.ptr(rd, A.f64p), A.ptr(isg, A.f32p), margin,
                                             0 if rd is None else 1, A.ptr(best, A.i32p), self.device))
        return best
