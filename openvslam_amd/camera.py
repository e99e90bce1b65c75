"""Host-side mirror of the camera models' per-keypoint conversions over the C ABI: keypoint
undistortion (src/openvslam/camera/perspective.cc:99-128, fisheye.cc:139-167,
equirectangular.cc:41-43) and keypoints to bearing vectors (perspective.cc:130-138,
fisheye.cc:169-177, equirectangular.cc:45-56), the two steps of data::frame construction between
feature extraction and grid assignment (data/frame.cc:36-51)."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, lib


class base:
    """The fields of camera::base and its subclasses that the conversions read."""

    model = None

    def __init__(self, cols=0, rows=0, fx=0.0, fy=0.0, cx=0.0, cy=0.0, dist=(), device=0):
        c = A.Camera()
        c.model = self.model
        c.cols, c.rows = int(cols), int(rows)
        c.fx, c.fy, c.cx, c.cy = float(fx), float(fy), float(cx), float(cy)
        for k, v in enumerate(dist):
            c.dist[k] = float(v)
        self._cam = c
        self.device = device

    def undistort_keypoints(self, dist_keypts):
        """dist_keypts: structured array with the layout of cv::KeyPoint; returns the undistorted records."""
        kps = np.ascontiguousarray(dist_keypts, dtype=A.KEYPOINT_DTYPE)
        out = np.zeros(len(kps), dtype=A.KEYPOINT_DTYPE)
        check(lib().ovs_undistort_keypoints(C.byref(self._cam), A.ptr(kps, A.kpp), len(kps), A.ptr(out, A.kpp), self.device))
        return out

    def undistort_keypoints_dev(self, d_dist_keypts, d_counts, n_frames, cap, d_undist_keypts, d_bearings=None, stream=None):
        """Device arrays (addresses): n_frames x cap keypoints in and out; with d_bearings the same
        launch also writes n_frames x cap x 3 doubles."""
        check(lib().ovs_undistort_keypoints_dev(C.byref(self._cam), d_dist_keypts, d_counts, n_frames, cap, d_undist_keypts,
                                                d_bearings, self.device, stream))

    def convert_keypoints_to_bearings(self, undist_keypts):
        """undist_keypts: structured array with the layout of cv::KeyPoint; returns n x 3 doubles."""
        kps = np.ascontiguousarray(undist_keypts, dtype=A.KEYPOINT_DTYPE)
        out = np.empty((len(kps), 3), dtype=np.float64)
        check(lib().ovs_convert_keypoints_to_bearings(C.byref(self._cam), A.ptr(kps, A.kpp), len(kps), A.ptr(out, A.f64p), self.device))
        return out

    def convert_keypoints_to_bearings_dev(self, d_undist_keypts, d_counts, n_frames, cap, d_bearings, stream=None):
        """Device arrays (addresses): n_frames x cap keypoints in, n_frames x cap x 3 doubles out."""
        check(lib().ovs_convert_keypoints_to_bearings_dev(C.byref(self._cam), d_undist_keypts, d_counts, n_frames, cap,
                                                          d_bearings, self.device, stream))


class perspective(base):
    """camera::perspective (camera/perspective.h): distortion k1, k2, p1, p2, k3."""
    model = A.CAMERA_PERSPECTIVE

    def __init__(self, fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0, k3=0.0, cols=0, rows=0, device=0):
        super().__init__(cols, rows, fx, fy, cx, cy, (k1, k2, p1, p2, k3), device)


class fisheye(base):
    """camera::fisheye (camera/fisheye.h): distortion k1, k2, k3, k4."""
    model = A.CAMERA_FISHEYE

    def __init__(self, fx, fy, cx, cy, k1=0.0, k2=0.0, k3=0.0, k4=0.0, cols=0, rows=0, device=0):
        super().__init__(cols, rows, fx, fy, cx, cy, (k1, k2, k3, k4), device)


class equirectangular(base):
    """camera::equirectangular (camera/equirectangular.h)."""
    model = A.CAMERA_EQUIRECTANGULAR

    def __init__(self, cols, rows, device=0):
        super().__init__(cols, rows, device=device)
