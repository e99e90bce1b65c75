"""Host-side mirror of the camera models' keypoint-to-bearing conversion over the C ABI
(src/openvslam/camera/perspective.cc:130-138, fisheye.cc:169-177, equirectangular.cc:45-56): the
step of data::frame construction that follows keypoint undistortion (data/frame.cc:36-51)."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, lib


class base:
    """The fields of camera::base and its subclasses that the conversion reads."""

    model = None

    def __init__(self, cols=0, rows=0, fx=0.0, fy=0.0, cx=0.0, cy=0.0, device=0):
        c = A.Camera()
        c.model = self.model
        c.cols, c.rows = int(cols), int(rows)
        c.fx, c.fy, c.cx, c.cy = float(fx), float(fy), float(cx), float(cy)
        self._cam = c
        self.device = device

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
    model = A.CAMERA_PERSPECTIVE

    def __init__(self, fx, fy, cx, cy, cols=0, rows=0, device=0):
        super().__init__(cols, rows, fx, fy, cx, cy, device)


class fisheye(base):
    model = A.CAMERA_FISHEYE

    def __init__(self, fx, fy, cx, cy, cols=0, rows=0, device=0):
        super().__init__(cols, rows, fx, fy, cx, cy, device)


class equirectangular(base):
    model = A.CAMERA_EQUIRECTANGULAR

    def __init__(self, cols, rows, device=0):
        super().__init__(cols, rows, device=device)
