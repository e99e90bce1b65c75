"""Host-side mirror of feature::orb_extractor (src/openvslam/feature/orb_extractor.h:13-150) over
the C ABI.  Method names and semantics follow the reference class; the work runs on the GPU."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, lib


class orb_params:
    """feature::orb_params (src/openvslam/feature/orb_params.h)."""

    def __init__(self, max_num_keypts=2000, scale_factor=1.2, num_levels=8, ini_fast_thr=20, min_fast_thr=7,
                 mask_rects=()):
        self.max_num_keypts_ = max_num_keypts
        self.scale_factor_ = scale_factor
        self.num_levels_ = num_levels
        self.ini_fast_thr_ = ini_fast_thr
        self.min_fast_thr = min_fast_thr
        self.mask_rects_ = [list(r) for r in mask_rects]
        for v in self.mask_rects_:
            # orb_params.cc:14-24
            if len(v) != 4:
                raise RuntimeError("mask rectangle must contain four parameters")
            if v[0] >= v[1]:
                raise RuntimeError("x_max must be greater than x_min")
            if v[2] >= v[3]:
                raise RuntimeError("y_max must be greater than x_min")

    @staticmethod
    def _table(name, num_scale_levels, scale_factor):
        out = np.zeros(num_scale_levels, dtype=np.float32)
        getattr(lib(), "ovs_calc_" + name)(num_scale_levels, scale_factor, A.ptr(out, A.f32p))
        return out

    @staticmethod
    def calc_scale_factors(num_scale_levels, scale_factor):
        return orb_params._table("scale_factors", num_scale_levels, scale_factor)

    @staticmethod
    def calc_inv_scale_factors(num_scale_levels, scale_factor):
        return orb_params._table("inv_scale_factors", num_scale_levels, scale_factor)

    @staticmethod
    def calc_level_sigma_sq(num_scale_levels, scale_factor):
        return orb_params._table("level_sigma_sq", num_scale_levels, scale_factor)

    @staticmethod
    def calc_inv_level_sigma_sq(num_scale_levels, scale_factor):
        return orb_params._table("inv_level_sigma_sq", num_scale_levels, scale_factor)


class orb_extractor:
    def __init__(self, params=None, device=0, **kw):
        if params is None:
            params = orb_params(**kw)
        self._abi_params = A.make_orb_params(params.max_num_keypts_, params.scale_factor_, params.num_levels_,
                                             params.ini_fast_thr_, params.min_fast_thr, params.mask_rects_)
        h = C.c_void_p()
        check(lib().ovs_orb_create(self._abi_params.ref(), device, C.byref(h)))
        self._h = h
        self._destroy = lib().ovs_orb_destroy  # kept on the object: module globals may be gone at interpreter exit
        self.device = device
        self._last_shape = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    # ---- extract --------------------------------------------------------------------------------
    def extract(self, in_image, in_image_mask=None):
        """Returns (keypts, descriptors): structured array with cv::KeyPoint layout and uint8 [n, 32].
        An empty image returns (None, None), i.e. the outputs are left untouched."""
        if in_image is None or in_image.size == 0:
            return None, None
        img = np.ascontiguousarray(in_image, dtype=np.uint8)
        assert img.ndim == 2, "image must be CV_8UC1"
        h, w = img.shape
        mask = None
        if in_image_mask is not None and in_image_mask.size:
            mask = np.ascontiguousarray(in_image_mask, dtype=np.uint8)
            assert mask.shape == img.shape
        cap = lib().ovs_orb_keypoint_capacity(self._h, w, h)
        if cap < 0:
            check(cap)
        kps = np.zeros(cap, dtype=A.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        check(lib().ovs_orb_extract(self._h, A.ptr(img, A.u8p), w, h, w, A.ptr(mask, A.u8p), w if mask is not None else 0,
                                    A.ptr(kps, A.kpp), A.ptr(desc, A.u8p), cap, C.byref(n)))
        self._last_shape = (h, w)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def keypoint_capacity(self, width, height):
        cap = lib().ovs_orb_keypoint_capacity(self._h, width, height)
        if cap < 0:
            check(cap)
        return cap

    def extract_batch_dev(self, d_imgs, n_frames, width, height, row_stride, frame_stride, d_kps, d_desc, d_counts, cap,
                          stream=0, d_mask=0, mask_stride=0):
        """All pointers are device addresses (ints); work is enqueued on `stream` (hipStream_t)."""
        check(lib().ovs_orb_extract_batch_dev(self._h, d_imgs, n_frames, width, height, row_stride, frame_stride,
                                              d_mask or None, mask_stride, d_kps, d_desc, d_counts, cap, stream or None))
        self._last_shape = (height, width)

    # ---- image_pyramid_ ---------------------------------------------------------------------------
    def level_size(self, level, width=None, height=None):
        if width is None:
            height, width = self._last_shape
        w, h = C.c_int(), C.c_int()
        check(lib().ovs_orb_level_size(self._h, width, height, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def image_pyramid(self, level, frame=0):
        """Host copy of image_pyramid_[level] of the last extract (levels >= 1 live on the device)."""
        w, h = self.level_size(level)
        out = np.zeros((h, w), dtype=np.uint8)
        check(lib().ovs_orb_download_pyramid_level(self._h, frame, level, A.ptr(out, A.u8p), w))
        return out

    def blurred_level(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), dtype=np.uint8)
        check(lib().ovs_orb_download_blurred_level(self._h, frame, level, A.ptr(out, A.u8p), w))
        return out

    # ---- coefficients of the 7 x 7 blur: which cv::GaussianBlur the caller's OpenCV has (see include/ovs.h)
    def set_blur_kernel(self, variant):
        check(lib().ovs_orb_set_blur_kernel(self._h, variant))

    def get_blur_kernel(self):
        return lib().ovs_orb_get_blur_kernel(self._h)

    # ---- getters / setters (orb_extractor.cc:146-203) -------------------------------------------
    def get_max_num_keypoints(self):
        return lib().ovs_orb_get_max_num_keypoints(self._h)

    def set_max_num_keypoints(self, v):
        check(lib().ovs_orb_set_max_num_keypoints(self._h, v))

    def get_scale_factor(self):
        return lib().ovs_orb_get_scale_factor(self._h)

    def set_scale_factor(self, v):
        check(lib().ovs_orb_set_scale_factor(self._h, v))

    def get_num_scale_levels(self):
        return lib().ovs_orb_get_num_scale_levels(self._h)

    def set_num_scale_levels(self, v):
        check(lib().ovs_orb_set_num_scale_levels(self._h, v))

    def get_initial_fast_threshold(self):
        return lib().ovs_orb_get_initial_fast_threshold(self._h)

    def set_initial_fast_threshold(self, v):
        check(lib().ovs_orb_set_initial_fast_threshold(self._h, v))

    def get_minimum_fast_threshold(self):
        return lib().ovs_orb_get_minimum_fast_threshold(self._h)

    def set_minimum_fast_threshold(self, v):
        check(lib().ovs_orb_set_minimum_fast_threshold(self._h, v))

    def _vec(self, name):
        out = np.zeros(self.get_num_scale_levels(), dtype=np.float32)
        check(getattr(lib(), "ovs_orb_get_" + name)(self._h, A.ptr(out, A.f32p)))
        return out

    def get_scale_factors(self):
        return self._vec("scale_factors")

    def get_inv_scale_factors(self):
        return self._vec("inv_scale_factors")

    def get_level_sigma_sq(self):
        return self._vec("level_sigma_sq")

    def get_inv_level_sigma_sq(self):
        return self._vec("inv_level_sigma_sq")

    def get_num_keypts_per_level(self):
        out = np.zeros(self.get_num_scale_levels(), dtype=np.uint32)
        check(lib().ovs_orb_get_num_keypts_per_level(self._h, A.ptr(out, A.u32p)))
        return out
