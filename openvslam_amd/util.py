"""Host-side mirror of util::image_converter (src/openvslam/util/image_converter.h) and
util::stereo_rectifier (src/openvslam/util/stereo_rectifier.h) over the C ABI: the image
preparation that runs in front of feature::orb_extractor::extract()."""
import ctypes as C
import enum

import numpy as np

from . import _abi as A
from ._lib import check, lib


class color_order_t(enum.IntEnum):
    """camera::color_order_t (src/openvslam/camera/base.h:32-36)."""
    Gray = 0
    RGB = 1
    BGR = 2


def convert_to_grayscale(img, in_color_order, device=0):
    """util::convert_to_grayscale (image_converter.cc:8-39).  The reference converts in place; this
    returns the converted image (or `img` itself where the reference leaves it untouched)."""
    img = np.asarray(img)
    if img.ndim == 2 or img.shape[2] == 1 or color_order_t(in_color_order) == color_order_t.Gray or img.shape[2] not in (3, 4):
        return img
    src = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, cn = src.shape
    out = np.empty((h, w), dtype=np.uint8)
    check(lib().ovs_convert_to_grayscale(A.ptr(src, A.u8p), w, h, cn, w * cn, int(in_color_order), A.ptr(out, A.u8p), w, device))
    return out


def equalize_histogram(img, device=0):
    """util::equalize_histogram (image_converter.cc:45-61): CV_8UC1 or CV_16UC1 in, CV_8UC1 out."""
    img = np.asarray(img)
    assert img.ndim == 2 and img.dtype in (np.uint8, np.uint16)
    src = np.ascontiguousarray(img)
    h, w = src.shape
    out = np.empty((h, w), dtype=np.uint8)
    check(lib().ovs_equalize_histogram(src.ctypes.data_as(C.c_void_p), w, h, w * src.itemsize, 8 * src.itemsize, A.ptr(out, A.u8p), w, device))
    return out


class remap:
    """cv::remap(src, dst, map_x, map_y, cv::INTER_LINEAR) with a fixed CV_32FC1 map pair."""

    def __init__(self, map_x, map_y, device=0):
        self._mx = np.ascontiguousarray(map_x, dtype=np.float32)
        self._my = np.ascontiguousarray(map_y, dtype=np.float32)
        assert self._mx.shape == self._my.shape and self._mx.ndim == 2
        self.height, self.width = self._mx.shape
        h = C.c_void_p()
        check(lib().ovs_remap_create(A.ptr(self._mx, A.f32p), A.ptr(self._my, A.f32p), self.width, self.height, device, C.byref(h)))
        self._h = h
        self._destroy = lib().ovs_remap_destroy  # kept on the object: module globals may be gone at interpreter exit

    def __del__(self):
        if getattr(self, "_h", None):
            self._destroy(self._h)
            self._h = None

    def apply(self, img):
        src = np.ascontiguousarray(img, dtype=np.uint8)
        out = np.empty((self.height, self.width), dtype=np.uint8)
        check(lib().ovs_remap_apply(self._h, A.ptr(src, A.u8p), src.shape[1], src.shape[0], src.shape[1], A.ptr(out, A.u8p), self.width))
        return out

    def apply_batch_dev(self, d_src, n_frames, src_w, src_h, src_row_stride, src_frame_stride, d_dst, dst_row_stride, dst_frame_stride, stream):
        check(lib().ovs_remap_apply_dev(self._h, d_src, n_frames, src_w, src_h, src_row_stride, src_frame_stride,
                                        d_dst, dst_row_stride, dst_frame_stride, stream))


class stereo_rectifier:
    """util::stereo_rectifier (stereo_rectifier.cc:12-60).  The reference builds its four maps once
    with cv::initUndistortRectifyMap from the StereoRectifier.* entries of the configuration; here
    the caller passes those maps."""

    def __init__(self, undist_map_x_l, undist_map_y_l, undist_map_x_r, undist_map_y_r, device=0):
        self._l = remap(undist_map_x_l, undist_map_y_l, device)
        self._r = remap(undist_map_x_r, undist_map_y_r, device)

    def rectify(self, in_img_l, in_img_r):
        return self._l.apply(in_img_l), self._r.apply(in_img_r)
