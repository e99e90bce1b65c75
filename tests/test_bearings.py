"""camera::{perspective,fisheye,equirectangular}::convert_keypoints_to_bearings
(src/openvslam/camera/perspective.cc:130-138, fisheye.cc:169-177, equirectangular.cc:45-56): the
oracle against a numpy restatement and hand-computed answers on the CPU, the GPU against the oracle.

The reference holds no test of these functions: parity unpinned.  The perspective / fisheye form
consists of correctly rounded IEEE operations only (subtract, divide, multiply, add, square root),
so oracle, numpy and device agree bit for bit.  The equirectangular form calls sin / cos, whose last
bits depend on the math library (glibc on the host, the device library on the GPU): the GPU is
compared with the oracle to an absolute tolerance of 4 units in the last place of 1.0 (2^-50); every
component of a bearing has magnitude <= 1."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

EQUIRECT_TOL = 2.0 ** -50

KITTI = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, cols=1241, rows=376)   # example/kitti/KITTI_mono_00-02.yaml
TUM_FISHEYE = dict(fx=190.97847715128717, fy=190.9733070521226, cx=254.93170605935475, cy=256.8974428996504, cols=512, rows=512)


def keypoints(rng, n, cols, rows):
    kps = np.zeros(n, A.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(-30.0, cols + 30.0, n).astype(np.float32)    # undistorted points may leave the image
    kps["y"] = rng.uniform(-30.0, rows + 30.0, n).astype(np.float32)
    kps["size"], kps["angle"], kps["octave"] = 31.0, rng.uniform(0, 360, n), rng.integers(0, 8, n)
    if n >= 4:
        kps["x"][:4] = (0.0, cols, 0.0, cols / 2)
        kps["y"][:4] = (0.0, 0.0, rows, rows / 2)
    return kps


def camera_struct(model, cols=0, rows=0, fx=0.0, fy=0.0, cx=0.0, cy=0.0):
    c = A.Camera()
    c.model, c.cols, c.rows, c.fx, c.fy, c.cx, c.cy = model, cols, rows, fx, fy, cx, cy
    return c


def oracle_bearings(cam, kps):
    out = np.full((len(kps), 3), np.nan)
    oracle.check(oracle.lib().ovo_convert_keypoints_to_bearings(C.byref(cam), A.ptr(kps, A.kpp), len(kps), A.ptr(out, A.f64p)))
    return out


def numpy_pinhole(kps, fx, fy, cx, cy, **_):
    xn = (kps["x"].astype(np.float64) - cx) / fx
    yn = (kps["y"].astype(np.float64) - cy) / fy
    l2 = np.sqrt(xn * xn + yn * yn + 1.0)
    return np.stack([xn / l2, yn / l2, 1.0 / l2], axis=1)


def numpy_equirect(kps, cols, rows):
    # the divisions are float divisions (cols_ / rows_ are unsigned int), everything after is double
    lon = ((kps["x"] / np.float32(cols)).astype(np.float64) - 0.5) * (2 * np.pi)
    lat = -((kps["y"] / np.float32(rows)).astype(np.float64) - 0.5) * np.pi
    return np.stack([np.cos(lat) * np.sin(lon), -np.sin(lat), np.cos(lat) * np.cos(lon)], axis=1)


@pytest.mark.parametrize("model,par", [(A.CAMERA_PERSPECTIVE, KITTI), (A.CAMERA_FISHEYE, TUM_FISHEYE)])
def test_oracle_pinhole_equals_numpy_bit_for_bit(model, par):
    kps = keypoints(np.random.default_rng(11), 5000, par["cols"], par["rows"])
    got = oracle_bearings(camera_struct(model, **par), kps)
    assert np.array_equal(got.view(np.uint64), numpy_pinhole(kps, **par).view(np.uint64))
    assert np.all(np.abs(np.linalg.norm(got, axis=1) - 1.0) < 1e-15)


def test_oracle_pinhole_known_answers():
    cam = camera_struct(A.CAMERA_PERSPECTIVE, fx=100.0, fy=50.0, cx=10.0, cy=20.0)
    kps = np.zeros(3, A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = (10.0, 310.0, 10.0), (20.0, 20.0, -180.0)     # centre; x_n = 3; y_n = -4
    b = oracle_bearings(cam, kps)
    assert b[0].tolist() == [0.0, 0.0, 1.0]
    assert b[1].tolist() == [3.0 / np.sqrt(10.0), 0.0, 1.0 / np.sqrt(10.0)]
    assert b[2].tolist() == [0.0, -4.0 / np.sqrt(17.0), 1.0 / np.sqrt(17.0)]


def test_oracle_equirectangular_known_answers_and_numpy():
    cols, rows = 1920, 960
    cam = camera_struct(A.CAMERA_EQUIRECTANGULAR, cols=cols, rows=rows)
    kps = np.zeros(4, A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = (960.0, 1440.0, 960.0, 480.0), (480.0, 480.0, 0.0, 480.0)
    b = oracle_bearings(cam, kps)
    assert np.allclose(b[0], [0, 0, 1], atol=1e-15)        # image centre looks along +z
    assert np.allclose(b[1], [1, 0, 0], atol=1e-15)        # a quarter turn to the right: +x
    assert np.allclose(b[2], [0, -1, 0], atol=1e-15)       # top row: -y (up)
    assert np.allclose(b[3], [-1, 0, 0], atol=1e-15)
    kps = keypoints(np.random.default_rng(12), 5000, cols, rows)
    got = oracle_bearings(cam, kps)
    assert np.max(np.abs(got - numpy_equirect(kps, cols, rows))) <= EQUIRECT_TOL
    assert np.all(np.abs(np.linalg.norm(got, axis=1) - 1.0) < 1e-15)


def test_oracle_rejects_an_unknown_model():
    kps = np.zeros(1, A.KEYPOINT_DTYPE)
    out = np.zeros((1, 3))
    cam = camera_struct(7, fx=1.0, fy=1.0)
    assert oracle.lib().ovo_convert_keypoints_to_bearings(C.byref(cam), A.ptr(kps, A.kpp), 1, A.ptr(out, A.f64p)) == A.OVS_ERR_INVALID_ARGUMENT


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 2000, 100003])
@pytest.mark.parametrize("kind", ["perspective", "fisheye", "equirectangular"])
def test_gpu_matches_oracle(kind, n):
    from openvslam_amd import camera
    if kind == "equirectangular":
        cam, cols, rows = camera.equirectangular(1920, 960), 1920, 960
    else:
        par = KITTI if kind == "perspective" else TUM_FISHEYE
        cam, cols, rows = getattr(camera, kind)(**par), par["cols"], par["rows"]
    kps = keypoints(np.random.default_rng(n + 1), n, cols, rows)
    got = cam.convert_keypoints_to_bearings(kps)
    want = oracle_bearings(cam._cam, kps)
    assert got.shape == (n, 3)
    if kind == "equirectangular":
        assert n == 0 or np.max(np.abs(got - want)) <= EQUIRECT_TOL
    else:
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
@pytest.mark.parametrize("with_counts", [False, True])
def test_gpu_batch_on_device_arrays(with_counts):
    """n_frames x cap keypoints resident on the device; entries beyond a frame's count stay untouched."""
    import torch
    from openvslam_amd import camera
    dev = torch.device("cuda", 0)
    n_frames, cap = 37, 2500
    rng = np.random.default_rng(3)
    kps = keypoints(rng, n_frames * cap, KITTI["cols"], KITTI["rows"])
    counts = rng.integers(0, cap + 1, n_frames).astype(np.int32)
    counts[0], counts[1], counts[2] = 0, cap, cap + 77       # a count above the capacity is clamped
    cam = camera.perspective(**KITTI)
    d_kps = torch.from_numpy(kps.view(np.uint8)).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_out = torch.full((n_frames, cap, 3), -7.0, dtype=torch.float64, device=dev)
    cam.convert_keypoints_to_bearings_dev(d_kps.data_ptr(), d_counts.data_ptr() if with_counts else None, n_frames, cap,
                                          d_out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    got = d_out.cpu().numpy()
    want = oracle_bearings(cam._cam, kps).reshape(n_frames, cap, 3)
    for f in range(n_frames):
        n = min(int(counts[f]), cap) if with_counts else cap
        assert np.array_equal(got[f, :n].view(np.uint64), want[f, :n].view(np.uint64)), f
        assert np.all(got[f, n:] == -7.0), f


@pytest.mark.gpu
def test_gpu_rejects_bad_cameras():
    from openvslam_amd import _lib, camera
    kps = np.zeros(4, A.KEYPOINT_DTYPE)
    with pytest.raises(_lib.OvsError):
        camera.perspective(0.0, 1.0, 0.0, 0.0).convert_keypoints_to_bearings(kps)
    with pytest.raises(_lib.OvsError):
        camera.equirectangular(0, 960).convert_keypoints_to_bearings(kps)
    bad = camera.perspective(1.0, 1.0, 0.0, 0.0)
    bad._cam.model = 9
    with pytest.raises(_lib.OvsError):
        bad.convert_keypoints_to_bearings(kps)
