"""camera::{perspective,fisheye,equirectangular}::undistort_keypoints
(src/openvslam/camera/perspective.cc:99-128, fisheye.cc:139-167, equirectangular.cc:41-43).

The arithmetic is OpenCV's (cv::undistortPoints with the termination criterion 20 / 1e-6,
cv::fisheye::undistortPoints), which the reference tree does not contain and no reference test
pins: parity unpinned.  The oracle restates OpenCV 4.1.0; here it is checked (i) against a numpy
restatement of the same iterations, (ii) by the property that defines the function -- undistorting
a point that the forward distortion model produced returns the ideal point -- and (iii) on the
zero-distortion camera of the KITTI configuration.  The GPU is compared with the oracle: bit for bit
for the perspective model (correctly rounded double operations only), to one float ulp for the
fisheye model (its tan() comes from glibc on the host and from the device library on the GPU)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

# example/kitti/KITTI_mono_00-02.yaml, example/euroc/EuRoC_mono.yaml, example/tum_vi/TUM_VI_mono.yaml
KITTI = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, cols=1241, rows=376)
EUROC = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375, k1=-0.28340811, k2=0.07395907, p1=0.00019359, p2=1.76187114e-05, k3=0.0,
             cols=752, rows=480)
STRONG = dict(fx=300.0, fy=310.0, cx=320.0, cy=240.0, k1=-0.5, k2=0.0, p1=0.001, p2=-0.002, k3=0.0, cols=640, rows=480)
TUM_VI = dict(fx=190.97847715128717, fy=190.9733070521226, cx=254.93170605935475, cy=256.8974428996504,
              k1=0.0034823894022493434, k2=0.0007150348452162257, k3=-0.0020532361418706202, k4=0.00020293673591811182, cols=512, rows=512)


def cam_struct(model, par):
    c = A.Camera()
    c.model = model
    c.cols, c.rows = par.get("cols", 0), par.get("rows", 0)
    c.fx, c.fy, c.cx, c.cy = par["fx"], par["fy"], par["cx"], par["cy"]
    names = ("k1", "k2", "p1", "p2", "k3") if model == A.CAMERA_PERSPECTIVE else ("k1", "k2", "k3", "k4")
    for i, k in enumerate(names):
        c.dist[i] = par.get(k, 0.0)
    return c


def keypoints(rng, n, cols, rows, pad=0.0):
    kps = np.zeros(n, A.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(-pad, cols + pad, n).astype(np.float32)
    kps["y"] = rng.uniform(-pad, rows + pad, n).astype(np.float32)
    kps["size"] = rng.uniform(7, 100, n)
    kps["angle"] = rng.uniform(0, 360, n)
    kps["response"] = rng.uniform(7, 255, n)
    kps["octave"] = rng.integers(0, 8, n)
    kps["class_id"] = rng.integers(0, 100, n)
    if n >= 4:   # the image corners, which compute_image_bounds undistorts (perspective.cc:82-95)
        kps["x"][:4] = (0.0, cols, 0.0, cols)
        kps["y"][:4] = (0.0, 0.0, rows, rows)
    return kps


def oracle_undistort(cam, kps):
    out = np.zeros(len(kps), A.KEYPOINT_DTYPE)
    oracle.check(oracle.lib().ovo_undistort_keypoints(C.byref(cam), A.ptr(kps, A.kpp), len(kps), A.ptr(out, A.kpp)))
    return out


def f32(v):
    return float(np.float32(v))


def numpy_perspective(kps, par):
    """cvUndistortPointsInternal, vectorised: every element runs its own number of iterations."""
    fx, fy, cx, cy = f32(par["fx"]), f32(par["fy"]), f32(par["cx"]), f32(par["cy"])
    k0, k1, k2, k3, k4 = (f32(par.get(n, 0.0)) for n in ("k1", "k2", "p1", "p2", "k3"))
    ifx, ify = 1.0 / fx, 1.0 / fy
    u, v = kps["x"].astype(np.float64), kps["y"].astype(np.float64)
    x0, y0 = (u - cx) * ifx + 0.0, (v - cy) * ify + 0.0    # the identity tilt matrix turns -0 into +0
    x, y = x0.copy(), y0.copy()
    error = np.full(len(u), np.finfo(np.float64).max)
    active = np.ones(len(u), bool)
    with np.errstate(all="ignore"):
        for _ in range(20):
            active &= ~(error < 1e-6)
            r2 = x * x + y * y
            icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2)
            neg = active & (icdist < 0)
            x[neg], y[neg] = ((u - cx) * ifx)[neg], ((v - cy) * ify)[neg]
            active &= ~neg
            dx = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2
            dy = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + 0 * r2 + 0 * r2 * r2
            xn, yn = (x0 - dx) * icdist, (y0 - dy) * icdist
            x, y = np.where(active, xn, x), np.where(active, yn, y)
            r2 = x * x + y * y
            r4 = r2 * r2
            r6 = r4 * r2
            a1, a2, a3 = 2 * x * y, r2 + 2 * x * x, r2 + 2 * y * y
            cdist = 1 + k0 * r2 + k1 * r4 + k4 * r6
            icdist2 = 1.0 / (1 + 0 * r2 + 0 * r4 + 0 * r6)
            xd = x * cdist * icdist2 + k2 * a1 + k3 * a2 + 0 * r2 + 0 * r4
            yd = y * cdist * icdist2 + k2 * a3 + k3 * a1 + 0 * r2 + 0 * r4
            ex, ey = (xd * fx + cx) - u, (yd * fy + cy) - v
            error = np.where(active, np.sqrt(ex * ex + ey * ey), error)
    xx = fx * x + 0 * y + cx
    yy = 0 * x + fy * y + cy
    ww = 1.0 / (0 * x + 0 * y + 1)
    return (xx * ww).astype(np.float32), (yy * ww).astype(np.float32)


def distort_perspective(xn, yn, par):
    """The forward model of cv::projectPoints on normalised coordinates -> pixels."""
    k1, k2, p1, p2, k3 = (f32(par.get(n, 0.0)) for n in ("k1", "k2", "p1", "p2", "k3"))
    r2 = xn * xn + yn * yn
    radial = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
    xd = xn * radial + 2 * p1 * xn * yn + p2 * (r2 + 2 * xn * xn)
    yd = yn * radial + p1 * (r2 + 2 * yn * yn) + 2 * p2 * xn * yn
    return xd * f32(par["fx"]) + f32(par["cx"]), yd * f32(par["fy"]) + f32(par["cy"])


def distort_fisheye(xn, yn, par):
    k = [f32(par[n]) for n in ("k1", "k2", "k3", "k4")]
    r = np.sqrt(xn * xn + yn * yn)
    theta = np.arctan(r)
    theta_d = theta * (1 + k[0] * theta ** 2 + k[1] * theta ** 4 + k[2] * theta ** 6 + k[3] * theta ** 8)
    s = np.where(r > 1e-12, theta_d / np.maximum(r, 1e-300), 1.0)
    return xn * s * f32(par["fx"]) + f32(par["cx"]), yn * s * f32(par["fy"]) + f32(par["cy"])


def float_ulps(a, b):
    """Distance in representable floats between two float32 arrays of equal sign."""
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_oracle_copies_the_fields_the_reference_copies():
    kps = keypoints(np.random.default_rng(1), 100, 752, 480)
    for model, par in ((A.CAMERA_PERSPECTIVE, EUROC), (A.CAMERA_FISHEYE, TUM_VI)):
        out = oracle_undistort(cam_struct(model, par), kps)
        for f in ("angle", "size", "octave"):
            assert np.array_equal(out[f], kps[f])
        assert np.all(out["response"] == 0.0) and np.all(out["class_id"] == -1)     # a default cv::KeyPoint
    eq = A.Camera()
    eq.model, eq.cols, eq.rows = A.CAMERA_EQUIRECTANGULAR, 1920, 960
    assert oracle_undistort(eq, kps).tobytes() == kps.tobytes()                     # undist_keypts = dist_keypts


def test_oracle_zero_distortion_returns_the_input_to_the_last_float():
    """KITTI: all coefficients are 0, cv::undistortPoints still runs (no early-out) and returns
    fx * ((u - cx) / fx) + cx rounded to float: the input, or its neighbour."""
    kps = keypoints(np.random.default_rng(2), 20000, KITTI["cols"], KITTI["rows"])
    out = oracle_undistort(cam_struct(A.CAMERA_PERSPECTIVE, KITTI), kps)
    ulps = np.maximum(float_ulps(out["x"] + np.float32(1), kps["x"] + np.float32(1)), float_ulps(out["y"] + np.float32(1), kps["y"] + np.float32(1)))
    assert ulps.max() <= 1
    assert np.mean(ulps == 0) > 0.99


@pytest.mark.parametrize("par", [KITTI, EUROC, STRONG], ids=["kitti", "euroc", "strong"])
def test_oracle_perspective_equals_numpy_restatement(par):
    kps = keypoints(np.random.default_rng(3), 20000, par["cols"], par["rows"], pad=300.0 if par is STRONG else 20.0)
    out = oracle_undistort(cam_struct(A.CAMERA_PERSPECTIVE, par), kps)
    wx, wy = numpy_perspective(kps, par)
    assert np.array_equal(out["x"].view(np.uint32), wx.view(np.uint32))
    assert np.array_equal(out["y"].view(np.uint32), wy.view(np.uint32))


def test_strong_distortion_reaches_the_negative_icdist_exit():
    kps = keypoints(np.random.default_rng(3), 20000, STRONG["cols"], STRONG["rows"], pad=300.0)
    xn = (kps["x"].astype(np.float64) - STRONG["cx"]) / STRONG["fx"]
    yn = (kps["y"].astype(np.float64) - STRONG["cy"]) / STRONG["fy"]
    assert np.any(1 + f32(STRONG["k1"]) * (xn * xn + yn * yn) < 0)


@pytest.mark.parametrize("par", [EUROC, STRONG], ids=["euroc", "strong"])
def test_oracle_perspective_inverts_the_forward_model(par):
    rng = np.random.default_rng(4)
    lim = 0.75 if par is EUROC else 0.45      # inside the region where the model is monotonic
    xn, yn = rng.uniform(-lim, lim, 5000), rng.uniform(-lim * 0.7, lim * 0.7, 5000)
    u, v = distort_perspective(xn, yn, par)
    kps = np.zeros(len(u), A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = u.astype(np.float32), v.astype(np.float32)
    out = oracle_undistort(cam_struct(A.CAMERA_PERSPECTIVE, par), kps)
    ideal_x, ideal_y = xn * f32(par["fx"]) + f32(par["cx"]), yn * f32(par["fy"]) + f32(par["cy"])
    # float input (2^-15 px at 750) amplified by the inverse model, termination at 1e-6 px, float output
    assert np.max(np.abs(out["x"] - ideal_x)) < 2e-3 and np.max(np.abs(out["y"] - ideal_y)) < 2e-3


def test_oracle_fisheye_inverts_the_forward_model():
    rng = np.random.default_rng(5)
    xn, yn = rng.uniform(-1.5, 1.5, 5000), rng.uniform(-1.5, 1.5, 5000)
    xn[0], yn[0] = 0.0, 0.0       # the principal point: theta_d = 0, scale stays 1
    u, v = distort_fisheye(xn, yn, TUM_VI)
    kps = np.zeros(len(u), A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = u.astype(np.float32), v.astype(np.float32)
    out = oracle_undistort(cam_struct(A.CAMERA_FISHEYE, TUM_VI), kps)
    ideal_x, ideal_y = xn * f32(TUM_VI["fx"]) + f32(TUM_VI["cx"]), yn * f32(TUM_VI["fy"]) + f32(TUM_VI["cy"])
    assert np.max(np.abs(out["x"] - ideal_x)) < 5e-3 and np.max(np.abs(out["y"] - ideal_y)) < 5e-3
    assert out["x"][0] == np.float32(f32(TUM_VI["cx"])) and out["y"][0] == np.float32(f32(TUM_VI["cy"]))


def test_oracle_fisheye_clips_beyond_a_half_turn():
    """theta_d is clipped to pi / 2: a point far outside the image circle still yields a finite result."""
    kps = np.zeros(1, A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = 5000.0, 256.0
    out = oracle_undistort(cam_struct(A.CAMERA_FISHEYE, TUM_VI), kps)
    assert np.isfinite(out["x"][0]) and np.isfinite(out["y"][0])


def _scalar_radial_undistort(u, fx, cx, k1):
    """cv::undistortPoints for a point on the x axis of a camera with one radial coefficient, as the
    scalar recurrence the OpenCV loop reduces to (all tangential / rational / prism terms are 0 and y
    stays 0):  x <- x0 / (1 + k1 x^2)  until the reprojection error |fx x (1 + k1 x^2) + cx - u| drops
    below 1e-6 (checked before each iteration) or 20 iterations have run.  Plain Python doubles; every
    omitted term is an exact + 0 or * 1.  Returns (pixel as float32, iterations run)."""
    fx, cx, k1 = f32(fx), f32(cx), f32(k1)
    x0 = (u - cx) * (1.0 / fx)
    x, err, it = x0, float("inf"), 0
    while it < 20 and not err < 1e-6:
        r2 = x * x
        icdist = 1.0 / (1.0 + k1 * r2)
        assert icdist > 0
        x = x0 * icdist
        r2 = x * x
        xd = x * (1.0 + k1 * r2)
        e = (xd * fx + cx) - u
        err = (e * e) ** 0.5
        it += 1
    return np.float32(fx * x + cx), it


@pytest.mark.parametrize("u,k1,stops_on", [(400.0, -0.05, "eps"), (530.0, -0.3, "count")])
def test_oracle_iteration_stops_on_eps_or_on_count(u, k1, stops_on):
    """Known answers of the termination criterion (perspective.cc:115-116: COUNT 20 + EPS 1e-6): a mild
    distortion converges in a few iterations and stops on EPS; a strong one near the fold of the model
    is still moving after 20 iterations and stops on COUNT.  The recurrence above is evaluated by hand
    (scalar code, not the vectorised restatement)."""
    par = dict(fx=300.0, fy=300.0, cx=320.0, cy=240.0, k1=k1, cols=640, rows=480)
    want, iterations = _scalar_radial_undistort(u, par["fx"], par["cx"], k1)
    assert (iterations < 20) == (stops_on == "eps") and iterations >= 3
    kps = np.zeros(1, A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = u, par["cy"]
    out = oracle_undistort(cam_struct(A.CAMERA_PERSPECTIVE, par), kps)
    assert out["x"][0] == want and out["y"][0] == np.float32(par["cy"])
    if stops_on == "count":
        # one more iteration would have changed the float result: the count really is what stopped it
        fx, cx = f32(par["fx"]), f32(par["cx"])
        x0 = (u - cx) * (1.0 / fx)
        x = (float(want) - cx) / fx
        assert np.float32(fx * (x0 / (1.0 + f32(k1) * x * x)) + cx) != want


def test_oracle_fisheye_clamp_known_answer():
    """cv::fisheye::undistortPoints clips theta_d to pi / 2.  With all four coefficients 0 the Newton
    loop leaves theta = theta_d (its first correction is exactly 0), so a pixel whose normalised radius
    exceeds pi / 2 is scaled by tan(pi / 2) / (pi / 2) whatever its distance: the same huge factor for a
    point just outside the limit and for one ten times farther out."""
    import math
    par = dict(fx=200.0, fy=200.0, cx=256.0, cy=256.0, k1=0.0, k2=0.0, k3=0.0, k4=0.0, cols=512, rows=512)
    scale = math.tan(math.pi / 2) / (math.pi / 2)
    kps = np.zeros(3, A.KEYPOINT_DTYPE)
    kps["x"] = (256.0 + 200.0 * 1.6, 256.0 + 200.0 * 16.0, 256.0 + 200.0 * 1.5)   # radius 1.6, 16 and (inside) 1.5
    kps["y"] = 256.0
    out = oracle_undistort(cam_struct(A.CAMERA_FISHEYE, par), kps)
    for i, r in ((0, 1.6), (1, 16.0)):
        assert out["x"][i] == np.float32(200.0 * (r * scale) + 256.0) and out["y"][i] == np.float32(256.0)
    # inside the limit nothing is clipped: x = fx tan(1.5) + cx
    assert out["x"][2] == np.float32(200.0 * (1.5 * (math.tan(1.5) / 1.5)) + 256.0)


GPU_CASES = [("perspective", KITTI, 20.0), ("perspective", EUROC, 20.0), ("perspective", STRONG, 300.0), ("fisheye", TUM_VI, 50.0)]


def make_camera(kind, par, device=0):
    from openvslam_amd import camera
    return getattr(camera, kind)(device=device, **par)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 257, 2000, 100003])
@pytest.mark.parametrize("kind,par,pad", GPU_CASES, ids=["kitti", "euroc", "strong", "tum_vi"])
def test_gpu_matches_oracle(kind, par, pad, n):
    cam = make_camera(kind, par)
    kps = keypoints(np.random.default_rng(n + 7), n, par["cols"], par["rows"], pad)
    got = cam.undistort_keypoints(kps)
    want = oracle_undistort(cam._cam, kps)
    assert got.shape == want.shape
    if kind == "perspective":
        assert got.tobytes() == want.tobytes()
    else:
        for f in ("size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(got[f], want[f])
        bias = np.float32(4096)       # compare away from zero so that neighbours differ by one step
        ulps = np.maximum(float_ulps(got["x"] + bias, want["x"] + bias), float_ulps(got["y"] + bias, want["y"] + bias))
        assert n == 0 or (ulps.max() <= 1 and np.mean(ulps == 0) >= 0.999)


@pytest.mark.gpu
def test_gpu_equirectangular_is_a_copy():
    from openvslam_amd import camera
    kps = keypoints(np.random.default_rng(9), 3000, 1920, 960)
    assert camera.equirectangular(1920, 960).undistort_keypoints(kps).tobytes() == kps.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("with_counts", [False, True])
def test_gpu_batch_with_bearings_in_the_same_launch(with_counts):
    """n_frames x cap keypoints resident on the device -> undistorted keypoints and their bearings;
    entries beyond a frame's count stay untouched in both outputs."""
    import torch
    dev = torch.device("cuda", 0)
    n_frames, cap = 23, 2300
    rng = np.random.default_rng(10)
    kps = keypoints(rng, n_frames * cap, EUROC["cols"], EUROC["rows"], 20.0)
    counts = rng.integers(0, cap + 1, n_frames).astype(np.int32)
    counts[0], counts[1], counts[2] = 0, cap, cap + 5
    cam = make_camera("perspective", EUROC)
    d_kps = torch.from_numpy(kps.view(np.uint8)).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    d_out = torch.full((n_frames * cap * 28,), 0xAB, dtype=torch.uint8, device=dev)
    d_bear = torch.full((n_frames, cap, 3), -7.0, dtype=torch.float64, device=dev)
    cam.undistort_keypoints_dev(d_kps.data_ptr(), d_counts.data_ptr() if with_counts else None, n_frames, cap, d_out.data_ptr(),
                                d_bear.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    got = d_out.cpu().numpy().view(A.KEYPOINT_DTYPE).reshape(n_frames, cap)
    got_b = d_bear.cpu().numpy()
    want = oracle_undistort(cam._cam, kps).reshape(n_frames, cap)
    want_b = np.zeros((n_frames * cap, 3))
    oracle.check(oracle.lib().ovo_convert_keypoints_to_bearings(C.byref(cam._cam), A.ptr(np.ascontiguousarray(want.reshape(-1)), A.kpp),
                                                                n_frames * cap, A.ptr(want_b, A.f64p)))
    want_b = want_b.reshape(n_frames, cap, 3)
    for f in range(n_frames):
        n = min(int(counts[f]), cap) if with_counts else cap
        assert got[f, :n].tobytes() == want[f, :n].tobytes(), f
        assert np.array_equal(got_b[f, :n].view(np.uint64), want_b[f, :n].view(np.uint64)), f
        assert np.all(got[f, n:].view(np.uint8) == 0xAB) and np.all(got_b[f, n:] == -7.0), f


@pytest.mark.gpu
def test_gpu_full_size_batch_properties():
    """256 frames x 2250 keypoints (the extractor's batch): properties that need no oracle run --
    bearings have unit length, pushing the undistorted points through the forward model returns
    the input -- plus an oracle comparison of a 5000-keypoint sample."""
    import torch
    dev = torch.device("cuda", 0)
    n_frames, cap = 256, 2250
    rng = np.random.default_rng(12)
    kps = keypoints(rng, n_frames * cap, EUROC["cols"], EUROC["rows"])
    cam = make_camera("perspective", EUROC)
    d_kps = torch.from_numpy(kps.view(np.uint8)).to(dev)
    d_out = torch.zeros_like(d_kps)
    d_bear = torch.zeros((n_frames * cap, 3), dtype=torch.float64, device=dev)
    cam.undistort_keypoints_dev(d_kps.data_ptr(), None, n_frames, cap, d_out.data_ptr(), d_bear.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    got = d_out.cpu().numpy().view(A.KEYPOINT_DTYPE)
    bear = d_bear.cpu().numpy()
    assert np.max(np.abs(np.linalg.norm(bear, axis=1) - 1.0)) < 1e-15
    xn = (got["x"].astype(np.float64) - f32(EUROC["cx"])) / f32(EUROC["fx"])
    yn = (got["y"].astype(np.float64) - f32(EUROC["cy"])) / f32(EUROC["fy"])
    u, v = distort_perspective(xn, yn, EUROC)
    assert np.max(np.abs(u - kps["x"])) < 2e-3 and np.max(np.abs(v - kps["y"])) < 2e-3
    # the bearing is the normalised (x_n, y_n, 1) of the undistorted point (unnarrowed camera parameters)
    xb = (got["x"].astype(np.float64) - EUROC["cx"]) / EUROC["fx"]
    assert np.allclose(bear[:, 0] / bear[:, 2], xb, rtol=0, atol=1e-14)
    pick = rng.choice(len(kps), 5000, replace=False)
    sample = np.ascontiguousarray(kps[pick])
    assert got[pick].tobytes() == oracle_undistort(cam._cam, sample).tobytes()
