// ovs_camera.hip -- keypoints to bearing vectors on the device, the step of data::frame
// construction between keypoint undistortion and grid assignment (src/openvslam/data/frame.cc:36-51).
// Replaces camera::perspective::convert_keypoints_to_bearings (camera/perspective.cc:130-138),
// camera::fisheye::convert_keypoints_to_bearings (camera/fisheye.cc:169-177) and
// camera::equirectangular::convert_keypoints_to_bearings (camera/equirectangular.cc:45-56).
//
// One thread per keypoint, 28 bytes in and 24 bytes out: a streaming kernel.  Double precision in
// the reference's operation order without contraction; the perspective / fisheye form is made of
// correctly rounded operations only (subtract, divide, multiply, add, square root) and is
// reproduced bit for bit, the equirectangular form calls sin / cos, whose last bit depends on the
// math library.
//
// Keypoint undistortion (camera::perspective::undistort_keypoints, camera/perspective.cc:99-128;
// camera::fisheye::undistort_keypoints, camera/fisheye.cc:139-167; camera::equirectangular::
// undistort_keypoints, camera/equirectangular.cc:41-43) is the same shape: one thread per keypoint,
// 28 bytes in, 28 out (+ 24 when the bearing is written by the same launch).  The iterations of
// cv::undistortPoints (termination criterion 20 iterations / 1e-6) and of
// cv::fisheye::undistortPoints (ten Newton steps) follow OpenCV 4.1.0 in double precision, operation
// by operation; the perspective form uses correctly rounded operations only, the fisheye form ends in
// tan().
#include <cfloat>

#include "ovs_match_common.h"

using namespace ovsm;

namespace {

// the accumulation order of cv::Matx33d * cv::Vec3d: s = 0; s += a(i,k) * b(k)
__device__ __forceinline__ double matx_row(double a0, double a1, double a2, double b0, double b1, double b2) {
    double s = 0.0;
    s += a0 * b0;
    s += a1 * b1;
    s += a2 * b2;
    return s;
}

// Camera parameters as OpenCV sees them: the reference stores the camera matrix and the distortion
// parameters as float matrices (perspective.cc:21-22, fisheye.cc:21-22), OpenCV widens them again.
struct undist_par {
    int model;
    double fx, fy, cx, cy, ifx, ify;
    double k[5];
};

// cvUndistortPointsInternal with distortion coefficients, R empty, P = camera matrix,
// criteria COUNT | EPS (20, 1e-6).  The coefficients beyond the fifth are zero; their terms stay in
// the expressions so that every intermediate (signed zeros included) is the one OpenCV computes.
__device__ void undistort_perspective(const undist_par& c, float px, float py, float& ox, float& oy) {
    const double k0 = c.k[0], k1 = c.k[1], k2 = c.k[2], k3 = c.k[3], k4 = c.k[4];
    const double k5 = 0, k6 = 0, k7 = 0, k8 = 0, k9 = 0, k10 = 0, k11 = 0;
    double x = px, y = py;
    const double u = x, v = y;
    x = (x - c.cx) * c.ifx;
    y = (y - c.cy) * c.ify;
    const double untilt2 = matx_row(0, 0, 1, x, y, 1);
    double inv_proj = untilt2 ? 1. / untilt2 : 1;
    const double ux = matx_row(1, 0, 0, x, y, 1), uy = matx_row(0, 1, 0, x, y, 1);
    x = inv_proj * ux;
    y = inv_proj * uy;
    const double x0 = x, y0 = y;
    double error = DBL_MAX;
    for (int j = 0; j < 20; ++j) {
        if (error < 1e-6) {
            break;
        }
        double r2 = x * x + y * y;
        const double icdist = (1 + ((k7 * r2 + k6) * r2 + k5) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        if (icdist < 0) {
            x = (u - c.cx) * c.ifx;
            y = (v - c.cy) * c.ify;
            break;
        }
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + k8 * r2 + k9 * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + k10 * r2 + k11 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
        r2 = x * x + y * y;
        const double r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k0 * r2 + k1 * r4 + k4 * r6;
        const double icdist2 = 1. / (1 + k5 * r2 + k6 * r4 + k7 * r6);
        const double xd0 = x * cdist * icdist2 + k2 * a1 + k3 * a2 + k8 * r2 + k9 * r4;
        const double yd0 = y * cdist * icdist2 + k2 * a3 + k3 * a1 + k10 * r2 + k11 * r4;
        const double t2 = matx_row(0, 0, 1, xd0, yd0, 1);
        inv_proj = t2 ? 1. / t2 : 1;
        const double xd = inv_proj * matx_row(1, 0, 0, xd0, yd0, 1);
        const double yd = inv_proj * matx_row(0, 1, 0, xd0, yd0, 1);
        const double ex = (xd * c.fx + c.cx) - u, ey = (yd * c.fy + c.cy) - v;
        error = __dsqrt_rn(ex * ex + ey * ey);
    }
    // RR = P * identity = P
    const double xx = c.fx * x + 0.0 * y + c.cx;
    const double yy = 0.0 * x + c.fy * y + c.cy;
    const double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    ox = (float)(xx * ww);
    oy = (float)(yy * ww);
}

// cv::fisheye::undistortPoints, R empty, P = K
__device__ void undistort_fisheye(const undist_par& c, float px, float py, float& ox, float& oy) {
    const double kPi = 3.1415926535897932384626433832795;
    const double pw0 = ((double)px - c.cx) / c.fx, pw1 = ((double)py - c.cy) / c.fy;
    double scale = 1.0;
    double theta_d = __dsqrt_rn(pw0 * pw0 + pw1 * pw1);
    theta_d = fmin(fmax(-kPi / 2., theta_d), kPi / 2.);
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int j = 0; j < 10; ++j) {
            const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
            const double k0_theta2 = c.k[0] * theta2, k1_theta4 = c.k[1] * theta4, k2_theta6 = c.k[2] * theta6, k3_theta8 = c.k[3] * theta8;
            const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d)
                                     / (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabs(theta_fix) < 1e-8) {
                break;
            }
        }
        scale = tan(theta) / theta_d;
    }
    const double pu0 = pw0 * scale, pu1 = pw1 * scale;
    const double pr0 = matx_row(c.fx, 0, c.cx, pu0, pu1, 1.0), pr1 = matx_row(0, c.fy, c.cy, pu0, pu1, 1.0), pr2 = matx_row(0, 0, 1, pu0, pu1, 1.0);
    ox = (float)(pr0 / pr2);
    oy = (float)(pr1 / pr2);
}

__device__ __forceinline__ void bearing_of(const ovs_camera& cam, float px, float py, double* __restrict__ b) {
    if (cam.model == OVS_CAMERA_EQUIRECTANGULAR) {
        // pt.x / cols_ is a float division (cols_ is unsigned int), the rest is double
        const double kPi = 3.14159265358979323846;
        const float fx = __fdiv_rn(px, (float)cam.cols), fy = __fdiv_rn(py, (float)cam.rows);
        const double lon = __dmul_rn(__dsub_rn((double)fx, 0.5), __dmul_rn(2.0, kPi));
        const double lat = __dmul_rn(-__dsub_rn((double)fy, 0.5), kPi);
        const double cl = cos(lat), sl = sin(lat);
        b[0] = __dmul_rn(cl, sin(lon));
        b[1] = -sl;
        b[2] = __dmul_rn(cl, cos(lon));
    }
    else {
        const double xn = __ddiv_rn(__dsub_rn((double)px, cam.cx), cam.fx);
        const double yn = __ddiv_rn(__dsub_rn((double)py, cam.cy), cam.fy);
        const double l2 = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(xn, xn), __dmul_rn(yn, yn)), 1.0));
        b[0] = __ddiv_rn(xn, l2);
        b[1] = __ddiv_rn(yn, l2);
        b[2] = __ddiv_rn(1.0, l2);
    }
}

__global__ __launch_bounds__(256) void k_undistort(ovs_camera cam, undist_par par, const ovs_keypoint* __restrict__ kps, const int32_t* __restrict__ counts,
                                                   int cap, ovs_keypoint* __restrict__ out, double* __restrict__ bearings) {
    const int frame = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = counts ? min(max(counts[frame], 0), cap) : cap;
    if (i >= n) {
        return;
    }
    const size_t at = (size_t)frame * cap + i;
    ovs_keypoint kp = kps[at];
    if (cam.model != OVS_CAMERA_EQUIRECTANGULAR) {
        float ox, oy;
        if (cam.model == OVS_CAMERA_PERSPECTIVE) {
            undistort_perspective(par, kp.x, kp.y, ox, oy);
        }
        else {
            undistort_fisheye(par, kp.x, kp.y, ox, oy);
        }
        kp.x = ox;
        kp.y = oy;
        kp.response = 0.0f; // the defaults of cv::KeyPoint: undist_keypts.resize(n)
        kp.class_id = -1;
    }
    out[at] = kp;
    if (bearings) {
        bearing_of(cam, kp.x, kp.y, bearings + 3 * at);
    }
}

__global__ __launch_bounds__(256) void k_bearings(ovs_camera cam, const ovs_keypoint* __restrict__ kps, const int32_t* __restrict__ counts,
                                                  int cap, double* __restrict__ out) {
    const int frame = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = counts ? min(max(counts[frame], 0), cap) : cap;
    if (i >= n) {
        return;
    }
    const size_t at = (size_t)frame * cap + i;
    bearing_of(cam, kps[at].x, kps[at].y, out + 3 * at);
}

int check_camera(const ovs_camera* cam) {
    if (!cam) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "null camera");
    }
    switch (cam->model) {
        case OVS_CAMERA_PERSPECTIVE:
        case OVS_CAMERA_FISHEYE:
            if (cam->fx == 0.0 || cam->fy == 0.0) {
                return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "focal length 0");
            }
            return OVS_OK;
        case OVS_CAMERA_EQUIRECTANGULAR:
            if (cam->cols == 0 || cam->rows == 0) {
                return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "equirectangular camera without an image size");
            }
            return OVS_OK;
        default:
            return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "unknown camera model");
    }
}

} // namespace

extern "C" {

int ovs_convert_keypoints_to_bearings_dev(const ovs_camera* cam, const ovs_keypoint* d_undist_keypts, const int32_t* d_counts,
                                          int n_frames, int cap, double* d_bearings, int device, void* stream_) {
    int rc = check_camera(cam);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n_frames < 0 || cap < 0 || n_frames > 65535 || ((n_frames > 0 && cap > 0) && (!d_undist_keypts || !d_bearings))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if ((rc = check_device(device)) != OVS_OK) {
        return rc;
    }
    if (n_frames == 0 || cap == 0) {
        return OVS_OK;
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    OVS_LAUNCH("k_bearings", k_bearings, dim3((cap + 255) / 256, n_frames), dim3(256), 0, st, *cam, d_undist_keypts, d_counts, cap, d_bearings);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

int ovs_convert_keypoints_to_bearings(const ovs_camera* cam, const ovs_keypoint* undist_keypts, int n, double* bearings, int device) {
    int rc = check_camera(cam);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n < 0 || (n > 0 && (!undist_keypts || !bearings))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if ((rc = check_device(device)) != OVS_OK) {
        return rc;
    }
    if (n == 0) {
        return OVS_OK; // bearings.resize(0)
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) return rc;
    const size_t nn = n;
    if ((rc = ctx->arena.reserve(al(nn * sizeof(ovs_keypoint)) + al(nn * 3 * sizeof(double)))) != OVS_OK) return rc;
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_kps = cv.take<ovs_keypoint>(nn);
    auto* d_out = cv.take<double>(3 * nn);
    OVS_HIP(hipMemcpyAsync(d_kps, undist_keypts, nn * sizeof(ovs_keypoint), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = ovs_convert_keypoints_to_bearings_dev(cam, d_kps, nullptr, 1, n, d_out, device, ctx->stream)) != OVS_OK) return rc;
    OVS_HIP(hipMemcpyAsync(bearings, d_out, nn * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    OVS_HIP(hipStreamSynchronize(ctx->stream));
    return OVS_OK;
}

int ovs_undistort_keypoints_dev(const ovs_camera* cam, const ovs_keypoint* d_dist_keypts, const int32_t* d_counts, int n_frames, int cap,
                                ovs_keypoint* d_undist_keypts, double* d_bearings, int device, void* stream_) {
    int rc = check_camera(cam);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n_frames < 0 || cap < 0 || n_frames > 65535 || ((n_frames > 0 && cap > 0) && (!d_dist_keypts || !d_undist_keypts))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if ((rc = check_device(device)) != OVS_OK) {
        return rc;
    }
    if (n_frames == 0 || cap == 0) {
        return OVS_OK;
    }
    undist_par par{};
    par.model = cam->model;
    par.fx = (float)cam->fx;
    par.fy = (float)cam->fy;
    par.cx = (float)cam->cx;
    par.cy = (float)cam->cy;
    par.ifx = 1. / par.fx;
    par.ify = 1. / par.fy;
    for (int i = 0; i < 5; ++i) {
        par.k[i] = (float)cam->dist[i];
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    OVS_LAUNCH("k_undistort", k_undistort, dim3((cap + 255) / 256, n_frames), dim3(256), 0, st, *cam, par, d_dist_keypts, d_counts, cap, d_undist_keypts,
               d_bearings);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}

int ovs_undistort_keypoints(const ovs_camera* cam, const ovs_keypoint* dist_keypts, int n, ovs_keypoint* undist_keypts, int device) {
    int rc = check_camera(cam);
    if (rc != OVS_OK) {
        return rc;
    }
    if (n < 0 || (n > 0 && (!dist_keypts || !undist_keypts))) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if ((rc = check_device(device)) != OVS_OK) {
        return rc;
    }
    if (n == 0) {
        return OVS_OK; // undist_keypts.clear()
    }
    ovs::device_guard guard(device);
    thread_ctx* ctx = ctx_for(device);
    if ((rc = ensure_stream(ctx)) != OVS_OK) return rc;
    const size_t nn = n;
    if ((rc = ctx->arena.reserve(2 * al(nn * sizeof(ovs_keypoint)))) != OVS_OK) return rc;
    carver cv;
    cv.base = ctx->arena.as<uint8_t>();
    auto* d_in = cv.take<ovs_keypoint>(nn);
    auto* d_out = cv.take<ovs_keypoint>(nn);
    OVS_HIP(hipMemcpyAsync(d_in, dist_keypts, nn * sizeof(ovs_keypoint), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = ovs_undistort_keypoints_dev(cam, d_in, nullptr, 1, n, d_out, nullptr, device, ctx->stream)) != OVS_OK) return rc;
    OVS_HIP(hipMemcpyAsync(undist_keypts, d_out, nn * sizeof(ovs_keypoint), hipMemcpyDeviceToHost, ctx->stream));
    OVS_HIP(hipStreamSynchronize(ctx->stream));
    return OVS_OK;
}

} // extern "C"
