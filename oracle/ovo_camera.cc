// CPU ORACLE (test infrastructure, never part of the product path).
// Keypoints to bearing vectors, restated from the reference:
//   camera::perspective::convert_keypoints_to_bearings      src/openvslam/camera/perspective.cc:130-138
//   camera::fisheye::convert_keypoints_to_bearings          src/openvslam/camera/fisheye.cc:169-177
//   camera::equirectangular::convert_keypoints_to_bearings  src/openvslam/camera/equirectangular.cc:45-56
// No third-party arithmetic is involved (plain double expressions and libm's sin / cos).
#include "ovo.h"

#include <algorithm>
#include <cmath>

extern "C" int ovo_convert_keypoints_to_bearings(const ovs_camera* cam, const ovs_keypoint* undist_keypts, int n, double* bearings) {
    if (!cam || n < 0 || (n > 0 && (!undist_keypts || !bearings))) {
        return OVS_ERR_INVALID_ARGUMENT;
    }
    for (int idx = 0; idx < n; ++idx) {
        const float ptx = undist_keypts[idx].x, pty = undist_keypts[idx].y;
        double* b = bearings + 3 * idx;
        if (cam->model == OVS_CAMERA_EQUIRECTANGULAR) {
            // cols_ / rows_ are unsigned int: pt.x / cols_ is evaluated in float (equirectangular.cc:49-50)
            const unsigned int cols = cam->cols, rows = cam->rows;
            const double lon = (ptx / cols - 0.5) * (2 * M_PI);
            const double lat = -(pty / rows - 0.5) * M_PI;
            b[0] = std::cos(lat) * std::sin(lon);
            b[1] = -std::sin(lat);
            b[2] = std::cos(lat) * std::cos(lon);
        }
        else if (cam->model == OVS_CAMERA_PERSPECTIVE || cam->model == OVS_CAMERA_FISHEYE) {
            const double x_normalized = (ptx - cam->cx) / cam->fx;
            const double y_normalized = (pty - cam->cy) / cam->fy;
            const double l2_norm = std::sqrt(x_normalized * x_normalized + y_normalized * y_normalized + 1.0);
            b[0] = x_normalized / l2_norm;
            b[1] = y_normalized / l2_norm;
            b[2] = 1.0 / l2_norm;
        }
        else {
            return OVS_ERR_INVALID_ARGUMENT;
        }
    }
    return OVS_OK;
}

// ---------------------------------------------------------------------------------------------
// Keypoint undistortion.  The arithmetic lives in OpenCV (absent from the reference tree):
// cv::undistortPoints with a termination criterion and cv::fisheye::undistortPoints, restated from
// the published source of OpenCV 4.1.0 (the version the reference's Dockerfile builds;
// calib3d/src/undistort.cpp cvUndistortPointsInternal, calib3d/src/fisheye.cpp).  No reference test
// holds a value of either: parity unpinned.
//   camera::perspective::undistort_keypoints      src/openvslam/camera/perspective.cc:99-128
//   camera::fisheye::undistort_keypoints          src/openvslam/camera/fisheye.cc:139-167
//   camera::equirectangular::undistort_keypoints  src/openvslam/camera/equirectangular.cc:41-43
// ---------------------------------------------------------------------------------------------
#include <cfloat>

namespace {

// s = 0; s += a(i,k) * b(k) for k = 0, 1, 2: the accumulation order of cv::Matx33d * cv::Vec3d
inline double matx_row(double a0, double a1, double a2, double b0, double b1, double b2) {
    double s = 0.0;
    s += a0 * b0;
    s += a1 * b1;
    s += a2 * b2;
    return s;
}

// cvUndistortPointsInternal: _distCoeffs given (5 x 1), matR empty, matP = the camera matrix,
// criteria = COUNT | EPS, 20, 1e-6
void undistort_perspective(const ovs_camera* cam, float px, float py, float* ox, float* oy) {
    // cv_cam_matrix_ and cv_dist_params_ are cv::Mat_<float> (perspective.cc:21-22); cvConvert widens them
    const double fx = (float)cam->fx, fy = (float)cam->fy, cx = (float)cam->cx, cy = (float)cam->cy;
    double k[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) {
        k[i] = (float)cam->dist[i];
    }
    const double ifx = 1. / fx, ify = 1. / fy;
    const int max_count = 20;
    const double epsilon = 1e-6;
    // RR = P * identity: every product is with 0 or 1, the result is P itself
    const double RR[3][3] = {{fx, 0, cx}, {0, fy, cy}, {0, 0, 1}};

    double x = px, y = py, x0, y0;
    const double u = x, v = y;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    // tilt compensation with k[12] = k[13] = 0: both tilt matrices are the identity
    double untilt2 = matx_row(0, 0, 1, x, y, 1);
    double inv_proj = untilt2 ? 1. / untilt2 : 1;
    const double ux = matx_row(1, 0, 0, x, y, 1), uy = matx_row(0, 1, 0, x, y, 1);
    x0 = x = inv_proj * ux;
    y0 = y = inv_proj * uy;

    double error = DBL_MAX;
    for (int j = 0;; j++) {
        if (j >= max_count) break;
        if (error < epsilon) break;
        double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {
            x = (u - cx) * ifx;
            y = (v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;

        // reprojection error of the estimate (criteria.type & EPS)
        r2 = x * x + y * y;
        const double r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6;
        const double icdist2 = 1. / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6);
        const double xd0 = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4;
        const double yd0 = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4;
        const double t2 = matx_row(0, 0, 1, xd0, yd0, 1);
        inv_proj = t2 ? 1. / t2 : 1;
        const double xd = inv_proj * matx_row(1, 0, 0, xd0, yd0, 1);
        const double yd = inv_proj * matx_row(0, 1, 0, xd0, yd0, 1);
        const double x_proj = xd * fx + cx, y_proj = yd * fy + cy;
        const double ex = x_proj - u, ey = y_proj - v;
        error = std::sqrt(ex * ex + ey * ey); // pow(e, 2) is e * e
    }
    const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
    const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
    const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
    *ox = (float)(xx * ww);
    *oy = (float)(yy * ww);
}

// cv::fisheye::undistortPoints: K and D as float matrices (fisheye.cc:21-22), R empty, P = K
void undistort_fisheye(const ovs_camera* cam, float px, float py, float* ox, float* oy) {
    const double f0 = (float)cam->fx, f1 = (float)cam->fy, c0 = (float)cam->cx, c1 = (float)cam->cy;
    const double k[4] = {(double)(float)cam->dist[0], (double)(float)cam->dist[1], (double)(float)cam->dist[2], (double)(float)cam->dist[3]};
    const double kPi = 3.1415926535897932384626433832795; // CV_PI
    const double pw0 = ((double)px - c0) / f0, pw1 = ((double)py - c1) / f1;
    double scale = 1.0;
    double theta_d = std::sqrt(pw0 * pw0 + pw1 * pw1);
    // the model is valid up to a field of view of 180 degrees; values are clipped
    theta_d = std::min(std::max(-kPi / 2., theta_d), kPi / 2.);
    if (theta_d > 1e-8) {
        double theta = theta_d;
        const double EPS = 1e-8;
        for (int j = 0; j < 10; j++) {
            const double theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta6 * theta2;
            const double k0_theta2 = k[0] * theta2, k1_theta4 = k[1] * theta4, k2_theta6 = k[2] * theta6, k3_theta8 = k[3] * theta8;
            // one Newton step on theta * (1 + k0 theta^2 + ...) = theta_d
            const double theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d)
                                     / (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (std::fabs(theta_fix) < EPS) break;
        }
        scale = std::tan(theta) / theta_d;
    }
    const double pu0 = pw0 * scale, pu1 = pw1 * scale;
    // RR = P * identity; pr = RR * (pu, 1)
    const double pr0 = matx_row(f0, 0, c0, pu0, pu1, 1.0), pr1 = matx_row(0, f1, c1, pu0, pu1, 1.0), pr2 = matx_row(0, 0, 1, pu0, pu1, 1.0);
    *ox = (float)(pr0 / pr2);
    *oy = (float)(pr1 / pr2);
}

} // namespace

extern "C" int ovo_undistort_keypoints(const ovs_camera* cam, const ovs_keypoint* dist_keypts, int n, ovs_keypoint* undist_keypts) {
    if (!cam || n < 0 || (n > 0 && (!dist_keypts || !undist_keypts))) {
        return OVS_ERR_INVALID_ARGUMENT;
    }
    if (cam->model != OVS_CAMERA_PERSPECTIVE && cam->model != OVS_CAMERA_FISHEYE && cam->model != OVS_CAMERA_EQUIRECTANGULAR) {
        return OVS_ERR_INVALID_ARGUMENT;
    }
    for (int idx = 0; idx < n; ++idx) {
        if (cam->model == OVS_CAMERA_EQUIRECTANGULAR) {
            undist_keypts[idx] = dist_keypts[idx]; // undist_keypts = dist_keypts
            continue;
        }
        // undist_keypts.resize(n): default cv::KeyPoint (response 0, class_id -1), then pt, angle, size, octave
        ovs_keypoint out;
        out.response = 0.0f;
        out.class_id = -1;
        if (cam->model == OVS_CAMERA_PERSPECTIVE) {
            undistort_perspective(cam, dist_keypts[idx].x, dist_keypts[idx].y, &out.x, &out.y);
        }
        else {
            undistort_fisheye(cam, dist_keypts[idx].x, dist_keypts[idx].y, &out.x, &out.y);
        }
        out.angle = dist_keypts[idx].angle;
        out.size = dist_keypts[idx].size;
        out.octave = dist_keypts[idx].octave;
        undist_keypts[idx] = out;
    }
    return OVS_OK;
}
