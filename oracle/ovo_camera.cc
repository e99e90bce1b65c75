// CPU ORACLE (test infrastructure, never part of the product path).
// Keypoints to bearing vectors, restated from the reference:
//   camera::perspective::convert_keypoints_to_bearings      src/openvslam/camera/perspective.cc:130-138
//   camera::fisheye::convert_keypoints_to_bearings          src/openvslam/camera/fisheye.cc:169-177
//   camera::equirectangular::convert_keypoints_to_bearings  src/openvslam/camera/equirectangular.cc:45-56
// No third-party arithmetic is involved (plain double expressions and libm's sin / cos).
#include "ovo.h"

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
