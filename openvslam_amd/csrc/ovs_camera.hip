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
#include "ovs_match_common.h"

using namespace ovsm;

namespace {

__global__ __launch_bounds__(256) void k_bearings(ovs_camera cam, const ovs_keypoint* __restrict__ kps, const int32_t* __restrict__ counts,
                                                  int cap, double* __restrict__ out) {
    const int frame = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = counts ? min(max(counts[frame], 0), cap) : cap;
    if (i >= n) {
        return;
    }
    const size_t at = (size_t)frame * cap + i;
    const float px = kps[at].x, py = kps[at].y;
    double b0, b1, b2;
    if (cam.model == OVS_CAMERA_EQUIRECTANGULAR) {
        // pt.x / cols_ is a float division (cols_ is unsigned int), the rest is double
        const double kPi = 3.14159265358979323846;
        const float fx = __fdiv_rn(px, (float)cam.cols), fy = __fdiv_rn(py, (float)cam.rows);
        const double lon = __dmul_rn(__dsub_rn((double)fx, 0.5), __dmul_rn(2.0, kPi));
        const double lat = __dmul_rn(-__dsub_rn((double)fy, 0.5), kPi);
        const double cl = cos(lat), sl = sin(lat);
        b0 = __dmul_rn(cl, sin(lon));
        b1 = -sl;
        b2 = __dmul_rn(cl, cos(lon));
    }
    else {
        const double xn = __ddiv_rn(__dsub_rn((double)px, cam.cx), cam.fx);
        const double yn = __ddiv_rn(__dsub_rn((double)py, cam.cy), cam.fy);
        const double l2 = __dsqrt_rn(__dadd_rn(__dadd_rn(__dmul_rn(xn, xn), __dmul_rn(yn, yn)), 1.0));
        b0 = __ddiv_rn(xn, l2);
        b1 = __ddiv_rn(yn, l2);
        b2 = __ddiv_rn(1.0, l2);
    }
    out[3 * at] = b0;
    out[3 * at + 1] = b1;
    out[3 * at + 2] = b2;
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

} // extern "C"
