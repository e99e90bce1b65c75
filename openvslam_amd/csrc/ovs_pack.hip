// ovs_pack.hip -- dense per-frame results for the trip to the host.
//
// The reference returns exactly sized containers: std::vector<cv::KeyPoint> and a cv::Mat with one row per
// keypoint (src/openvslam/feature/orb_extractor.cc:116-120), landmark vectors of num_keypts_ entries
// (data/frame.cc:30-40).  The batched device entry points write frame f at slot f of capacity-strided arrays
// (the keypoint count is only known on the device), so a download of those arrays moves the capacity, about
// 1.8 times the payload at 2000 keypoints.  ovs_pack_batch_results_dev moves every frame's count[f] records
// to offsets[f] of dense arrays: a prefix sum over the counts and three streaming copies, all row
// contiguous (28-byte keypoint records and 32-byte descriptors as dwords, one matched index per keypoint).
#include <algorithm>

#include "ovs_match_common.h"

using namespace ovsm;

namespace {

constexpr int kPackScanThreads = 1024;

__global__ __launch_bounds__(kPackScanThreads) void k_pack_offsets(const int* __restrict__ counts, int n_frames, int cap, int* __restrict__ offsets) {
    __shared__ u32 red[kPackScanThreads / 64];
    __shared__ u32 carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        carry = 0;
        offsets[0] = 0;
    }
    __syncthreads();
    for (int f0 = 0; f0 < n_frames; f0 += kPackScanThreads) {
        const int f = f0 + tid;
        const u32 c = f < n_frames ? (u32)max(0, min(counts[f], cap)) : 0u;
        const u32 inc = wave_incl_scan_u32(c);
        if (lane == 63) {
            red[wave] = inc;
        }
        __syncthreads();
        u32 before = carry, total = 0;
        for (int w = 0; w < kPackScanThreads / 64; ++w) {
            const u32 v = red[w];
            before += w < wave ? v : 0u;
            total += v;
        }
        if (f < n_frames) {
            offsets[f + 1] = (int)(before + inc);
        }
        __syncthreads();
        if (tid == 0) {
            carry += total;
        }
        __syncthreads();
    }
}

// grid (chunks, n_frames): a frame's records as runs of dwords
__global__ __launch_bounds__(256) void k_pack_rows(const int* __restrict__ counts, int cap, const int* __restrict__ offsets, const u32* __restrict__ kps,
                                                   const u32* __restrict__ desc, const int* __restrict__ matched, u32* __restrict__ out_kps,
                                                   u32* __restrict__ out_desc, int* __restrict__ out_matched) {
    const int f = blockIdx.y;
    const size_t n = (size_t)max(0, min(counts[f], cap));
    const size_t src = (size_t)f * cap, dst = (size_t)offsets[f];
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (kps) {
        for (size_t i = t0; i < 7 * n; i += stride) {
            out_kps[7 * dst + i] = kps[7 * src + i];
        }
    }
    if (desc) {
        for (size_t i = t0; i < 8 * n; i += stride) {
            out_desc[8 * dst + i] = desc[8 * src + i];
        }
    }
    if (matched) {
        for (size_t i = t0; i < n; i += stride) {
            out_matched[dst + i] = matched[src + i];
        }
    }
}

} // namespace

extern "C" int ovs_pack_batch_results_dev(const int32_t* d_counts, int n_frames, int cap, const ovs_keypoint* d_kps, const uint8_t* d_desc,
                                          const int32_t* d_matched, int32_t* d_offsets, ovs_keypoint* d_packed_kps, uint8_t* d_packed_desc,
                                          int32_t* d_packed_matched, int device, void* stream_) {
    if (!d_counts || !d_offsets || n_frames <= 0 || n_frames > 65535 || cap <= 0 || (d_kps && !d_packed_kps) || (d_desc && !d_packed_desc)
        || (d_matched && !d_packed_matched)) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "invalid argument");
    }
    if ((reinterpret_cast<uintptr_t>(d_desc) | reinterpret_cast<uintptr_t>(d_packed_desc) | reinterpret_cast<uintptr_t>(d_kps)
         | reinterpret_cast<uintptr_t>(d_packed_kps))
        & 3) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "arrays must be 4-byte aligned");
    }
    int rc = check_device(device);
    if (rc != OVS_OK) {
        return rc;
    }
    ovs::device_guard guard(device);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    OVS_LAUNCH("k_pack_offsets", k_pack_offsets, dim3(1), dim3(kPackScanThreads), 0, st, d_counts, n_frames, cap, d_offsets);
    const int chunks = std::max(1, std::min(16, (cap * 8 + 256 * 4 - 1) / (256 * 4)));
    OVS_LAUNCH("k_pack_rows", k_pack_rows, dim3(chunks, n_frames), dim3(256), 0, st, d_counts, cap, d_offsets, reinterpret_cast<const u32*>(d_kps),
               reinterpret_cast<const u32*>(d_desc), d_matched, reinterpret_cast<u32*>(d_packed_kps), reinterpret_cast<u32*>(d_packed_desc),
               d_packed_matched);
    OVS_HIP(hipGetLastError());
    return OVS_OK;
}
