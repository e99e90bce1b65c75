// What a streaming kernel's access pattern costs on this part: the same number of bytes read (and
// optionally written) by wavefronts that each walk (a) one contiguous run, (b) a 256-byte wide column
// of 64 image rows (the tile of the blur kernel), (c) a column of 16 rows that is 1024 bytes wide.
// Build: hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kW = 1280, kH = 384;            // one plane (bytes per row, rows)
constexpr int kPlane = kW * kH;
constexpr int kFrames = 768;                   // 377 MB per buffer

// MODE 0: wave w of the launch reads rows of 256 bytes that follow each other in memory (64 rows = 16 KB)
// MODE 1: wave = tile of 256 columns x 64 rows of a plane (row pitch kW); tiles column-fastest
// MODE 2: wave = tile of 1024 columns x 16 rows, 16 bytes per lane; 1280-byte rows: one tile of 1024 and one of 256 columns
// DEPTH rows are in flight per wave.  WRITE: every row is also stored to the second buffer.
template <int MODE, int DEPTH, bool WRITE, int EXTRA = 0, int MASKED = 0>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned acc = 0;
    if (MODE == 0 || MODE == 1) {
        long long base;
        int pitch;
        if (MODE == 0) {
            base = wave * (64LL * 256);
            pitch = 256;
        }
        else {
            const int tiles_per_plane = (kW / 256) * (kH / 64);
            const long long plane = wave / tiles_per_plane;
            const int t = (int)(wave - plane * tiles_per_plane);
            const int ty = t / (kW / 256), tx = t - ty * (kW / 256);
            base = plane * kPlane + (long long)ty * 64 * kW + tx * 256;
            pitch = kW;
        }
        const unsigned char* p = src + base + 4 * lane;
        unsigned char* q = dst + base + 4 * lane;
        unsigned v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            v[d] = *reinterpret_cast<const unsigned*>(p + (long long)d * pitch);
        }
#pragma unroll 4
        for (int r = 0; r < 64; ++r) {
            const int nr = r + DEPTH < 64 ? r + DEPTH : 63;
            unsigned n = *reinterpret_cast<const unsigned*>(p + (long long)nr * pitch);
            if (EXTRA >= 1) {
                n ^= *reinterpret_cast<const unsigned*>(p + (long long)nr * pitch + 4);
            }
            if (EXTRA >= 2) {
                n ^= *reinterpret_cast<const unsigned*>(p + (long long)nr * pitch + 8);
            }
            if (MASKED > 0 && (lane == 0 || lane == 63)) {
#pragma unroll
                for (int m = 0; m < MASKED; ++m) {
                    n ^= p[(long long)nr * pitch + 12 + m];
                }
            }
            const unsigned cur = v[0];
#pragma unroll
            for (int d = 0; d + 1 < DEPTH; ++d) {
                v[d] = v[d + 1];
            }
            v[DEPTH - 1] = n;
            acc ^= cur;
            if (WRITE) {
                *reinterpret_cast<unsigned*>(q + (long long)r * pitch) = cur + 1u;
            }
        }
    }
    else {
        // 16 rows, 16 bytes per lane; a plane row is one 1024-byte tile and one 256-byte remainder that
        // lanes 0..15 of the next wave-tile handle: here simply tiles of 1024 columns over a 1280 pitch,
        // the remainder columns are read by a second pass of the same wave (lanes 0..15)
        const int tiles_per_plane = kH / 16;
        const long long plane = wave / tiles_per_plane;
        const int ty = (int)(wave - plane * tiles_per_plane);
        const long long base = plane * kPlane + (long long)ty * 16 * kW;
        const unsigned char* p = src + base + 16 * lane;
        unsigned char* q = dst + base + 16 * lane;
        uint4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            v[d] = *reinterpret_cast<const uint4*>(p + (long long)d * kW);
        }
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int nr = r + DEPTH < 16 ? r + DEPTH : 15;
            const uint4 n = *reinterpret_cast<const uint4*>(p + (long long)nr * kW);
            const uint4 cur = v[0];
#pragma unroll
            for (int d = 0; d + 1 < DEPTH; ++d) {
                v[d] = v[d + 1];
            }
            v[DEPTH - 1] = n;
            acc ^= cur.x ^ cur.y ^ cur.z ^ cur.w;
            if (WRITE) {
                *reinterpret_cast<uint4*>(q + (long long)r * kW) = make_uint4(cur.x + 1u, cur.y, cur.z, cur.w);
            }
        }
        if (lane < 16) {
            for (int r = 0; r < 16; ++r) {
                const uint4 cur = *reinterpret_cast<const uint4*>(p + 1024 + (long long)r * kW);
                acc ^= cur.x ^ cur.y ^ cur.z ^ cur.w;
                if (WRITE) {
                    *reinterpret_cast<uint4*>(q + 1024 + (long long)r * kW) = cur;
                }
            }
        }
    }
    if (acc == 0x12345678u) {
        sink[0] = acc;
    }
}

template <int MODE, int DEPTH, bool WRITE, int EXTRA = 0, int MASKED = 0>
void run(const char* name, const unsigned char* src, unsigned char* dst, unsigned* sink) {
    const long long bytes = (long long)kFrames * kPlane;
    long long waves;
    if (MODE == 0) {
        waves = bytes / (64 * 256);
    }
    else if (MODE == 1) {
        waves = (long long)kFrames * (kW / 256) * (kH / 64);
    }
    else {
        waves = (long long)kFrames * (kH / 16);
    }
    const int blocks = (int)(waves / 4);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, DEPTH, WRITE, EXTRA, MASKED>), dim3(blocks), dim3(256), 0, 0, src, dst, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) {
            best = ms;
        }
    }
    const double moved = (double)bytes * (WRITE ? 2.0 : 1.0);
    printf("%-44s depth %d %s: %.3f ms, %.0f GB/s\n", name, DEPTH, WRITE ? "read+write" : "read only ", best, moved / best / 1e6);
}

int main() {
    const long long bytes = (long long)kFrames * kPlane;
    unsigned char *src, *dst;
    unsigned* sink;
    CK(hipMalloc(&src, bytes + 4096));
    CK(hipMalloc(&dst, bytes + 4096));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, bytes + 4096));
    CK(hipMemset(dst, 0, bytes + 4096));
    run<0, 2, false>("contiguous 16 KB per wave, 4 B per lane", src, dst, sink);
    run<0, 8, false>("contiguous 16 KB per wave, 4 B per lane", src, dst, sink);
    run<1, 2, false>("256 x 64 tile, 4 B per lane", src, dst, sink);
    run<1, 4, false>("256 x 64 tile, 4 B per lane", src, dst, sink);
    run<1, 8, false>("256 x 64 tile, 4 B per lane", src, dst, sink);
    run<1, 16, false>("256 x 64 tile, 4 B per lane", src, dst, sink);
    run<2, 2, false>("1024 (+256) x 16 tile, 16 B per lane", src, dst, sink);
    run<2, 4, false>("1024 (+256) x 16 tile, 16 B per lane", src, dst, sink);
    run<0, 2, true>("contiguous 16 KB per wave, 4 B per lane", src, dst, sink);
    run<0, 8, true>("contiguous 16 KB per wave, 4 B per lane", src, dst, sink);
    run<1, 2, true>("256 x 64 tile, 4 B per lane", src, dst, sink);
    run<1, 8, true>("256 x 64 tile, 4 B per lane", src, dst, sink);
    run<2, 2, true>("1024 (+256) x 16 tile, 16 B per lane", src, dst, sink);
    run<2, 4, true>("1024 (+256) x 16 tile, 16 B per lane", src, dst, sink);
    run<1, 2, true, 2, 0>("256 x 64 tile, 3 dword loads per row", src, dst, sink);
    run<1, 2, false, 2, 0>("256 x 64 tile, 3 dword loads per row", src, dst, sink);
    run<1, 2, true, 0, 4>("256 x 64 tile, 1 dword + 4 masked byte loads", src, dst, sink);
    run<1, 2, false, 0, 4>("256 x 64 tile, 1 dword + 4 masked byte loads", src, dst, sink);
    run<1, 2, true, 0, 1>("256 x 64 tile, 1 dword + 1 masked byte load", src, dst, sink);
    return 0;
}
