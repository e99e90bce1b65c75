// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access widths the
// kernels of this library use: every kernel below reads one 512 MiB buffer exactly once (cold,
// larger than the 256 MiB Infinity Cache) and writes 4 bytes per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o tools/calib/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o calib -- tools/calib/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr size_t kBytes = 512ull << 20;

__global__ void read_b8(const uint8_t* p, unsigned* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void read_b32(const uint32_t* p, unsigned* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void read_b32_unaligned(const uint8_t* p, unsigned* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned v;
        __builtin_memcpy(&v, p + 1 + 4 * i, 4);
        acc += v;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void read_b96(const uint32_t* p, unsigned* out, size_t n) { // 12 bytes per lane, stride 12
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        acc += p[3 * i] + p[3 * i + 1] + p[3 * i + 2];
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void read_b128(const uint4* p, unsigned* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 0x12345678u) out[blockIdx.x] = acc;
}
__global__ void write_b32(uint32_t* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
__global__ void write_b128(uint4* p, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(i, i, i, i);
}

int main() {
    uint8_t *a, *flush;
    unsigned* out;
    if (hipMalloc(&a, kBytes + 64) != hipSuccess || hipMalloc(&flush, kBytes) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) return 1;
    hipMemset(a, 1, kBytes + 64);
    const int grid = 256 * 16, block = 256;
#define FLUSH() hipMemset(flush, 0, kBytes); hipDeviceSynchronize();
    FLUSH(); read_b8<<<grid, block>>>(a, out, kBytes); hipDeviceSynchronize();
    FLUSH(); read_b32<<<grid, block>>>((const uint32_t*)a, out, kBytes / 4); hipDeviceSynchronize();
    FLUSH(); read_b32_unaligned<<<grid, block>>>(a, out, kBytes / 4); hipDeviceSynchronize();
    FLUSH(); read_b96<<<grid, block>>>((const uint32_t*)a, out, kBytes / 12); hipDeviceSynchronize();
    FLUSH(); read_b128<<<grid, block>>>((const uint4*)a, out, kBytes / 16); hipDeviceSynchronize();
    FLUSH(); write_b32<<<grid, block>>>((uint32_t*)a, kBytes / 4); hipDeviceSynchronize();
    FLUSH(); write_b128<<<grid, block>>>((uint4*)a, kBytes / 16); hipDeviceSynchronize();
    std::printf("done %zu bytes per kernel\n", kBytes);
    return 0;
}
