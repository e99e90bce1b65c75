// Cost of LDS read instructions on one compute unit: cycles per wave-instruction with 8 waves per SIMD
// resident, for byte / dword / dword-pair / qword reads at scattered (aligned) addresses and at
// conflict-free addresses.  Build: hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// KIND 0: ds_read_u8, 1: ds_read_b32, 2: ds_read2_b32 (two neighbouring dwords), 3: ds_read_b64 (8-byte aligned)
template <int KIND, bool SCATTER>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[70 * 72 + 64];
    for (int i = threadIdx.x; i < 70 * 72 + 64; i += 256) {
        tile[i] = (unsigned char)(i * 7 + 1);
    }
    __syncthreads();
    // 16 addresses per lane, as the exact-score stage has: a pseudo-random tile position (SCATTER) or one
    // dword per lane (conflict free)
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        unsigned base = SCATTER ? (((h >> 8) % (64 * 18)) * 4 + 3 * 72) : ((threadIdx.x & 63) * 4 + 3 * 72);
        base &= ~7u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned a = base + ((j & 3) - 1) * 72 + (j >> 2) * 8;
            if (KIND == 0) {
                acc += tile[a + (h & 3)];
            }
            else if (KIND == 1) {
                acc += *reinterpret_cast<const unsigned*>(&tile[a]);
            }
            else if (KIND == 2) {
                const uint2 v = *reinterpret_cast<const uint2*>(&tile[a]); // 8-byte aligned: the compiler may use b64
                acc += v.x ^ v.y;
            }
            else {
                const unsigned long long v = *reinterpret_cast<const unsigned long long*>(&tile[a]);
                acc += (unsigned)v ^ (unsigned)(v >> 32);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND, bool SCATTER>
void run(const char* name, unsigned* out) {
    const int iters = 2000, blocks = 256 * 8; // 8 workgroups of 4 waves per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<KIND, SCATTER>), dim3(blocks), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) {
            best = ms;
        }
    }
    // wave-instructions per CU: 32 waves x iters x 16
    const double per = best * 1e-3 * 2.4e9 / (32.0 * iters * 16.0);
    printf("%-18s %-14s %.3f ms: %.2f cycles (2.4 GHz) of CU time per wave-instruction\n", name, SCATTER ? "scattered" : "conflict free", best, per);
}

int main() {
    unsigned* out;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    run<0, true>("ds_read_u8", out);
    run<0, false>("ds_read_u8", out);
    run<1, true>("ds_read_b32", out);
    run<1, false>("ds_read_b32", out);
    run<2, true>("ds_read_b64 (uint2)", out);
    run<2, false>("ds_read_b64 (uint2)", out);
    run<3, true>("ds_read_b64", out);
    run<3, false>("ds_read_b64", out);
    return 0;
}
