// ovs_match_common.h -- device helpers (popcount Hamming, wave reductions) and host plumbing
// (per-thread stream + scratch arena) shared by the matcher translation units.
#ifndef OVS_MATCH_COMMON_H
#define OVS_MATCH_COMMON_H

#include "ovs_internal.h"

#include <cstring>
#include <vector>

namespace ovsm {

typedef unsigned int u32;
typedef unsigned long long u64;

constexpr u32 kThrLow = 50;   // HAMMING_DIST_THR_LOW, src/openvslam/match/base.h:13
constexpr u32 kThrHigh = 100; // HAMMING_DIST_THR_HIGH, src/openvslam/match/base.h:14
constexpr u32 kMaxDist = 256; // MAX_HAMMING_DIST, src/openvslam/match/base.h:15
constexpr u64 kNoKey = ~0ull;

// compute_descriptor_distance_32 (src/openvslam/match/base.h:18-39): the SWAR bit count of the
// reference is v_bcnt_u32_b32 here, 8 dwords per descriptor.
__device__ __forceinline__ u32 hamming8(const u32* a, const u32* b) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        d += __popc(a[i] ^ b[i]);
    }
    return d;
}

__device__ __forceinline__ void load_desc(const uint8_t* p, u32* out) {
    const uint4 lo = *reinterpret_cast<const uint4*>(p);
    const uint4 hi = *reinterpret_cast<const uint4*>(p + 16);
    out[0] = lo.x; out[1] = lo.y; out[2] = lo.z; out[3] = lo.w;
    out[4] = hi.x; out[5] = hi.y; out[6] = hi.z; out[7] = hi.w;
}

// descriptors that are only 4-byte aligned (caller buffers)
__device__ __forceinline__ void load_desc4(const uint8_t* p, u32* out) {
    const u32* q = reinterpret_cast<const u32*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        out[i] = q[i];
    }
}

template <int W>
__device__ __forceinline__ u64 group_min_u64(u64 v) {
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) {
        const u32 lo = __shfl_xor((u32)v, d, W);
        const u32 hi = __shfl_xor((u32)(v >> 32), d, W);
        const u64 o = ((u64)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

// minimum over aligned groups of W lanes, in every lane of the group; 8-lane groups stay inside the
// vector ALU (two quad permutes and the half-row mirror as DPP operands of v_min_u32)
template <int W>
__device__ __forceinline__ u32 group_min_u32(u32 v) {
    if (W == 8) {
        v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xf, 0xf, false));  // quad_perm [1,0,3,2]
        v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xf, 0xf, false));  // quad_perm [2,3,0,1]
        v = min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xf, 0xf, false)); // row_half_mirror
        return v;
    }
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) {
        v = min(v, (u32)__shfl_xor(v, d, W));
    }
    return v;
}

template <int W>
__device__ __forceinline__ u32 group_sum_u32(u32 v) {
    if (W == 8) { // the same three DPP operands as group_min_u32
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); // row_half_mirror
        return v;
    }
#pragma unroll
    for (int d = W / 2; d >= 1; d >>= 1) {
        v += __shfl_xor(v, d, W);
    }
    return v;
}

__device__ __forceinline__ u32 wave_sum_u32(u32 v) { return group_sum_u32<64>(v); }

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts (no LDS round trips).
__device__ __forceinline__ u32 wave_incl_scan_u32(u32 x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 into rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 into rows 2, 3
    return (u32)v;
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) { return group_min_u64<64>(v); }

__device__ __forceinline__ int cv_floor_dd(double v) {
    const int i = (int)v;
    return i - (i > v);
}
__device__ __forceinline__ int cv_ceil_dd(double v) {
    const int i = (int)v;
    return i + (i < v);
}
__device__ __forceinline__ int cv_floor_ff(float v) {
    const int i = (int)v;
    return i - (i > v);
}
__device__ __forceinline__ int cv_ceil_ff(float v) {
    const int i = (int)v;
    return i + (i < v);
}

// angle_checker::append_delta_angle, src/openvslam/match/angle_checker.h:72-83 (30 bins)
__device__ __forceinline__ int angle_bin(float delta_angle) {
    if (delta_angle < 0.0f) {
        delta_angle = __fadd_rn(delta_angle, 360.0f);
    }
    if (360.0f <= delta_angle) {
        delta_angle = __fsub_rn(delta_angle, 360.0f);
    }
    return __float2int_rn(__fmul_rn(delta_angle, 1.0f / 30));
}

// ---- host plumbing ----------------------------------------------------------------------------------
// One growable scratch arena per (host thread, device, caller stream) for the device-pointer entry
// points, plus one internal context per (host thread, device) -- its own stream, arena and page-locked
// mirror -- for the entry points that take host buffers.  The matcher entry points keep no other
// state, so they can be called from the tracking, mapping and global optimisation threads
// concurrently, and the device-pointer entry points may be in flight on several caller streams at
// once: work queued on one stream never shares scratch memory with work queued on another.  A caller
// that passes the null stream gets a context of its own as well (the internal context is not the
// null stream's: the internal stream is non-blocking and not ordered against it).
struct thread_ctx {
    int device = -1;
    bool internal = false;        // the context of the host-buffer entry points
    hipStream_t key = nullptr;    // caller stream this context belongs to (internal == false)
    hipStream_t stream = nullptr; // internal stream (internal == true)
    ovs::dev_buffer arena;
    ovs::pinned_buffer mirror; // page-locked image of the arena, see host_stager
};

// The contexts of a host thread; released when the thread ends (its streams are drained first).
struct thread_ctx_list {
    std::vector<thread_ctx*> ctxs; // pointers stay valid when the vector grows
    ~thread_ctx_list() {
        for (thread_ctx* c : ctxs) {
            int prev = -1;
            if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(c->device) == hipSuccess) {
                if (c->stream) {
                    (void)hipStreamSynchronize(c->stream);
                    (void)hipStreamDestroy(c->stream);
                }
                c->arena.release(); // hipFree waits for work that still uses the memory
                c->mirror.release();
                (void)hipSetDevice(prev);
            }
            delete c;
        }
    }
};

inline thread_ctx* ctx_lookup(int device, bool internal, hipStream_t key) {
    static thread_local thread_ctx_list list;
    for (thread_ctx* c : list.ctxs) {
        if (c->device == device && c->internal == internal && c->key == key) {
            return c;
        }
    }
    thread_ctx* c = new thread_ctx;
    c->device = device;
    c->internal = internal;
    c->key = key;
    list.ctxs.push_back(c);
    return c;
}

// context of the host-buffer entry points
inline thread_ctx* ctx_for(int device) { return ctx_lookup(device, true, nullptr); }
// context of work the caller enqueues on `stream` (the null stream included)
inline thread_ctx* ctx_for_stream(int device, hipStream_t stream) { return ctx_lookup(device, false, stream); }

struct carver {
    uint8_t* base = nullptr;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        T* p = reinterpret_cast<T*>(base + off);
        off += (n * sizeof(T) + 255) & ~static_cast<size_t>(255);
        return p;
    }
};

inline size_t al(size_t bytes) { return (bytes + 255) & ~static_cast<size_t>(255); }

inline int check_device(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        return ovs::fail(OVS_ERR_HIP, "no HIP device available (this library has no CPU path)");
    }
    if (device < 0 || device >= ndev) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    }
    return OVS_OK;
}

inline int ensure_stream(thread_ctx* c) {
    if (!c->stream) {
        OVS_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    return OVS_OK;
}

// The host-buffer entry points move a dozen small arrays per call.  Copied one by one from pageable
// caller memory each costs a runtime staging step; instead they are written into a page-locked
// mirror of the arena at the offsets of their device buffers and travel in ONE transfer covering
// the span from the first to the last of them (bytes in between belong to buffers that kernels
// fill later).  Results come back the same way: one transfer of the span of all requested
// outputs, one synchronisation, then plain copies into the caller's arrays.
struct host_stager {
    thread_ctx* ctx = nullptr;
    hipStream_t st = nullptr;
    size_t up_lo = ~static_cast<size_t>(0), up_hi = 0, dn_lo = ~static_cast<size_t>(0), dn_hi = 0;
    struct pending {
        void* dst;
        size_t off, bytes;
    };
    std::vector<pending> gets;

    // after ctx->arena.reserve(...)
    int init(thread_ctx* c, hipStream_t stream) {
        ctx = c;
        st = stream;
        return c->mirror.reserve(c->arena.bytes);
    }
    size_t offset_of(const void* d) const { return static_cast<size_t>(static_cast<const uint8_t*>(d) - ctx->arena.as<uint8_t>()); }
    void touch_up(size_t off, size_t bytes) {
        up_lo = off < up_lo ? off : up_lo;
        up_hi = off + bytes > up_hi ? off + bytes : up_hi;
    }
    void put(void* d_dst, const void* src, size_t bytes) {
        if (bytes) {
            const size_t off = offset_of(d_dst);
            std::memcpy(ctx->mirror.as<uint8_t>() + off, src, bytes);
            touch_up(off, bytes);
        }
    }
    void fill(void* d_dst, int value, size_t bytes) {
        if (bytes) {
            const size_t off = offset_of(d_dst);
            std::memset(ctx->mirror.as<uint8_t>() + off, value, bytes);
            touch_up(off, bytes);
        }
    }
    hipError_t upload() {
        if (up_hi <= up_lo) {
            return hipSuccess;
        }
        return hipMemcpyAsync(ctx->arena.as<uint8_t>() + up_lo, ctx->mirror.as<uint8_t>() + up_lo, up_hi - up_lo, hipMemcpyHostToDevice, st);
    }
    void get(void* host_dst, const void* d_src, size_t bytes) {
        if (bytes) {
            const size_t off = offset_of(d_src);
            gets.push_back(pending{host_dst, off, bytes});
            dn_lo = off < dn_lo ? off : dn_lo;
            dn_hi = off + bytes > dn_hi ? off + bytes : dn_hi;
        }
    }
    // enqueues the transfer, waits for the stream, delivers
    hipError_t download() {
        hipError_t e = hipSuccess;
        if (dn_hi > dn_lo) {
            e = hipMemcpyAsync(ctx->mirror.as<uint8_t>() + dn_lo, ctx->arena.as<uint8_t>() + dn_lo, dn_hi - dn_lo, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) {
            e = hipStreamSynchronize(st);
        }
        if (e == hipSuccess) {
            for (const pending& g : gets) {
                std::memcpy(g.dst, ctx->mirror.as<uint8_t>() + g.off, g.bytes);
            }
        }
        return e;
    }
};

// Image rows between pageable caller memory and a packed device buffer of the arena, through the
// page-locked mirror (one contiguous transfer instead of a strided copy from pageable memory).
inline hipError_t image_up(thread_ctx* ctx, hipStream_t st, void* d_dst, const void* src, size_t src_stride, size_t row_bytes, int rows) {
    uint8_t* pin = ctx->mirror.as<uint8_t>() + (static_cast<uint8_t*>(d_dst) - ctx->arena.as<uint8_t>());
    for (int y = 0; y < rows; ++y) {
        std::memcpy(pin + row_bytes * y, static_cast<const uint8_t*>(src) + src_stride * y, row_bytes);
    }
    return hipMemcpyAsync(d_dst, pin, row_bytes * rows, hipMemcpyHostToDevice, st);
}
// enqueues the transfer and waits for the stream before the rows are delivered
inline hipError_t image_down(thread_ctx* ctx, hipStream_t st, void* dst, size_t dst_stride, const void* d_src, size_t row_bytes, int rows) {
    uint8_t* pin = ctx->mirror.as<uint8_t>() + (static_cast<const uint8_t*>(d_src) - ctx->arena.as<uint8_t>());
    hipError_t e = hipMemcpyAsync(pin, d_src, row_bytes * rows, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) {
        e = hipStreamSynchronize(st);
    }
    if (e == hipSuccess) {
        for (int y = 0; y < rows; ++y) {
            std::memcpy(static_cast<uint8_t*>(dst) + dst_stride * y, pin + row_bytes * y, row_bytes);
        }
    }
    return e;
}

} // namespace ovsm

#endif
