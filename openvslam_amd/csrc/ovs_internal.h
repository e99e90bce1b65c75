// ovs_internal.h -- shared host-side helpers of the HIP library (error state, launch timing).
#ifndef OVS_INTERNAL_H
#define OVS_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ovs.h"

namespace ovs {

int fail(int code, const std::string& msg);
int fail_hip(hipError_t e, const char* what, const char* file, int line);

#define OVS_HIP(expr)                                                   \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if (_e != hipSuccess) {                                         \
            return ovs::fail_hip(_e, #expr, __FILE__, __LINE__);        \
        }                                                               \
    } while (0)

// Optional per-kernel timing: every launch made through OVS_LAUNCH is bracketed by two events on
// its stream while timing is enabled on the calling thread.
bool timing_enabled();
bool timing_begin(const char* name, hipStream_t stream);
void timing_end(hipStream_t stream);

#define OVS_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                   \
    do {                                                                            \
        const bool ovs_timed_ = ovs::timing_enabled() && ovs::timing_begin(name, stream); \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);        \
        if (ovs_timed_) ovs::timing_end(stream);                                    \
    } while (0)

// A growable device buffer owned by a handle.
struct dev_buffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t n);
    void release();
    template <typename T>
    T* as() const { return static_cast<T*>(ptr); }
};

// A growable page-locked host buffer: staging for the host-buffer entry points (copies between
// pageable caller memory and the device go through the runtime's own small staging chunks, which
// costs milliseconds for an image; one memcpy into pinned memory and one DMA does not).
struct pinned_buffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    int reserve(size_t n);
    void release();
    template <typename T>
    T* as() const { return static_cast<T*>(ptr); }
};

// RAII device selection for entry points that take a device ordinal.
struct device_guard {
    int prev = -1;
    bool ok = false;
    explicit device_guard(int device);
    ~device_guard();
};

// cvRound / cvFloor / cvCeil semantics on the host side of the library.
inline int cv_floor_d(double v) {
    int i = static_cast<int>(v);
    return i - (i > v);
}
inline int cv_ceil_d(double v) {
    int i = static_cast<int>(v);
    return i + (i < v);
}

} // namespace ovs

#endif
