// ovs_common.hip -- error state, device helpers, launch timing, static scale tables.
#include "ovs_internal.h"

#include <cmath>
#include <cstring>
#include <map>

namespace ovs {

static thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char* what, const char* file, int line) {
    g_last_error = std::string("HIP error '") + hipGetErrorString(e) + "' in " + what + " at " + file + ":" + std::to_string(line);
    return OVS_ERR_HIP;
}

int dev_buffer::reserve(size_t n) {
    if (n <= bytes) {
        return OVS_OK;
    }
    if (ptr) {
        OVS_HIP(hipFree(ptr));
        ptr = nullptr;
        bytes = 0;
    }
    OVS_HIP(hipMalloc(&ptr, n));
    bytes = n;
    return OVS_OK;
}

void dev_buffer::release() {
    if (ptr) {
        (void)hipFree(ptr);
    }
    ptr = nullptr;
    bytes = 0;
}

int pinned_buffer::reserve(size_t n) {
    if (n <= bytes) {
        return OVS_OK;
    }
    if (ptr) {
        OVS_HIP(hipHostFree(ptr));
        ptr = nullptr;
        bytes = 0;
    }
    OVS_HIP(hipHostMalloc(&ptr, n, hipHostMallocDefault));
    bytes = n;
    return OVS_OK;
}

void pinned_buffer::release() {
    if (ptr) {
        (void)hipHostFree(ptr);
    }
    ptr = nullptr;
    bytes = 0;
}

device_guard::device_guard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) {
        prev = -1;
    }
    ok = hipSetDevice(device) == hipSuccess;
}

device_guard::~device_guard() {
    if (0 <= prev) {
        (void)hipSetDevice(prev);
    }
}

// ---- launch timing ---------------------------------------------------------------------------
struct timing_pending {
    std::string name;
    hipEvent_t start, stop;
};
struct timing_total {
    double ms = 0;
    int launches = 0;
};
static thread_local bool g_timing = false;
static thread_local std::vector<timing_pending> g_pending;
static thread_local std::map<std::string, timing_total> g_totals;
static thread_local std::vector<std::string> g_order;

bool timing_enabled() { return g_timing; }

// Returns whether an entry was opened: timing_end must only close what its timing_begin opened
// (otherwise the stop event of an earlier launch would be recorded a second time).
bool timing_begin(const char* name, hipStream_t stream) {
    timing_pending p;
    p.name = name;
    if (hipEventCreate(&p.start) != hipSuccess) {
        return false;
    }
    if (hipEventCreate(&p.stop) != hipSuccess) {
        (void)hipEventDestroy(p.start);
        return false;
    }
    (void)hipEventRecord(p.start, stream);
    g_pending.push_back(p);
    return true;
}

void timing_end(hipStream_t stream) { (void)hipEventRecord(g_pending.back().stop, stream); }

static void timing_collect() {
    for (auto& p : g_pending) {
        float ms = 0;
        if (hipEventSynchronize(p.stop) == hipSuccess && hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            if (!g_totals.count(p.name)) {
                g_order.push_back(p.name);
            }
            auto& t = g_totals[p.name];
            t.ms += ms;
            t.launches += 1;
        }
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    g_pending.clear();
}

} // namespace ovs

extern "C" {

const char* ovs_last_error(void) { return ovs::g_last_error.c_str(); }

int ovs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

void ovs_timing_enable(int on) {
    if (!on) {
        ovs::timing_collect();
    }
    ovs::g_timing = on != 0;
}

void ovs_timing_reset(void) {
    ovs::timing_collect();
    ovs::g_totals.clear();
    ovs::g_order.clear();
}

int ovs_timing_query(const char* kernel_name, double* total_ms, int* launches) {
    ovs::timing_collect();
    auto it = ovs::g_totals.find(kernel_name);
    if (it == ovs::g_totals.end()) {
        return ovs::fail(OVS_ERR_INVALID_ARGUMENT, std::string("no timing recorded for kernel ") + kernel_name);
    }
    *total_ms = it->second.ms;
    *launches = it->second.launches;
    return OVS_OK;
}

int ovs_timing_names(char* names, int cap) {
    ovs::timing_collect();
    int n = 0;
    for (const auto& name : ovs::g_order) {
        if (n < cap) {
            std::strncpy(names + 64 * n, name.c_str(), 63);
            names[64 * n + 63] = 0;
        }
        ++n;
    }
    return n;
}

void ovs_orb_params_default(ovs_orb_params* p) {
    // orb_params(const YAML::Node&) defaults, src/openvslam/feature/orb_params.cc:27-33
    p->max_num_keypts = 2000;
    p->scale_factor = 1.2f;
    p->num_levels = 8;
    p->ini_fast_thr = 20;
    p->min_fast_thr = 7;
    p->num_mask_rects = 0;
    p->mask_rects = nullptr;
}

// orb_params::calc_* (src/openvslam/feature/orb_params.cc:53-87): running float products.
void ovs_calc_scale_factors(uint32_t num_levels, float scale_factor, float* out) {
    for (uint32_t level = 0; level < num_levels; ++level) {
        out[level] = level == 0 ? 1.0f : scale_factor * out[level - 1];
    }
}

void ovs_calc_inv_scale_factors(uint32_t num_levels, float scale_factor, float* out) {
    for (uint32_t level = 0; level < num_levels; ++level) {
        out[level] = level == 0 ? 1.0f : (1.0f / scale_factor) * out[level - 1];
    }
}

void ovs_calc_level_sigma_sq(uint32_t num_levels, float scale_factor, float* out) {
    float scale_factor_at_level = 1.0f;
    for (uint32_t level = 0; level < num_levels; ++level) {
        if (level == 0) {
            out[level] = 1.0f;
            continue;
        }
        scale_factor_at_level = scale_factor * scale_factor_at_level;
        out[level] = scale_factor_at_level * scale_factor_at_level;
    }
}

void ovs_calc_inv_level_sigma_sq(uint32_t num_levels, float scale_factor, float* out) {
    float scale_factor_at_level = 1.0f;
    for (uint32_t level = 0; level < num_levels; ++level) {
        if (level == 0) {
            out[level] = 1.0f;
            continue;
        }
        scale_factor_at_level = scale_factor * scale_factor_at_level;
        out[level] = 1.0f / (scale_factor_at_level * scale_factor_at_level);
    }
}

} // extern "C"
