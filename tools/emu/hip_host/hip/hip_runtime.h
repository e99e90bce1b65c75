// Host-only stand-in for <hip/hip_runtime.h>: lets a translation unit of the HIP library (host code and
// kernels, unchanged) be compiled with g++ and run without a GPU, for tests of index arithmetic, data flow
// and memory safety.  Not a performance model and not a memory-model checker.
//
//   * Device memory (hipMalloc) is host memory surrounded by inaccessible address space: a wild index ends
//     in SIGSEGV with a report of the buffer and the offset.  Freed blocks stay inaccessible.
//   * A kernel launch runs synchronously, one workgroup after the other; every thread of a workgroup is a
//     fiber.  A fiber runs until it reaches a workgroup barrier, a wave operation or its end.  A barrier
//     completes when every unfinished thread of the workgroup waits at it.  A wave operation is an exchange:
//     every lane publishes a value, and when all lanes of the wave wait (at an exchange, at a barrier, or are
//     finished) the lanes that wait at the source line that comes first in the file exchange their values
//     and go on -- loops finish before the code after them, as on the hardware, and lanes that took another
//     branch are simply not part of the exchange (the hardware's execution mask).  ballot, shuffles,
//     readlane, DPP moves and the wave barrier are built on that one primitive.
//   * Fibers never run concurrently, so atomics are plain read-modify-writes, and everything between two
//     synchronisation points of a thread happens at once: one legal schedule among many.  EMU_REVERSE=1 runs
//     the threads from the last to the first, EMU_SHUFFLE=<seed> in a new random order for every stretch,
//     EMU_PREEMPT=<seed> makes every atomic operation a possible switch to another thread.
//   * Dynamic LDS is a guarded block of exactly the launch's size, filled with 0xA5 before every workgroup
//     (LDS is not zero on the hardware either); reads past it return 0 as on the hardware unless
//     EMU_STRICT_LDS=1, writes past it always fault.  `extern __shared__ T name[];` cannot be expressed on
//     the host: the build script rewrites that one line into `T* name = emu::dyn_lds<T>();`.
//     Static __shared__ variables become function-local statics (one workgroup runs at a time).
//   * Streams and events are ordinal: everything has happened when the call returns.
#pragma once
#include <algorithm>
#include <cassert>
#include <chrono>
#include <cmath>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <map>
#include <mutex>
#include <sys/mman.h>
#include <type_traits>
#include <ucontext.h>
#include <unistd.h>
#include <vector>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) (&(x))

// ---- types ----------------------------------------------------------------------------------------
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotReady = 600 };
struct ihipStream_t;
typedef ihipStream_t* hipStream_t;
struct emu_event {
    double t;
};
typedef emu_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
constexpr unsigned hipStreamNonBlocking = 1, hipHostMallocDefault = 0;

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

#define EMU_VEC2(name, T)                  \
    struct name {                          \
        T x, y;                            \
    } __attribute__((aligned(2 * sizeof(T)))); \
    static inline name make_##name(T x, T y) { return name{x, y}; }
#define EMU_VEC4(name, T, A)               \
    struct name {                          \
        T x, y, z, w;                      \
    } __attribute__((aligned(A)));         \
    static inline name make_##name(T x, T y, T z, T w) { return name{x, y, z, w}; }
EMU_VEC2(uint2, unsigned)
EMU_VEC2(int2, int)
EMU_VEC2(float2, float)
EMU_VEC2(double2, double)
EMU_VEC2(ulonglong2, unsigned long long)
EMU_VEC2(ushort2, unsigned short)
EMU_VEC4(uint4, unsigned, 16)
EMU_VEC4(int4, int, 16)
EMU_VEC4(float4, float, 16)
EMU_VEC4(short4, short, 8)
EMU_VEC4(ushort4, unsigned short, 8)
EMU_VEC4(uchar4, unsigned char, 4)

namespace emu {

typedef unsigned int u32;
typedef unsigned long long u64;

struct idx3 {
    unsigned x = 0, y = 0, z = 0;
};

// ---- guarded memory -------------------------------------------------------------------------------
struct block {
    char* map;        // start of the reserved address range
    size_t map_bytes;
    char* base;       // the caller's block
    size_t bytes;
    bool live;
    const char* what;
};
inline std::map<const void*, block>& blocks() {
    static std::map<const void*, block> b;
    return b;
}
inline std::recursive_mutex& big_lock() {
    static std::recursive_mutex m;
    return m;
}

inline void on_segv(int, siginfo_t* si, void*);
inline void install_handler() {
    static bool done = false;
    if (done) {
        return;
    }
    done = true;
    static char alt[1 << 16];
    stack_t ss;
    ss.ss_sp = alt;
    ss.ss_size = sizeof(alt);
    ss.ss_flags = 0;
    sigaltstack(&ss, nullptr);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = on_segv;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigaction(SIGSEGV, &sa, nullptr);
    sigaction(SIGBUS, &sa, nullptr);
}

// The block ends at a page boundary (an access just past it faults); 64 MiB before and 8 GiB after it are
// inaccessible, or readable zeros after it when `zeros_after` (reads past an LDS allocation).
inline void* guarded_alloc(size_t bytes, const char* what, int fill, bool zeros_after = false, size_t align = 256) {
    install_handler();
    const size_t page = 4096, before = 64ull << 20, after = 8ull << 30;
    const size_t body = (std::max<size_t>(bytes, 1) + page - 1) / page * page;
    char* m = static_cast<char*>(mmap(nullptr, before + body + after, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (m == MAP_FAILED) {
        return nullptr;
    }
    if (zeros_after && mprotect(m + before + body, after, PROT_READ) != 0) {
        munmap(m, before + body + after);
        return nullptr;
    }
    if (mprotect(m + before, body, PROT_READ | PROT_WRITE) != 0) {
        munmap(m, before + body + after);
        return nullptr;
    }
    memset(m + before, fill, body);
    // aligned as hipMalloc aligns (256); the tail of the last page stays accessible when the size is not a
    // multiple of the alignment
    char* blk = m + before + (body - (bytes + align - 1) / align * align);
    if (blk < m + before) {
        blk = m + before;
    }
    std::lock_guard<std::recursive_mutex> g(big_lock());
    blocks()[blk] = block{m, before + body + after, blk, bytes, true, what};
    return blk;
}
inline bool guarded_free(void* p) {
    std::lock_guard<std::recursive_mutex> g(big_lock());
    auto it = blocks().find(p);
    if (it == blocks().end() || !it->second.live) {
        return false;
    }
    // the block and 64 MiB on either side stay reserved and inaccessible: a use after free faults; the rest of the
    // range behind it goes back (a long run would otherwise use up the address space)
    const size_t body = (std::max<size_t>(it->second.bytes, 1) + 4095) / 4096 * 4096;
    const size_t keep = std::min<size_t>(it->second.map_bytes, ((size_t)64 << 20) + body + ((size_t)64 << 20)); // the guard in front is 64 MiB
    mprotect(it->second.map, keep, PROT_NONE);
    madvise(it->second.map, keep, MADV_DONTNEED);
    if (keep < it->second.map_bytes) {
        munmap(it->second.map + keep, it->second.map_bytes - keep);
        it->second.map_bytes = keep;
    }
    it->second.live = false;
    return true;
}

inline int g_cur = -1;
inline const char* g_kernel = "";
inline idx3 g_block_now;

inline void on_segv(int, siginfo_t* si, void*) {
    const char* a = static_cast<const char*>(si->si_addr);
    char line[320];
    int n = snprintf(line, sizeof(line), "emu: memory fault at %p in kernel %s, workgroup (%u, %u), thread %d\n", (void*)a, g_kernel, g_block_now.x,
                     g_block_now.y, g_cur);
    (void)!write(2, line, n);
    for (const auto& kv : blocks()) {
        const block& b = kv.second;
        if (a >= b.map && a < b.map + b.map_bytes) {
            n = snprintf(line, sizeof(line), "emu:   %lld bytes from the start of a %s block of %zu bytes%s\n", (long long)(a - b.base), b.what, b.bytes,
                         b.live ? "" : " (freed)");
            (void)!write(2, line, n);
        }
    }
    void* bt[48];
    const int k = backtrace(bt, 48);
    backtrace_symbols_fd(bt, k, 2);
    _exit(3);
}

// ---- fibers ------------------------------------------------------------------------------------------
enum state_t { READY, AT_BARRIER, AT_EXCHANGE, DONE };

struct fiber {
    ucontext_t ctx;
    state_t st = READY;
    long long site = 0;
    u64 value = 0;
};

struct snapshot {
    u64 val[64];
    u64 active = 0;
};

constexpr size_t kStack = 192 * 1024;
inline std::vector<fiber> g_fibers;
inline std::vector<char*> g_stacks;
inline std::vector<snapshot> g_snap; // one per wave
inline ucontext_t g_sched;
inline void (*g_body)(void*) = nullptr;
inline void* g_body_arg = nullptr;
inline char* g_dyn_lds = nullptr;
inline long long g_barriers = 0, g_exchanges = 0;

inline void yield_to_scheduler() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

inline void barrier() {
    g_fibers[g_cur].st = AT_BARRIER;
    yield_to_scheduler();
}

// EMU_PREEMPT=<seed>: every atomic operation is a possible switch to another thread (one time in four), so that
// what other threads do falls between a thread's atomics and the plain accesses around them, not only
// between its synchronisation points.
inline void preempt_point() {
    static const char* e = getenv("EMU_PREEMPT");
    static u64 rng = e ? strtoull(e, nullptr, 0) * 0x9E3779B97F4A7C15ull + 12345 : 0;
    if (!e || g_cur < 0) {
        return;
    }
    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
    if (((rng >> 40) & 3) == 0) {
        g_fibers[g_cur].st = READY;
        yield_to_scheduler();
    }
}

// Publishes v; returns the values and the set of the lanes of this wave that took part.
inline const snapshot& exchange(long long site, u64 v) {
    const int me = g_cur;
    fiber& f = g_fibers[me];
    f.site = site;
    f.value = v;
    f.st = AT_EXCHANGE;
    yield_to_scheduler();
    return g_snap[me >> 6];
}

inline void fiber_entry() {
    g_body(g_body_arg);
    g_fibers[g_cur].st = DONE;
    yield_to_scheduler();
}

template <typename T>
inline T* dyn_lds() {
    return reinterpret_cast<T*>(g_dyn_lds);
}

} // namespace emu

inline emu::idx3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

namespace emu {

[[noreturn]] inline void die(const char* msg) {
    fprintf(stderr, "emu: %s (kernel %s, workgroup (%u, %u), thread %d)\n", msg, g_kernel, g_block_now.x, g_block_now.y, g_cur);
    void* bt[32];
    const int k = backtrace(bt, 32);
    backtrace_symbols_fd(bt, k, 2);
    abort();
}

inline void run_workgroup(int nt, idx3 block_id) {
    const int nwaves = (nt + 63) / 64;
    if ((int)g_fibers.size() < nwaves * 64) {
        g_fibers.resize(nwaves * 64);
    }
    while ((int)g_stacks.size() < nt) {
        char* s = static_cast<char*>(mmap(nullptr, kStack + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (s == MAP_FAILED) {
            die("out of memory for fiber stacks");
        }
        mprotect(s, 4096, PROT_NONE); // a stack overflow faults instead of running into the next stack
        g_stacks.push_back(s + 4096);
    }
    g_snap.resize(nwaves);
    blockIdx = block_id;
    g_block_now = block_id;
    for (int i = 0; i < nwaves * 64; ++i) {
        fiber& f = g_fibers[i];
        if (i >= nt) {
            f.st = DONE; // lanes of a partial last wave do not exist
            continue;
        }
        f.st = READY;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = g_stacks[i];
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    static const bool reverse = [] {
        const char* e = getenv("EMU_REVERSE");
        return e && e[0] == '1';
    }();
    static const char* shuf = getenv("EMU_SHUFFLE");
    static u64 rng = shuf ? strtoull(shuf, nullptr, 0) * 2654435761ull + 1 : 0;
    std::vector<int> order(nt);
    for (int i = 0; i < nt; ++i) {
        order[i] = reverse ? nt - 1 - i : i;
    }
    int done = 0;
    while (done < nt) {
        if (shuf) {
            for (int i = nt - 1; i > 0; --i) {
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(order[i], order[(rng >> 33) % (u64)(i + 1)]);
            }
        }
        bool ran = false;
        for (int ii = 0; ii < nt; ++ii) {
            const int i = order[ii];
            if (g_fibers[i].st == READY) {
                g_cur = i;
                threadIdx.x = (unsigned)i % blockDim.x;
                threadIdx.y = ((unsigned)i / blockDim.x) % blockDim.y;
                threadIdx.z = (unsigned)i / (blockDim.x * blockDim.y);
                swapcontext(&g_sched, &g_fibers[i].ctx);
                ran = true;
                done += g_fibers[i].st == DONE;
            }
        }
        g_cur = -1;
        // exchanges: per wave, once all its lanes wait
        bool resolved = false;
        for (int w = 0; w < nwaves; ++w) {
            fiber* l = &g_fibers[w * 64];
            long long first = -1;
            bool all_wait = true;
            for (int k = 0; k < 64; ++k) {
                if (l[k].st == READY) {
                    all_wait = false;
                    break;
                }
                if (l[k].st == AT_EXCHANGE && (first < 0 || l[k].site < first)) {
                    first = l[k].site;
                }
            }
            if (!all_wait || first < 0) {
                continue;
            }
            snapshot& s = g_snap[w];
            s.active = 0;
            for (int k = 0; k < 64; ++k) {
                s.val[k] = 0;
                if (l[k].st == AT_EXCHANGE && l[k].site == first) {
                    s.val[k] = l[k].value;
                    s.active |= 1ull << k;
                    l[k].st = READY;
                }
            }
            ++g_exchanges;
            resolved = true;
        }
        if (resolved) {
            continue;
        }
        int at_barrier = 0;
        for (int i = 0; i < nt; ++i) {
            at_barrier += g_fibers[i].st == AT_BARRIER;
        }
        if (done == nt) {
            break;
        }
        if (at_barrier && at_barrier + done == nt) {
            // the hardware lets a barrier pass when the waves that have not ended have arrived; a wave that
            // ended early is fine, a wave of which only some lanes ended is too, as long as the rest arrives
            ++g_barriers;
            for (int i = 0; i < nt; ++i) {
                if (g_fibers[i].st == AT_BARRIER) {
                    g_fibers[i].st = READY;
                }
            }
            continue;
        }
        if (!ran) {
            die("deadlock: some threads wait at a barrier while others wait at a wave operation that cannot complete");
        }
    }
}

struct launch_state {
    std::map<const void*, size_t> max_dyn_lds; // hipFuncSetAttribute
};
inline launch_state& ls() {
    static launch_state s;
    return s;
}

template <typename F>
struct body_holder {
    F f;
    static void call(void* p) { static_cast<body_holder*>(p)->f(); }
};

template <typename... KArgs, typename... Args>
inline void launch(const char* name, void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t shmem, Args... args) {
    std::lock_guard<std::recursive_mutex> g(big_lock());
    const unsigned long long block_threads = (unsigned long long)block.x * block.y * block.z;
    if (block_threads == 0 || block_threads > 1024) {
        g_kernel = name;
        die("a workgroup has 1 to 1024 threads");
    }
    if (grid.x == 0 || grid.y == 0 || grid.z == 0) {
        g_kernel = name;
        die("empty grid (hipErrorInvalidConfiguration on the hardware)");
    }
    const void* fn = reinterpret_cast<const void*>(kernel);
    size_t allowed = 64 * 1024;
    auto it = ls().max_dyn_lds.find(fn);
    if (it != ls().max_dyn_lds.end()) {
        allowed = std::max(allowed, it->second);
    }
    g_kernel = name;
    if (shmem > allowed || shmem > 160 * 1024) {
        die("dynamic LDS beyond what the kernel has been granted (hipFuncSetAttribute) or beyond 160 KB");
    }
    static const bool strict = [] {
        const char* e = getenv("EMU_STRICT_LDS");
        return e && e[0] == '1';
    }();
    char* lds = shmem ? static_cast<char*>(guarded_alloc(shmem, "dynamic LDS", 0xA5, !strict, 16)) : nullptr;
    if (shmem && !lds) {
        die("cannot reserve address space for the LDS block");
    }
    auto fn_body = [&]() { kernel(static_cast<KArgs>(args)...); };
    body_holder<decltype(fn_body)> holder{fn_body};
    g_body = &body_holder<decltype(fn_body)>::call;
    g_body_arg = &holder;
    g_dyn_lds = lds;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz) {
        for (unsigned by = 0; by < grid.y; ++by) {
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                if (lds) {
                    memset(lds, 0xA5, shmem);
                }
                idx3 b;
                b.x = bx;
                b.y = by;
                b.z = bz;
                run_workgroup((int)block_threads, b);
            }
        }
    }
    if (lds) {
        std::lock_guard<std::recursive_mutex> g2(big_lock());
        auto bi = blocks().find(lds);
        munmap(bi->second.map, bi->second.map_bytes);
        blocks().erase(bi);
    }
    g_dyn_lds = nullptr;
    g_kernel = "";
}

// ---- wave operations on top of exchange() ------------------------------------------------------------
inline int lane_id() { return g_cur & 63; }

template <typename T>
inline u64 to_bits(T v) {
    static_assert(sizeof(T) <= 8, "value too wide for an exchange");
    u64 b = 0;
    memcpy(&b, &v, sizeof(T));
    return b;
}
template <typename T>
inline T from_bits(u64 b) {
    T v;
    memcpy(&v, &b, sizeof(T));
    return v;
}

inline u64 ballot(long long site, bool p) {
    const snapshot& s = exchange(site, p ? 1 : 0);
    u64 m = 0;
    for (int k = 0; k < 64; ++k) {
        if (((s.active >> k) & 1) && s.val[k]) {
            m |= 1ull << k;
        }
    }
    return m;
}

template <typename T>
inline T shfl_from(long long site, T v, int src_lane, const char* what) {
    const snapshot& s = exchange(site, to_bits(v));
    if (src_lane < 0 || src_lane >= 64 || !((s.active >> src_lane) & 1)) {
        fprintf(stderr, "emu: %s at line %lld reads lane %d, which does not take part\n", what, site & 0xFFFFF, src_lane);
        die("shuffle from an inactive lane");
    }
    return from_bits<T>(s.val[src_lane]);
}
template <typename T>
inline T shfl(long long site, T v, int src, int width = 64) {
    const int me = lane_id();
    return shfl_from(site, v, (me & ~(width - 1)) | (src & (width - 1)), "__shfl");
}
template <typename T>
inline T shfl_xor(long long site, T v, int mask, int width = 64) {
    const int me = lane_id();
    const int src = me ^ mask;
    // a source outside the lane's group of `width` returns the lane's own value
    return shfl_from(site, v, (src & ~(width - 1)) == (me & ~(width - 1)) ? src : me, "__shfl_xor");
}
template <typename T>
inline T shfl_down(long long site, T v, unsigned delta, int width = 64) {
    const int me = lane_id();
    const int src = me + (int)delta;
    return shfl_from(site, v, (src & ~(width - 1)) == (me & ~(width - 1)) ? src : me, "__shfl_down");
}
template <typename T>
inline T shfl_up(long long site, T v, unsigned delta, int width = 64) {
    const int me = lane_id();
    const int src = me - (int)delta;
    return shfl_from(site, v, src >= 0 && (src & ~(width - 1)) == (me & ~(width - 1)) ? src : me, "__shfl_up");
}
inline int readlane(long long site, int v, int lane) { return shfl_from(site, v, lane, "readlane"); }
inline int readfirstlane(long long site, int v) {
    const snapshot& s = exchange(site, to_bits(v));
    return from_bits<int>(s.val[__builtin_ctzll(s.active)]);
}
inline void wave_barrier(long long site) { (void)exchange(site, 0); }

// v_mov_b32 with a DPP operand: `src` of another lane, or `old` when the lane is not written
inline int update_dpp(long long site, int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const snapshot& s = exchange(site, to_bits(src));
    const int me = lane_id(), row = me >> 4, in_row = me & 15;
    if (!((row_mask >> row) & 1) || !((bank_mask >> (in_row >> 2)) & 1)) {
        return old;
    }
    int from = -1; // -1: no source lane
    if (ctrl >= 0 && ctrl <= 0xFF) { // quad_perm
        from = (me & ~3) | ((ctrl >> (2 * (me & 3))) & 3);
    }
    else if (ctrl >= 0x101 && ctrl <= 0x10F) { // row_shl
        const int k = in_row + (ctrl & 15);
        from = k < 16 ? (me & ~15) | k : -1;
    }
    else if (ctrl >= 0x111 && ctrl <= 0x11F) { // row_shr
        const int k = in_row - (ctrl & 15);
        from = k >= 0 ? (me & ~15) | k : -1;
    }
    else if (ctrl >= 0x121 && ctrl <= 0x12F) { // row_ror
        from = (me & ~15) | ((in_row - (ctrl & 15)) & 15);
    }
    else if (ctrl == 0x130) { // wave_shl:1
        from = me + 1 < 64 ? me + 1 : -1;
    }
    else if (ctrl == 0x134) { // wave_rol:1
        from = (me + 1) & 63;
    }
    else if (ctrl == 0x138) { // wave_shr:1
        from = me - 1;
    }
    else if (ctrl == 0x13C) { // wave_ror:1
        from = (me - 1) & 63;
    }
    else if (ctrl == 0x140) { // row_mirror
        from = (me & ~15) | (15 - in_row);
    }
    else if (ctrl == 0x141) { // row_half_mirror
        from = (me & ~7) | (7 - (me & 7));
    }
    else if (ctrl == 0x142) { // row_bcast:15 -- lane 15 of a row to the next row
        from = row >= 1 ? 16 * row - 1 : -1;
    }
    else if (ctrl == 0x143) { // row_bcast:31 -- lane 31 to rows 2 and 3
        from = row >= 2 ? 31 : -1;
    }
    else {
        die("DPP control not emulated");
    }
    if (from < 0 || !((s.active >> from) & 1)) {
        return bound_ctrl ? 0 : old;
    }
    return from_bits<int>(s.val[from]);
}

// the site of a wave operation: helpers in headers sort before the translation unit's own lines
constexpr long long site_of(const char* file, int line) {
    int n = 0;
    while (file[n]) {
        ++n;
    }
    const bool header = n >= 2 && file[n - 2] == '.' && file[n - 1] == 'h';
    return (header ? 0ll : 1ll << 24) | line;
}

} // namespace emu

#define EMU_SITE emu::site_of(__FILE__, __LINE__)
#define __syncthreads() emu::barrier()
#define __ballot(p) emu::ballot(EMU_SITE, (p))
#define __shfl(...) emu::shfl(EMU_SITE, __VA_ARGS__)
#define __shfl_xor(...) emu::shfl_xor(EMU_SITE, __VA_ARGS__)
#define __shfl_down(...) emu::shfl_down(EMU_SITE, __VA_ARGS__)
#define __shfl_up(...) emu::shfl_up(EMU_SITE, __VA_ARGS__)
#define __builtin_amdgcn_readlane(v, lane) emu::readlane(EMU_SITE, (v), (lane))
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane(EMU_SITE, (v))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu::update_dpp(EMU_SITE, (old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier(EMU_SITE)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_barrier() emu::barrier()
#define __threadfence() ((void)0)
#define __threadfence_block() ((void)0)

// ---- device functions -----------------------------------------------------------------------------------
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline unsigned __brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) {
        r |= ((v >> i) & 1u) << (31 - i);
    }
    return r;
}
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __float_as_uint(float f) { return emu::from_bits<unsigned>(emu::to_bits(f)); }
static inline int __float_as_int(float f) { return emu::from_bits<int>(emu::to_bits(f)); }
static inline float __uint_as_float(unsigned u) { return emu::from_bits<float>(emu::to_bits(u)); }
static inline float __int_as_float(int u) { return emu::from_bits<float>(emu::to_bits(u)); }
static inline long long __double_as_longlong(double d) { return emu::from_bits<long long>(emu::to_bits(d)); }
static inline double __longlong_as_double(long long v) { return emu::from_bits<double>(emu::to_bits(v)); }
// IEEE operations without contraction (the build passes -ffp-contract=off)
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline double __dsqrt_rn(double a) { return sqrt(a); }
static inline double __fma_rn(double a, double b, double c) { return fma(a, b, c); }
static inline int __float2int_rn(float v) { return (int)nearbyintf(v); }
static inline int __float2int_rd(float v) { return (int)floorf(v); }
static inline int __float2int_ru(float v) { return (int)ceilf(v); }
static inline int __float2int_rz(float v) { return (int)v; }
static inline int __double2int_rn(double v) { return (int)nearbyint(v); }
static inline float __int2float_rn(int v) { return (float)v; }
static inline float __uint2float_rn(unsigned v) { return (float)v; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (8 * (sh & 3)));
}

static inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) {
    return (unsigned)(((((unsigned long long)hi) << 32) | lo) >> (sh & 31));
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
// v_perm_b32 D, S0 = hi, S1 = lo, selector: bytes 0-3 of the selector pick, for result bytes 0-3, byte 0-3 of lo,
// 4-7 = byte 0-3 of hi, 8-11 = the sign of lo byte 1 / lo byte 3 / hi byte 1 / hi byte 3 replicated, 12 = 0x00, 13+ = 0xFF
static inline unsigned __builtin_amdgcn_perm(unsigned hi, unsigned lo, unsigned sel) {
    const unsigned long long both = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned s = (sel >> (8 * i)) & 0xFF;
        unsigned b;
        if (s <= 7) {
            b = (unsigned)(both >> (8 * s)) & 0xFF;
        }
        else if (s <= 11) {
            const int bit = s == 8 ? 15 : (s == 9 ? 31 : (s == 10 ? 47 : 63));
            b = ((both >> bit) & 1) ? 0xFF : 0x00;
        }
        else {
            b = s == 12 ? 0x00 : 0xFF;
        }
        r |= b << (8 * i);
    }
    return r;
}
static inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool clamp) {
    unsigned long long acc = c;
    for (int i = 0; i < 4; ++i) {
        acc += (unsigned long long)((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    }
    return clamp && acc > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)acc;
}
static inline int __builtin_amdgcn_sdot4(int a, int b, int c, bool clamp) {
    long long acc = c;
    for (int i = 0; i < 4; ++i) {
        acc += (long long)(signed char)((unsigned)a >> (8 * i)) * (signed char)((unsigned)b >> (8 * i));
    }
    if (clamp) {
        acc = acc > 2147483647ll ? 2147483647ll : (acc < -2147483648ll ? -2147483648ll : acc);
    }
    return (int)acc;
}

#if defined(__clang__)
typedef unsigned short emu_u16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int emu_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int emu_u32x4 __attribute__((ext_vector_type(4)));
typedef int emu_i32x4 __attribute__((ext_vector_type(4)));
static inline unsigned __builtin_amdgcn_udot2(emu_u16x2 a, emu_u16x2 b, unsigned c, bool clamp) {
    const unsigned long long acc = (unsigned long long)c + (unsigned long long)a.x * b.x + (unsigned long long)a.y * b.y;
    return clamp && acc > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)acc;
}

// Buffer resources: a base address and a number of bytes; a raw (unstrided) access is range-checked per dword,
// a load out of range returns 0 and a store out of range is dropped, as on the hardware.  In-range addresses
// still have to lie in accessible memory.
struct __amdgpu_buffer_rsrc_t {
    unsigned char* base;
    unsigned num_records;
};
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short stride, int num, int) {
    if (stride != 0) {
        emu::die("strided buffer resources are not emulated");
    }
    return __amdgpu_buffer_rsrc_t{static_cast<unsigned char*>(p), (unsigned)num};
}
static inline unsigned emu_buffer_dword(const __amdgpu_buffer_rsrc_t& r, unsigned off) {
    if ((unsigned long long)off + 4 > r.num_records) {
        return 0;
    }
    unsigned v;
    memcpy(&v, r.base + off, 4);
    return v;
}
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    return emu_buffer_dword(r, voff + soff);
}
static inline emu_u32x3 __builtin_amdgcn_raw_buffer_load_b96(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    const unsigned o = voff + soff;
    return emu_u32x3{emu_buffer_dword(r, o), emu_buffer_dword(r, o + 4), emu_buffer_dword(r, o + 8)};
}
static inline emu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    const unsigned o = voff + soff;
    return emu_u32x4{emu_buffer_dword(r, o), emu_buffer_dword(r, o + 4), emu_buffer_dword(r, o + 8), emu_buffer_dword(r, o + 12)};
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, int) {
    const unsigned o = voff + soff;
    if ((unsigned long long)o + 4 <= r.num_records) {
        memcpy(r.base + o, &v, 4);
    }
}

// v_mfma_i32_16x16x64_i8: D (16 x 16) = A (16 x 64) * B (64 x 16) + C.  Lane l holds row l % 16 of A and
// column l % 16 of B, sixteen consecutive k each (k = 16 * (l / 16) + byte), and rows 4 * (l / 16) + r of
// column l % 16 of C / D.  (Which sixteen k a lane holds is the same for A and B, so a sum over k does not
// depend on it.)  Two exchanges: operands, then "everybody has read them".
namespace emu {
struct mfma_ops {
    emu_i32x4 a, b;
};
inline emu_i32x4 mfma_i32_16x16x64_i8(long long site, emu_i32x4 a, emu_i32x4 b, emu_i32x4 c) {
    mfma_ops mine{a, b};
    const snapshot& s = exchange(site, (u64)(uintptr_t)&mine);
    if (s.active != ~0ull) {
        die("MFMA with inactive lanes");
    }
    const int me = lane_id(), col = me & 15, rg = me >> 4;
    emu_i32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rg + r;
        int acc = 0;
        for (int g = 0; g < 4; ++g) {
            const mfma_ops* pa = reinterpret_cast<const mfma_ops*>((uintptr_t)s.val[16 * g + row]);
            const mfma_ops* pb = reinterpret_cast<const mfma_ops*>((uintptr_t)s.val[16 * g + col]);
            const signed char* ab = reinterpret_cast<const signed char*>(&pa->a);
            const signed char* bb = reinterpret_cast<const signed char*>(&pb->b);
            for (int k = 0; k < 16; ++k) {
                acc += (int)ab[k] * (int)bb[k];
            }
        }
        d[r] += acc;
    }
    (void)exchange(site | (1ll << 30), 0);
    return d;
}
} // namespace emu
#define __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c, cbsz, abid, blgp) emu::mfma_i32_16x16x64_i8(EMU_SITE, (a), (b), (c))
#endif

#define EMU_MINMAX(T)                                      \
    static inline T min(T a, T b) { return b < a ? b : a; } \
    static inline T max(T a, T b) { return a < b ? b : a; }
EMU_MINMAX(int)
EMU_MINMAX(unsigned)
EMU_MINMAX(long)
EMU_MINMAX(unsigned long)
EMU_MINMAX(long long)
EMU_MINMAX(unsigned long long)
EMU_MINMAX(float)
EMU_MINMAX(double)
// mixed signedness converts to unsigned, as the HIP headers do
static inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
static inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }
static inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }
static inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }

template <typename T>
static inline T atomicAdd(T* p, T v) {
    emu::preempt_point();
    const T o = *p;
    *p = o + v;
    return o;
}
template <typename T>
static inline T atomicSub(T* p, T v) {
    emu::preempt_point();
    const T o = *p;
    *p = o - v;
    return o;
}
template <typename T>
static inline T atomicMin(T* p, T v) {
    emu::preempt_point();
    const T o = *p;
    *p = v < o ? v : o;
    return o;
}
template <typename T>
static inline T atomicMax(T* p, T v) {
    emu::preempt_point();
    const T o = *p;
    *p = o < v ? v : o;
    return o;
}
template <typename T>
static inline T atomicOr(T* p, T v) {
    emu::preempt_point();
    const T o = *p;
    *p = o | v;
    return o;
}
template <typename T>
static inline T atomicAnd(T* p, T v) {
    emu::preempt_point();
    const T o = *p;
    *p = o & v;
    return o;
}
template <typename T>
static inline T atomicExch(T* p, T v) {
    emu::preempt_point();
    const T o = *p;
    *p = v;
    return o;
}
template <typename T>
static inline T atomicCAS(T* p, T expected, T v) {
    emu::preempt_point();
    const T o = *p;
    if (o == expected) {
        *p = v;
    }
    return o;
}

// ---- runtime API ----------------------------------------------------------------------------------------
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) {
    *n = 1;
    return hipSuccess;
}
static inline hipError_t hipGetDevice(int* d) {
    *d = 0;
    return hipSuccess;
}
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t bytes) {
    *p = emu::guarded_alloc(bytes, "hipMalloc", 0xCD);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
template <typename T>
static inline hipError_t hipMalloc(T** p, size_t bytes) {
    return hipMalloc(reinterpret_cast<void**>(p), bytes);
}
static inline hipError_t hipFree(void* p) {
    if (!p) {
        return hipSuccess;
    }
    return emu::guarded_free(p) ? hipSuccess : hipErrorInvalidValue;
}
static inline hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    *p = emu::guarded_alloc(bytes, "hipHostMalloc", 0xCE);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipHostFree(void* p) { return hipFree(p); }
static inline hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
    std::lock_guard<std::recursive_mutex> g(emu::big_lock());
    memmove(dst, src, bytes);
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t) { return hipMemcpy(dst, src, bytes, k); }
static inline hipError_t hipMemset(void* dst, int v, size_t bytes) {
    std::lock_guard<std::recursive_mutex> g(emu::big_lock());
    memset(dst, v, bytes);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t) { return hipMemset(dst, v, bytes); }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) {
    *s = reinterpret_cast<hipStream_t>(new char);
    return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
static inline hipError_t hipStreamDestroy(hipStream_t s) {
    delete reinterpret_cast<char*>(s);
    return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) {
    *e = new emu_event{0.0};
    return hipSuccess;
}
static inline hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = (float)(b->t - a->t);
    return hipSuccess;
}
constexpr unsigned hipEventDisableTiming = 2;
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipGetSymbolAddress(void** p, const void* symbol) {
    *p = const_cast<void*>(symbol);
    return hipSuccess;
}
static inline hipError_t hipMemcpyToSymbol(const void* symbol, const void* src, size_t bytes, size_t offset = 0, hipMemcpyKind = hipMemcpyHostToDevice) {
    std::lock_guard<std::recursive_mutex> g(emu::big_lock());
    memcpy(static_cast<char*>(const_cast<void*>(symbol)) + offset, src, bytes);
    return hipSuccess;
}
static inline hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    std::lock_guard<std::recursive_mutex> g(emu::big_lock());
    for (size_t y = 0; y < height; ++y) {
        memmove(static_cast<char*>(dst) + y * dpitch, static_cast<const char*>(src) + y * spitch, width);
    }
    return hipSuccess;
}
static inline hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t) {
    return hipMemcpy2D(dst, dpitch, src, spitch, width, height, k);
}
static inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) {
    *free_b = *total_b = (size_t)256 << 30;
    return hipSuccess;
}
static inline hipError_t hipFuncSetAttribute(const void* fn, hipFuncAttribute attr, int value) {
    if (attr != hipFuncAttributeMaxDynamicSharedMemorySize || value < 0 || value > 160 * 1024) {
        return hipErrorInvalidValue;
    }
    std::lock_guard<std::recursive_mutex> g(emu::big_lock());
    emu::ls().max_dyn_lds[fn] = (size_t)value;
    return hipSuccess;
}

// for test programs that hand "device" arrays to the library: the same guarded blocks hipMalloc gives
extern "C" __attribute__((used, visibility("default"))) inline void* ovs_emu_guarded_alloc(size_t bytes, int fill) {
    return emu::guarded_alloc(bytes, "test array", fill);
}
extern "C" __attribute__((used, visibility("default"))) inline int ovs_emu_guarded_free(void* p) { return emu::guarded_free(p) ? 0 : -1; }

#define EMU_STR2(x) #x
#define EMU_STR(x) EMU_STR2(x)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch(EMU_STR(kernel), kernel, dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)
