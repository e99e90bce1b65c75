// Single-workgroup SIMT emulator for host-side debugging of a HIP kernel's control logic.
//
// Every thread of the workgroup is a ucontext fiber on one OS thread.  A fiber runs until it reaches
// a workgroup barrier, a wave operation (ballot / readlane / inclusive scan) or its end; the scheduler
// then resolves what can be resolved:
//   * a wave operation completes when every lane of the wave waits at a wave operation, waits at a
//     barrier or has finished: the lanes at the operation are the wave's active lanes (a lane that
//     left a loop earlier is parked at the following barrier, exactly the lanes the hardware masks off);
//   * a barrier completes when every unfinished thread waits at it.
// Fibers never run concurrently, so atomics are plain read-modify-writes.  The emulator reproduces
// index arithmetic and data flow, not memory-model races: a schedule in which every thread runs from
// one synchronisation point to the next without interruption is one legal GPU schedule among many.
//
// Memory the kernel indexes (the "LDS" block and global buffers) can be taken from guarded_alloc(),
// which surrounds the block with inaccessible address space (32 GiB after it, 1 GiB before it): an
// access through a wild 32-bit index ends in SIGSEGV, and the handler prints the offset and a backtrace.
// A block that stands for LDS can instead be followed by readable zeros, as the hardware answers reads
// past a workgroup's LDS allocation.
// EMU_REVERSE=1 in the environment runs the threads of a workgroup from the last to the first between
// synchronisation points (another legal schedule: what depends on the order shows up as a difference);
// EMU_SHUFFLE=<seed> draws a new random order of the threads for every such stretch.
#pragma once
#include <algorithm>
#include <cassert>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <functional>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>
#include <vector>

typedef unsigned int u32;
typedef unsigned long long u64;

struct emu_dim3 {
    unsigned x = 1, y = 1, z = 1;
};
static emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

struct short4 {
    short x, y, z, w;
} __attribute__((aligned(8)));
struct uint2 {
    u32 x, y;
} __attribute__((aligned(8)));
static inline uint2 make_uint2(u32 x, u32 y) {
    uint2 r;
    r.x = x;
    r.y = y;
    return r;
}
using std::max;
using std::min;

#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_readlane(v, lane) emu::readlane((v), (lane))

namespace emu {

enum state_t { READY, AT_BARRIER, AT_WAVEOP, DONE };
enum op_t { OP_NONE, OP_BALLOT, OP_READLANE, OP_SCAN };

struct fiber {
    ucontext_t ctx;
    state_t st = READY;
    op_t op = OP_NONE;
    long long arg = 0, arg2 = 0, result = 0;
    char* stack = nullptr;
};

static std::vector<fiber> g_fibers;
static ucontext_t g_sched;
static int g_cur = -1;
static std::function<void()> g_body;
static long long g_barriers = 0, g_waveops = 0;

static inline void yield_to_scheduler() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

static inline void barrier() {
    g_fibers[g_cur].st = AT_BARRIER;
    yield_to_scheduler();
}
static inline long long waveop(op_t op, long long a, long long b) {
    fiber& f = g_fibers[g_cur];
    f.op = op;
    f.arg = a;
    f.arg2 = b;
    f.st = AT_WAVEOP;
    yield_to_scheduler();
    return g_fibers[g_cur].result;
}
static inline u64 ballot(bool p) { return (u64)waveop(OP_BALLOT, p ? 1 : 0, 0); }
static inline int readlane(int v, int lane) { return (int)waveop(OP_READLANE, v, lane); }
static inline int incl_scan(int v) { return (int)waveop(OP_SCAN, v, 0); }

static void fiber_entry() {
    g_body();
    g_fibers[g_cur].st = DONE;
    yield_to_scheduler();
}

// Runs body() once per thread of a workgroup of nt threads (nt a multiple of 64).
static void run_workgroup(int nt, emu_dim3 block_id, emu_dim3 grid, std::function<void()> body) {
    constexpr size_t kStack = 256 * 1024;
    g_body = std::move(body);
    g_fibers.assign(nt, fiber());
    blockIdx = block_id;
    gridDim = grid;
    blockDim.x = nt;
    for (int i = 0; i < nt; ++i) {
        fiber& f = g_fibers[i];
        f.stack = static_cast<char*>(mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStack;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, fiber_entry, 0);
    }
    const int nwaves = nt / 64;
    const char* rev = getenv("EMU_REVERSE");
    const bool reverse = rev && rev[0] == '1';
    const char* shuf = getenv("EMU_SHUFFLE");
    u64 rng = shuf ? strtoull(shuf, nullptr, 0) * 2654435761ull + 1 : 0;
    std::vector<int> order(nt);
    for (int i = 0; i < nt; ++i) {
        order[i] = reverse ? nt - 1 - i : i;
    }
    while (true) {
        bool ran = false;
        if (shuf) {
            for (int i = nt - 1; i > 0; --i) {
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                std::swap(order[i], order[(rng >> 33) % (u64)(i + 1)]);
            }
        }
        for (int ii = 0; ii < nt; ++ii) {
            const int i = order[ii];
            if (g_fibers[i].st == READY) {
                g_cur = i;
                threadIdx.x = i;
                swapcontext(&g_sched, &g_fibers[i].ctx);
                ran = true;
            }
        }
        // wave operations
        bool resolved = false;
        for (int w = 0; w < nwaves; ++w) {
            fiber* l = &g_fibers[w * 64];
            int waiting = 0;
            op_t op = OP_NONE;
            for (int k = 0; k < 64; ++k) {
                if (l[k].st == AT_WAVEOP) {
                    ++waiting;
                    if (op != OP_NONE && op != l[k].op) {
                        fprintf(stderr, "emu: lanes of wave %d wait at different wave operations\n", w);
                        abort();
                    }
                    op = l[k].op;
                }
            }
            if (!waiting) {
                continue;
            }
            ++g_waveops;
            if (op == OP_BALLOT) {
                u64 m = 0;
                for (int k = 0; k < 64; ++k) {
                    if (l[k].st == AT_WAVEOP && l[k].arg) {
                        m |= 1ull << k;
                    }
                }
                for (int k = 0; k < 64; ++k) {
                    l[k].result = (long long)m;
                }
            }
            else if (op == OP_READLANE) {
                for (int k = 0; k < 64; ++k) {
                    if (l[k].st == AT_WAVEOP) {
                        const int src = (int)l[k].arg2;
                        if (src < 0 || src >= 64 || l[src].st != AT_WAVEOP) {
                            fprintf(stderr, "emu: readlane from inactive lane %d (wave %d)\n", src, w);
                            abort();
                        }
                        l[k].result = l[src].arg;
                    }
                }
            }
            else {
                long long run = 0;
                for (int k = 0; k < 64; ++k) {
                    if (l[k].st == AT_WAVEOP) {
                        run += l[k].arg;
                        l[k].result = run;
                    }
                }
            }
            for (int k = 0; k < 64; ++k) {
                if (l[k].st == AT_WAVEOP) {
                    l[k].st = READY;
                    l[k].op = OP_NONE;
                }
            }
            resolved = true;
        }
        if (resolved) {
            continue;
        }
        int at_barrier = 0, done = 0;
        for (int i = 0; i < nt; ++i) {
            at_barrier += g_fibers[i].st == AT_BARRIER;
            done += g_fibers[i].st == DONE;
        }
        if (done == nt) {
            break;
        }
        if (at_barrier && at_barrier + done == nt) {
            // (a barrier some threads never reach because they returned is a hang on the hardware
            // unless whole waves returned; the kernels emulated here return uniformly)
            if (done) {
                fprintf(stderr, "emu: %d threads wait at a barrier that %d finished threads never reach\n", at_barrier, done);
                abort();
            }
            ++g_barriers;
            for (int i = 0; i < nt; ++i) {
                g_fibers[i].st = READY;
            }
            continue;
        }
        if (!ran) {
            fprintf(stderr, "emu: deadlock\n");
            abort();
        }
    }
    for (int i = 0; i < nt; ++i) {
        munmap(g_fibers[i].stack, kStack);
    }
    g_fibers.clear();
}

// ---- guarded memory ------------------------------------------------------------------------------
struct guarded_block {
    char* base;
    size_t bytes;
    const char* name;
};
static std::vector<guarded_block> g_blocks;

// lds_reads: the block stands for LDS, where the hardware answers a read past the allocation with 0
// and drops a write: the space after the block is readable zero pages (a write there still faults).
static void* guarded_alloc(size_t bytes, const char* name, int fill, bool lds_reads = false) {
    const size_t page = 4096, before = 1ull << 30, after = 40ull << 30;
    const size_t body = (bytes + page - 1) / page * page;
    char* m = static_cast<char*>(mmap(nullptr, before + body + after, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (m == MAP_FAILED) {
        perror("mmap");
        exit(2);
    }
    if (lds_reads && mprotect(m + before + body, after, PROT_READ) != 0) {
        perror("mprotect");
        exit(2);
    }
    // the block ends at a page boundary so that an access just past it faults too
    char* blk = m + before + (body - bytes);
    if (mprotect(m + before, body, PROT_READ | PROT_WRITE) != 0) {
        perror("mprotect");
        exit(2);
    }
    memset(m + before, fill, body);
    g_blocks.push_back({blk, bytes, name});
    return blk;
}

static void on_segv(int, siginfo_t* si, void*) {
    const char* a = static_cast<const char*>(si->si_addr);
    char line[256];
    int n = snprintf(line, sizeof(line), "emu: SIGSEGV at %p, thread %d of the workgroup\n", (void*)a, g_cur);
    (void)!write(2, line, n);
    for (const auto& b : g_blocks) {
        const long long off = a - b.base;
        if (off > -(1ll << 30) && off < (41ll << 30)) {
            n = snprintf(line, sizeof(line), "emu:   %lld bytes from the start of '%s' (%zu bytes long)\n", off, b.name, b.bytes);
            (void)!write(2, line, n);
        }
    }
    void* bt[48];
    const int k = backtrace(bt, 48);
    backtrace_symbols_fd(bt, k, 2);
    _exit(3);
}

static void install_segv_handler() {
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

} // namespace emu

static inline void __syncthreads() { emu::barrier(); }
static inline u64 __ballot(bool p) { return emu::ballot(p); }
static inline int wave_incl_scan(int v) { return emu::incl_scan(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __popcll(u64 v) { return __builtin_popcountll(v); }
static inline u32 atomicAdd(u32* p, u32 v) {
    const u32 o = *p;
    *p = o + v;
    return o;
}
static inline u32 atomicMin(u32* p, u32 v) {
    const u32 o = *p;
    *p = std::min(o, v);
    return o;
}
static inline u32 atomicMax(u32* p, u32 v) {
    const u32 o = *p;
    *p = std::max(o, v);
    return o;
}
