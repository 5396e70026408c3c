// TEST INFRASTRUCTURE (not product code): a host-side execution model of one gfx950 workgroup, so that the HIP kernel
// sources of csrc/ can be compiled for the CPU -- the SAME source text, under -DTMPC_HOST_SIM -- and run under
// AddressSanitizer / UndefinedBehaviorSanitizer / MemorySanitizer (the GPU pool offers no sanitizer).
//
// Model.  Every lane of every wave of the workgroup is a fiber (ucontext) that runs the kernel function as plain scalar C++.
// What makes a wavefront a wavefront is enforced at the points where lanes meet:
//   * cross-lane operations (v_readlane, DPP moves, ballots, MFMA) are rendezvous of the 64 fibers of a wave: each lane
//     deposits its operand, waits for the others, and computes its own result from the deposited operands with the lane
//     maps of the ISA (DPP controls quad_perm / row_mirror / row_half_mirror / row_newbcast; v_mfma_f64_16x16x4_f64:
//     A[i][k] on lane i + 16 k, B[k][j] on lane j + 16 k, D rows (lane >> 4) + 4 reg, column lane & 15);
//   * the kernels' wave-level LDS fences (`asm volatile("" ::: "memory")` on the device: lockstep execution orders a wave's
//     LDS traffic, the statement only stops the compiler) are rendezvous as well.  Between two rendezvous a lane may run
//     arbitrarily far ahead of the others, so an LDS hand-over between lanes that is NOT separated by a fence or a
//     cross-lane operation reads stale data here -- exactly the hand-overs that a compiler is free to break on the device;
//   * __syncthreads() is a rendezvous of all fibers of the workgroup.
// A rendezvous that not every lane reaches (divergent control flow around a cross-lane operation: undefined on the
// device) is reported as a deadlock, and so are lanes that meet in different operations.
// LDS is one heap block (red zones of ASan on both sides); device memory is host memory.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define TMPC_SIM_ASAN 1
#endif
#endif
#if defined(__SANITIZE_ADDRESS__)
#define TMPC_SIM_ASAN 1
#endif
#ifdef TMPC_SIM_ASAN
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#endif

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

namespace sim {

constexpr int WAVE = 64;
struct Dim3 { unsigned x = 1, y = 1, z = 1; };

enum Op : int { OP_NONE, OP_FENCE, OP_READLANE, OP_DPP32, OP_DPP64, OP_BALLOT, OP_MFMA, OP_SYNC, OP_SYNC_OR };

struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    int tid = 0;
    bool done = false;
#ifdef TMPC_SIM_ASAN
    void *fake = nullptr;
#endif
};

struct Slot { double d[6]; };       // operand deposit of one lane (MFMA: a, b)

struct WaveState {
    unsigned long gen = 0;
    int arrived = 0;
    int op[2] = {OP_NONE, OP_NONE};
    Slot slot[2][WAVE];
};

struct Block {
    int nthreads = 0, nwaves = 0;
    Dim3 block_idx, grid_dim, block_dim;
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    unsigned long bgen = 0;         // __syncthreads generation
    int barrived = 0;
    int bor = 0, bor_result[2] = {0, 0};
    char *lds = nullptr;
    size_t lds_bytes = 0;
    ucontext_t sched;
    int current = -1;
    unsigned long progress = 0;     // bumped by every completed rendezvous / finished fiber (deadlock detection)
    std::function<void()> body;
    unsigned long rendezvous = 0;
#ifdef TMPC_SIM_ASAN
    void *sched_fake = nullptr;
    const void *sched_stack_bottom = nullptr;
    size_t sched_stack_size = 0;
#endif
};

inline Block *&blk() { static Block *b = nullptr; return b; }
inline int cur_tid() { return blk()->current; }
inline int cur_lane() { return blk()->current & (WAVE - 1); }
inline int cur_wave() { return blk()->current >> 6; }

constexpr size_t STACK_BYTES = 256 * 1024;

inline void to_scheduler() {
    Block *b = blk();
    Fiber &f = b->fibers[b->current];
#ifdef TMPC_SIM_ASAN
    __sanitizer_start_switch_fiber(f.done ? nullptr : &f.fake, b->sched_stack_bottom, b->sched_stack_size);
#endif
    swapcontext(&f.ctx, &b->sched);
#ifdef TMPC_SIM_ASAN
    __sanitizer_finish_switch_fiber(f.fake, &b->sched_stack_bottom, &b->sched_stack_size);
#endif
}

[[noreturn]] inline void die(const char *what) {
    std::fprintf(stderr, "wavesim: %s (thread %d)\n", what, blk() ? blk()->current : -1);
    std::abort();
}

// rendezvous of the calling lane's wave: deposits `in`, returns the buffer that holds all 64 deposits
inline const Slot *wave_meet(int op, const Slot &in) {
    Block *b = blk();
    WaveState &w = b->waves[cur_wave()];
    const unsigned long my = w.gen;
    const int par = static_cast<int>(my & 1ul);
    if (w.arrived == 0) w.op[par] = op;
    else if (w.op[par] != op) die("the lanes of a wave meet in different cross-lane operations (divergent control flow)");
    w.slot[par][cur_lane()] = in;
    if (++w.arrived == WAVE) {
        w.arrived = 0;
        ++w.gen;
        ++b->progress;
        ++b->rendezvous;
    } else {
        while (w.gen == my) to_scheduler();
    }
    return w.slot[par];
}

inline void block_meet(int orv, int *or_out) {
    Block *b = blk();
    const unsigned long my = b->bgen;
    const int par = static_cast<int>(my & 1ul);
    if (b->barrived == 0) b->bor = 0;
    b->bor |= orv;
    if (++b->barrived == b->nthreads) {
        b->barrived = 0;
        b->bor_result[par] = b->bor;
        ++b->bgen;
        ++b->progress;
    } else {
        while (b->bgen == my) to_scheduler();
    }
    if (or_out) *or_out = b->bor_result[par];
}

inline void fiber_entry() {
    Block *b = blk();
#ifdef TMPC_SIM_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &b->sched_stack_bottom, &b->sched_stack_size);
#endif
    b->body();
    b->fibers[b->current].done = true;
    ++b->progress;
    to_scheduler();
    die("a finished fiber was resumed");
}

// Runs one workgroup of `nthreads` threads with `lds_bytes` of LDS; `body` is the kernel call.
inline unsigned long run_block(int nthreads, size_t lds_bytes, Dim3 block_idx, Dim3 grid_dim, const std::function<void()> &body) {
    if (nthreads % WAVE) die("workgroup size must be a multiple of 64");
    Block b;
    b.nthreads = nthreads;
    b.nwaves = nthreads / WAVE;
    b.block_idx = block_idx;
    b.grid_dim = grid_dim;
    b.block_dim.x = static_cast<unsigned>(nthreads);
    b.fibers.resize(nthreads);
    b.waves.resize(b.nwaves);
    b.lds_bytes = lds_bytes;
    b.lds = static_cast<char *>(std::malloc(lds_bytes ? lds_bytes : 16));        // uninitialised on purpose (MSan)
    b.body = body;
    blk() = &b;
    for (int t = 0; t < nthreads; ++t) {
        Fiber &f = b.fibers[t];
        f.tid = t;
        f.stack = static_cast<char *>(std::malloc(STACK_BYTES));
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STACK_BYTES;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, reinterpret_cast<void (*)()>(fiber_entry), 0);
    }
    int live = nthreads;
    unsigned long last_progress = 0;
    int idle_rounds = 0;
    while (live > 0) {
        live = 0;
        for (int t = 0; t < nthreads; ++t) {
            Fiber &f = b.fibers[t];
            if (f.done) continue;
            ++live;
            b.current = t;
#ifdef TMPC_SIM_ASAN
            __sanitizer_start_switch_fiber(&b.sched_fake, f.stack, STACK_BYTES);
#endif
            swapcontext(&b.sched, &f.ctx);
#ifdef TMPC_SIM_ASAN
            __sanitizer_finish_switch_fiber(b.sched_fake, nullptr, nullptr);
#endif
        }
        if (b.progress == last_progress) {
            if (++idle_rounds > 2) { b.current = -1; die("deadlock: some lanes wait in a cross-lane operation or barrier that the others never reach"); }
        } else {
            idle_rounds = 0;
            last_progress = b.progress;
        }
    }
    const unsigned long n = b.rendezvous;
    for (Fiber &f : b.fibers) std::free(f.stack);
    std::free(b.lds);
    blk() = nullptr;
    return n;
}

// ---------------------------------------------------------------- what the kernels see
struct Idx { unsigned x, y, z; };
struct ThreadIdxProxy { operator Idx() const { return Idx{static_cast<unsigned>(cur_tid()), 0, 0}; } };

inline void wave_fence() { Slot s{}; (void)wave_meet(OP_FENCE, s); }

inline int readlane_i(int v, int l) {
    Slot s{};
    std::memcpy(&s.d[0], &v, sizeof v);
    const Slot *all = wave_meet(OP_READLANE, s);
    int r;
    std::memcpy(&r, &all[l & 63].d[0], sizeof r);
    return r;
}
inline int dpp_source_lane(int l, int ctrl) {
    if (ctrl >= 0 && ctrl <= 0xFF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);        // quad_perm
    if (ctrl == 0x140) return (l & ~15) | (15 - (l & 15));                                   // row_mirror
    if (ctrl == 0x141) return (l & ~7) | (7 - (l & 7));                                      // row_half_mirror
    if (ctrl >= 0x150 && ctrl <= 0x15F) return (l & ~15) | (ctrl - 0x150);                   // row_newbcast
    die("DPP control not modelled");
}
inline int dpp_i(int src, int ctrl) {
    Slot s{};
    std::memcpy(&s.d[0], &src, sizeof src);
    const Slot *all = wave_meet(OP_DPP32, s);
    int r;
    std::memcpy(&r, &all[dpp_source_lane(cur_lane(), ctrl)].d[0], sizeof r);
    return r;
}
inline double dpp_d(double src, int ctrl) {
    Slot s{};
    s.d[0] = src;
    const Slot *all = wave_meet(OP_DPP64, s);
    return all[dpp_source_lane(cur_lane(), ctrl)].d[0];
}
inline unsigned long long ballot(bool p) {
    Slot s{};
    s.d[0] = p ? 1.0 : 0.0;
    const Slot *all = wave_meet(OP_BALLOT, s);
    unsigned long long m = 0;
    for (int l = 0; l < WAVE; ++l) if (all[l].d[0] != 0.0) m |= 1ull << l;
    return m;
}
typedef double v4d_t __attribute__((ext_vector_type(4)));
inline v4d_t mfma_f64_16x16x4(double a, double b, v4d_t c) {
    Slot s{};
    s.d[0] = a;
    s.d[1] = b;
    const Slot *all = wave_meet(OP_MFMA, s);
    const int l = cur_lane(), col = l & 15;
    v4d_t d = c;
    for (int reg = 0; reg < 4; ++reg) {
        const int row = (l >> 4) + 4 * reg;
        double acc = c[reg];
        for (int k = 0; k < 4; ++k) acc = std::fma(all[row + 16 * k].d[0], all[col + 16 * k].d[1], acc);
        d[reg] = acc;
    }
    return d;
}
// v_rcp_f64 / v_rsq_f64 stand-ins good to 2^-14 (the kernels refine them with Newton steps and must not rely on more)
inline double coarse(double x) {
    if (!(x == x) || std::isinf(x) || x == 0.0) return x;
    uint64_t u;
    std::memcpy(&u, &x, 8);
    u &= ~((1ull << 38) - 1ull);
    std::memcpy(&x, &u, 8);
    return x;
}
inline double rcp(double x) { return coarse(1.0 / x); }
inline double rsq(double x) { return coarse(1.0 / std::sqrt(x)); }

template <class T> inline T *lds() { return reinterpret_cast<T *>(blk()->lds); }

}  // namespace sim

// ---- names of the HIP dialect and of the AMDGPU builtins the kernels use
#define threadIdx (static_cast<sim::Idx>(sim::ThreadIdxProxy{}))
#define blockIdx (sim::Idx{sim::blk()->block_idx.x, 0, 0})
#define blockDim (sim::Idx{sim::blk()->block_dim.x, 1, 1})
#define gridDim (sim::Idx{sim::blk()->grid_dim.x, 1, 1})
#define __syncthreads() sim::block_meet(0, nullptr)
inline int __syncthreads_or(int v) { int r = 0; sim::block_meet(v != 0, &r); return r; }
#define __builtin_amdgcn_readlane(v, l) sim::readlane_i((v), (l))
#define __builtin_amdgcn_readfirstlane(v) sim::readlane_i((v), 0)
inline int sim_update_dpp(int, int src, int ctrl, int, int, bool) { return sim::dpp_i(src, ctrl); }
#define __builtin_amdgcn_update_dpp sim_update_dpp
inline double sim_mov_dpp(double src, int ctrl, int, int, bool) { return sim::dpp_d(src, ctrl); }
#define __builtin_amdgcn_mov_dpp sim_mov_dpp
#define __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, x, y, z) sim::mfma_f64_16x16x4((a), (b), (c))
#define __builtin_amdgcn_rcp(x) sim::rcp(x)
#define __builtin_amdgcn_rsq(x) sim::rsq(x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
inline long long sim_clock() { static long long t = 0; return t += 7; }
#define __builtin_amdgcn_s_memtime() sim_clock()
#define __builtin_amdgcn_s_memrealtime() sim_clock()
inline unsigned long long __ballot(bool p) { return sim::ballot(p); }
inline int __any(bool p) { return sim::ballot(p) != 0ull; }
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return static_cast<unsigned long long>((static_cast<unsigned __int128>(a) * b) >> 64); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
#define __builtin_amdgcn_fence(order, scope) sim::wave_fence()
inline int __double2loint(double v) { int64_t u; std::memcpy(&u, &v, 8); return static_cast<int>(u & 0xffffffffll); }
inline int __double2hiint(double v) { int64_t u; std::memcpy(&u, &v, 8); return static_cast<int>(u >> 32); }
inline double __hiloint2double(int hi, int lo) {
    const uint64_t u = (static_cast<uint64_t>(static_cast<uint32_t>(hi)) << 32) | static_cast<uint32_t>(lo);
    double v;
    std::memcpy(&v, &u, 8);
    return v;
}
inline double __longlong_as_double(long long u) { double v; std::memcpy(&v, &u, 8); return v; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
// (__hip_atomic_load is a clang builtin on the host as well; its scope constants come with the HIP language mode)
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
using std::fabs;
using std::fma;
using std::fmax;
using std::fmin;
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
typedef int hipError_t;
typedef void *hipStream_t;
constexpr hipError_t hipSuccess = 0, hipErrorInvalidValue = 1;
