// hip_emu.h — minimal CPU emulation of the HIP device model, used ONLY by tests/.
//
// The product kernels (enerf_amd/csrc/*.hip) are compiled a second time with
//   g++ -x c++ -DENERF_EMU -include tests/emu/hip_emu.h
// into tests/emu/_build/libenerf_emu.so, so the exact kernel source (indexing, MFMA lane layouts,
// LDS staging, cross-lane reductions) can be checked against the oracle on a machine without a GPU.
// It is test infrastructure: the product loader (enerf_amd/lib.py) never loads this library.
//
// Model: a block is executed by ONE OS thread; its lanes are ucontext coroutines that yield at every
// cross-lane operation (wave shuffles, MFMA, __syncthreads).  Blocks of a grid are distributed over a
// small pool of OS threads.  __shared__ is `static thread_local`, so each worker owns one copy.
// Kernels launched with ENERF_LAUNCH_SIMPLE promise to contain no cross-lane operation and are run
// as plain loops (fast path); a cross-lane operation there aborts with a message.
#pragma once
#include <ucontext.h>

#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__ __restrict

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct float3 { float x, y, z; };
inline float2 make_float2(float x, float y) { return {x, y}; }
inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
typedef float emu_f32x4 __attribute__((vector_size(16)));
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }

namespace emu {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

struct Lane {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    unsigned tid = 0;
    uint64_t wave_ops = 0;   // cross-lane ops issued by this lane within its wave
    uint64_t block_ops = 0;  // __syncthreads issued by this lane
};

struct WaveState {
    uint64_t arrived = 0;            // total arrivals at wave-level ops
    float buf[2][3][kWave];          // double-buffered exchange slots (up to 3 operands)
};

struct BlockCtx {
    dim3 blockIdx, blockDim, gridDim;
    unsigned tid = 0;                // current lane (thread id x)
    bool lockstep = false;
    std::vector<Lane> lanes;
    std::vector<WaveState> waves;
    uint64_t block_arrived = 0;
    ucontext_t main_ctx;
    Lane* cur = nullptr;
    std::function<void()> body;
    char* dyn_smem = nullptr;
};

inline BlockCtx*& ctx() {
    static thread_local BlockCtx* c = nullptr;
    return c;
}

inline void die(const char* msg) {
    fprintf(stderr, "[hip_emu] %s\n", msg);
    abort();
}

inline void yield() {
    BlockCtx* c = ctx();
    swapcontext(&c->cur->ctx, &c->main_ctx);
}

// Exchange up to 3 floats with the other lanes of the wave; returns pointer to the slot arrays.
inline float (*wave_exchange(float a, float b = 0.f, float c3 = 0.f))[kWave] {
    BlockCtx* c = ctx();
    if (!c->lockstep) die("cross-lane op inside a kernel launched with ENERF_LAUNCH_SIMPLE");
    Lane* l = c->cur;
    unsigned lane = l->tid % kWave;
    WaveState& w = c->waves[l->tid / kWave];
    unsigned wave_lanes = std::min<unsigned>(kWave, c->blockDim.x - (l->tid / kWave) * kWave);
    int par = (int)(l->wave_ops & 1);
    w.buf[par][0][lane] = a;
    w.buf[par][1][lane] = b;
    w.buf[par][2][lane] = c3;
    w.arrived++;
    l->wave_ops++;
    uint64_t need = l->wave_ops * wave_lanes;
    while (w.arrived < need) yield();
    return w.buf[par];
}

inline void syncthreads() {
    BlockCtx* c = ctx();
    if (!c->lockstep) die("__syncthreads inside a kernel launched with ENERF_LAUNCH_SIMPLE");
    Lane* l = c->cur;
    c->block_arrived++;
    l->block_ops++;
    uint64_t need = l->block_ops * (uint64_t)c->blockDim.x;
    while (c->block_arrived < need) yield();
}

inline void lane_entry() {
    BlockCtx* c = ctx();
    c->body();
    c->cur->done = true;
    swapcontext(&c->cur->ctx, &c->main_ctx);
}

inline void run_block_lockstep(BlockCtx& c) {
    unsigned n = c.blockDim.x;
    if (c.lanes.size() != n) {
        for (auto& l : c.lanes) free(l.stack);
        c.lanes.assign(n, Lane());
        for (auto& l : c.lanes) l.stack = (char*)malloc(kStack);
    }
    c.waves.assign((n + kWave - 1) / kWave, WaveState());
    c.block_arrived = 0;
    for (unsigned t = 0; t < n; ++t) {
        Lane& l = c.lanes[t];
        l.done = false; l.tid = t; l.wave_ops = 0; l.block_ops = 0;
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack;
        l.ctx.uc_stack.ss_size = kStack;
        l.ctx.uc_link = &c.main_ctx;
        makecontext(&l.ctx, (void (*)())lane_entry, 0);
    }
    unsigned remaining = n;
    uint64_t rounds = 0;
    while (remaining) {
        remaining = 0;
        for (unsigned t = 0; t < n; ++t) {
            Lane& l = c.lanes[t];
            if (l.done) continue;
            c.cur = &l;
            c.tid = t;
            swapcontext(&c.main_ctx, &l.ctx);
            if (!l.done) remaining++;
        }
        if (++rounds > (1ull << 32)) die("lockstep deadlock");
    }
}

template <class F>
inline void launch(bool lockstep, dim3 grid, dim3 block, size_t shmem, F&& kernel_call) {
    if (block.y != 1 || block.z != 1) die("emu supports 1-D blocks only");
    unsigned nblk = grid.x * grid.y * grid.z;                     // x fastest, like the hardware's dispatch order
    unsigned nthreads = std::min<unsigned>(nblk, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* e = getenv("ENERF_EMU_THREADS")) nthreads = std::max(1, atoi(e));
    nthreads = std::min(nthreads, nblk);
    std::atomic<unsigned> next{0};
    auto worker = [&]() {
        BlockCtx c;
        c.blockDim = block; c.gridDim = grid; c.lockstep = lockstep;
        std::vector<char> smem(shmem + 64);
        c.dyn_smem = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
        ctx() = &c;
        for (;;) {
            unsigned b = next.fetch_add(1);
            if (b >= nblk) break;
            c.blockIdx = dim3(b % grid.x, (b / grid.x) % grid.y, b / (grid.x * grid.y));
            if (!lockstep) {
                for (unsigned t = 0; t < block.x; ++t) { c.tid = t; kernel_call(); }
            } else {
                c.body = kernel_call;
                run_block_lockstep(c);
            }
        }
        for (auto& l : c.lanes) free(l.stack);
        ctx() = nullptr;
    };
    if (nthreads <= 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
}

struct TidProxy { unsigned y = 0, z = 0; struct X { operator unsigned() const { return ctx()->tid; } } x; };
struct BidProxy {
    struct X { operator unsigned() const { return ctx()->blockIdx.x; } } x;
    struct Y { operator unsigned() const { return ctx()->blockIdx.y; } } y;
    struct Z { operator unsigned() const { return ctx()->blockIdx.z; } } z;
};
struct BdimProxy { unsigned y = 1, z = 1; struct X { operator unsigned() const { return ctx()->blockDim.x; } } x; };
struct GdimProxy {
    struct X { operator unsigned() const { return ctx()->gridDim.x; } } x;
    struct Y { operator unsigned() const { return ctx()->gridDim.y; } } y;
    struct Z { operator unsigned() const { return ctx()->gridDim.z; } } z;
};

}  // namespace emu

static emu::TidProxy threadIdx;
static emu::BidProxy blockIdx;
static emu::BdimProxy blockDim;
static emu::GdimProxy gridDim;

#define ENERF_DYN_SMEM(type, name) type* name = (type*)emu::ctx()->dyn_smem

inline void __syncthreads() { emu::syncthreads(); }

inline float __shfl_xor(float v, int mask, int width = 64) {
    (void)width;
    unsigned lane = emu::ctx()->cur->tid % emu::kWave;
    auto buf = emu::wave_exchange(v);
    return buf[0][lane ^ (unsigned)mask];
}
inline float __shfl(float v, int src, int width = 64) {
    (void)width;
    auto buf = emu::wave_exchange(v);
    return buf[0][(unsigned)src % emu::kWave];
}
inline int __shfl(int v, int src, int width = 64) {
    float f; memcpy(&f, &v, 4);
    f = __shfl(f, src, width);
    memcpy(&v, &f, 4);
    return v;
}
inline double __shfl_xor(double v, int mask, int width = 64) {      // two 32-bit exchanges, as the hardware does it
    int w[2]; memcpy(w, &v, 8);
    float f0, f1; memcpy(&f0, &w[0], 4); memcpy(&f1, &w[1], 4);
    f0 = __shfl_xor(f0, mask, width); f1 = __shfl_xor(f1, mask, width);
    memcpy(&w[0], &f0, 4); memcpy(&w[1], &f1, 4); memcpy(&v, w, 8);
    return v;
}
inline int __shfl_xor(int v, int mask, int width = 64) {
    float f; memcpy(&f, &v, 4);
    f = __shfl_xor(f, mask, width);
    memcpy(&v, &f, 4);
    return v;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=4*(l>>4)+r][col=l&15];
// result is a k-ordered fmaf chain (cdna_hip_programming.md §3).
inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    unsigned lane = emu::ctx()->cur->tid % emu::kWave;
    auto buf = emu::wave_exchange(a, b);
    unsigned col = lane & 15, g = lane >> 4;
    emu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        unsigned row = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(buf[0][row + 16 * k], buf[1][col + 16 * k], acc);
        d[r] = acc;
    }
    return d;
}

// v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 blocks, k = 1.  Lane l = 4b + i supplies A_b[i] and B_b[i]; lane 4b + j
// receives D_b[0..3][j] in its four accumulator registers.
// cbsz / abid (the A-operand broadcast controls): the 16 blocks form groups of 2^cbsz consecutive blocks and every block of a group
// takes the A column of the group's block `abid` (cbsz = 4: ONE block's weights for all 16).
inline emu_f32x4 __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, emu_f32x4 c, int cbsz, int abid, int) {
    unsigned lane = emu::ctx()->cur->tid % emu::kWave;
    auto buf = emu::wave_exchange(a, b);
    unsigned blk = lane & ~3u;
    if (cbsz > 0) blk = ((((lane >> 2) >> cbsz) << cbsz) + (unsigned)abid) * 4u;
    emu_f32x4 d = c;
    for (int i = 0; i < 4; ++i) d[i] = fmaf(buf[0][blk + i], b, c[i]);
    return d;
}

inline void __builtin_amdgcn_sched_barrier(int) {}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }   // only ever applied to wave-uniform values
inline int __builtin_amdgcn_readlane(int v, int lane) { return __shfl(v, lane); }   // uniform lane index, uniform control flow
inline void __builtin_amdgcn_s_sleep(int) {}
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline double atomicAdd(double* p, double v) {          // blocks run on several OS threads: real atomic
    uint64_t* u = reinterpret_cast<uint64_t*>(p);
    uint64_t old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
    double od;
    do { memcpy(&od, &old, 8); double nd = od + v; memcpy(&neu, &nd, 8);
    } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return od;
}

inline float atomicAdd(float* p, float v) {            // blocks run on several OS threads: real atomic
    uint32_t* u = reinterpret_cast<uint32_t*>(p);
    uint32_t old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
    float of;
    do { memcpy(&of, &old, 4); float nf = of + v; memcpy(&neu, &nf, 4);
    } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return of;
}

inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

inline int atomicMin(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old > v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

#define ENERF_LAUNCH(kern, grid, block, shmem, stream, ...) \
    emu::launch(true, dim3(grid), dim3(block), (size_t)(shmem), [&]() { kern(__VA_ARGS__); })
#define ENERF_LAUNCH_SIMPLE(kern, grid, block, shmem, stream, ...) \
    emu::launch(false, dim3(grid), dim3(block), (size_t)(shmem), [&]() { kern(__VA_ARGS__); })
