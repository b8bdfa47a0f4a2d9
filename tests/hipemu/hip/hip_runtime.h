// TEST INFRASTRUCTURE ONLY — never part of the product build.
//
// A tiny CPU emulation of the HIP subset used by aruco_slam_amd/csrc/*.hip, selected purely by
// include path (g++ -Itests/hipemu ...) so that the product sources carry no #ifdef.  It lets the
// kernel logic (indexing, control flow, LDS staging, atomics, wave ops) be exercised in this
// GPU-less container before the same sources are run on a real MI355X through gpurun.  It is NOT
// a fallback: the shipped library is always the hipcc/gfx950 build and fails loudly without a GPU.
//
// Model: every block runs in one OS thread (several blocks in parallel on the host cores); its threads are
// cooperative coroutines (ucontext) that switch at __syncthreads and at wave collectives (__shfl/__ballot,
// which must be called by every live lane of the wave, as in the product code).  __shared__ variables are
// thread_local statics, i.e. private to the block currently running on that OS thread.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <thread>
#include <vector>
#include <chrono>
#include <algorithm>
#include <ucontext.h>
#include <atomic>
#include <mutex>

#define __global__
#define __device__
#define __host__
#define __shared__ static thread_local
#ifndef __align__
#define __align__(n) __attribute__((aligned(n)))
#endif
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct double2 { double x, y; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

namespace hipemu {
struct Lane {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    uint3_emu tidx;
    int lane, wave;
};
struct WaveCtx {
    unsigned long long slot[64];
    unsigned long long slot2[64];
    int lanes = 0;        // lanes that exist
    int live = 0;         // lanes that have not returned yet
    int arrived = 0;
    unsigned gen = 0;
};
struct BlockCtx {
    std::vector<Lane> lanes;
    std::vector<WaveCtx> waves;
    ucontext_t sched;
    int cur = 0;
    int live = 0;         // threads that have not returned
    int arrived = 0;
    unsigned gen = 0;
    uint3_emu bidx;
};
inline thread_local uint3_emu t_threadIdx, t_blockIdx;
inline thread_local int t_lane, t_wave;
inline thread_local BlockCtx* t_block;
inline dim3 g_blockDim, g_gridDim;
inline void yield_lane() {                     // back to the block scheduler, which resumes the next live lane
    BlockCtx* b = t_block;
    swapcontext(&b->lanes[b->cur].ctx, &b->sched);
}
}

#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
#define warpSize 64
using std::min;
using std::max;
using std::isfinite;
using std::isnan;

inline void __syncthreads() {
    hipemu::BlockCtx* b = hipemu::t_block;
    const unsigned g = b->gen;
    if (++b->arrived >= b->live) { b->arrived = 0; b->gen++; return; }
    while (b->gen == g) hipemu::yield_lane();
}
inline void __threadfence() { __sync_synchronize(); }
inline void __threadfence_block() { __sync_synchronize(); }
#define __HIP_MEMORY_SCOPE_AGENT 4
template <class T, class U> inline void __hip_atomic_store(T* p, U v, int, int) { __atomic_store_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T> inline T __hip_atomic_load(const T* p, int, int) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
inline unsigned long long wall_clock64() { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10; }
inline unsigned long long clock64() { return wall_clock64() * 24; }

namespace hipemu {
template <class T> inline unsigned long long to_bits(T v) { unsigned long long b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(unsigned long long b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
inline WaveCtx& wave() { return t_block->waves[t_wave]; }
inline void wave_sync() {                      // all live lanes of the wave
    WaveCtx& w = wave();
    const unsigned g = w.gen;
    if (++w.arrived >= w.live) { w.arrived = 0; w.gen++; return; }
    while (w.gen == g) yield_lane();
}
template <class T> inline T shfl_any(T v, int src) {
    WaveCtx& w = wave();
    const int me = t_lane;
    w.slot[me] = to_bits(v);
    wave_sync();
    T r = from_bits<T>(w.slot[src & 63]);
    wave_sync();
    return r;
}
}
template <class T> inline T __shfl(T v, int src, int = 64) { return hipemu::shfl_any(v, src); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) { int s = hipemu::t_lane + (int)d; return hipemu::shfl_any(v, s < 64 ? s : hipemu::t_lane); }
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) { int s = hipemu::t_lane - (int)d; return hipemu::shfl_any(v, s >= 0 ? s : hipemu::t_lane); }
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return hipemu::shfl_any(v, hipemu::t_lane ^ m); }
inline unsigned long long __ballot(int pred) {
    hipemu::WaveCtx& w = hipemu::wave();
    const int me = hipemu::t_lane;
    w.slot[me] = pred ? 1ull : 0ull;
    hipemu::wave_sync();
    unsigned long long m = 0;
    for (int i = 0; i < w.lanes; i++) m |= (w.slot[i] & 1ull) << i;
    hipemu::wave_sync();
    return m;
}
// v_mfma_f64_16x16x4_f64: A[i][k] from lane k*16+i, B[k][j] from lane k*16+j, D rows (lane>>4)+4*reg, column lane&15
typedef double hipemu_v4d __attribute__((vector_size(4 * sizeof(double))));
inline hipemu_v4d __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hipemu_v4d c, int, int, int) {
    hipemu::WaveCtx& w = hipemu::wave();
    const int me = hipemu::t_lane;
    w.slot[me] = hipemu::to_bits(a);
    w.slot2[me] = hipemu::to_bits(b);
    hipemu::wave_sync();
    hipemu_v4d d = c;
    const int col = me & 15;
    for (int reg = 0; reg < 4; reg++) {
        const int row = (me >> 4) + 4 * reg;
        double s = c[reg];
        for (int k = 0; k < 4; k++) s += hipemu::from_bits<double>(w.slot[k * 16 + row]) * hipemu::from_bits<double>(w.slot2[k * 16 + col]);
        d[reg] = s;
    }
    hipemu::wave_sync();
    return d;
}
// lanes of a wave run in lockstep on the device; here they are coroutines, so code that relies on lockstep says so
inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_sync(); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }

template <class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline double atomicAdd(double* p, double v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    unsigned long long old = __atomic_load_n(q, __ATOMIC_SEQ_CST), nw;
    do { nw = hipemu::to_bits(hipemu::from_bits<double>(old) + v); } while (!__atomic_compare_exchange_n(q, &old, nw, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST));
    return hipemu::from_bits<double>(old);
}
template <class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicAnd(T* p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> inline T atomicCAS(T* p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
template <class T> inline T atomicMin(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}
template <class T> inline T atomicMax(T* p, T v) {
    T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return old;
}

// ---- host API ---------------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct hipEventRec { std::chrono::steady_clock::time_point t; };
typedef hipEventRec* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; char gcnArchName[256]; };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::strcpy(p->name, "hipemu-cpu"); std::strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 8; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); if (*p) std::memset(*p, 0xCD, n); return *p ? hipSuccess : hipErrorInvalidValue; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
constexpr unsigned hipHostMallocDefault = 0;
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::malloc(n ? n : 1); return hipSuccess; }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind) {
    for (size_t y = 0; y < h; y++) std::memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t y = 0; y < h; y++) std::memmove((char*)d + y * dp, (const char*)s + y * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(1); return hipSuccess; }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, unsigned, const unsigned*) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEventRec(); return hipSuccess; }
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipEventRec(); return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }


namespace hipemu {
constexpr size_t kStackBytes = 96 * 1024;
template <class F> struct Tramp {
    static void entry(unsigned lo, unsigned hi) {
        F* f = reinterpret_cast<F*>(((unsigned long long)hi << 32) | lo);
        (*f)();
        BlockCtx* b = t_block;
        Lane& me = b->lanes[b->cur];
        me.done = true;
        b->live--;
        WaveCtx& w = b->waves[me.wave];
        w.live--;
        // a lane that returns may complete a pending barrier / collective of the others
        if (b->live > 0 && b->arrived >= b->live) { b->arrived = 0; b->gen++; }
        if (w.live > 0 && w.arrived >= w.live) { w.arrived = 0; w.gen++; }
        swapcontext(&me.ctx, &b->sched);
    }
};
template <class F> void run_block(F& body, dim3 block, unsigned bx, unsigned by, unsigned bz, std::vector<char>& stacks) {
    const int T = (int)(block.x * block.y * block.z);
    const int nw = (T + 63) / 64;
    BlockCtx ctx;
    ctx.lanes.resize(T);
    ctx.waves.resize(nw);
    ctx.live = T;
    ctx.bidx = uint3_emu{bx, by, bz};
    if (stacks.size() < (size_t)T * kStackBytes) stacks.resize((size_t)T * kStackBytes);
    for (int w = 0; w < nw; w++) { ctx.waves[w].lanes = std::min(64, T - 64 * w); ctx.waves[w].live = ctx.waves[w].lanes; }
    t_block = &ctx;
    t_blockIdx = ctx.bidx;
    const unsigned long long fp = reinterpret_cast<unsigned long long>(&body);
    for (int t = 0; t < T; t++) {
        Lane& l = ctx.lanes[t];
        l.tidx = uint3_emu{(unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / (block.x * block.y))};
        l.lane = t & 63;
        l.wave = t >> 6;
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = stacks.data() + (size_t)t * kStackBytes;
        l.ctx.uc_stack.ss_size = kStackBytes;
        l.ctx.uc_link = nullptr;
        makecontext(&l.ctx, reinterpret_cast<void (*)()>(&Tramp<F>::entry), 2, (unsigned)(fp & 0xffffffffu), (unsigned)(fp >> 32));
    }
    while (ctx.live > 0) {
        for (int t = 0; t < T; t++) {
            Lane& l = ctx.lanes[t];
            if (l.done) continue;
            ctx.cur = t;
            t_threadIdx = l.tidx;
            t_lane = l.lane;
            t_wave = l.wave;
            swapcontext(&ctx.sched, &l.ctx);
        }
    }
    t_block = nullptr;
}
template <class F> void run_grid(const char*, dim3 grid, dim3 block, F&& body) {
    g_blockDim = block;
    g_gridDim = grid;
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    std::atomic<unsigned long long> next{0};
    // HIPEMU_ORDER = reverse | shuffle: the order in which the workgroups of a launch are started (the device promises none): a
    // kernel whose workgroups hand data to each other through global memory must give the same result in every order
    static const int order_mode = [] { const char* e = std::getenv("HIPEMU_ORDER"); return !e ? 0 : e[0] == 'r' ? 1 : e[0] == 's' ? 2 : 0; }();
    std::vector<unsigned long long> perm;
    if (order_mode == 2) {
        perm.resize(nblocks);
        for (unsigned long long k = 0; k < nblocks; k++) perm[k] = k;
        unsigned long long st = 0x9E3779B97F4A7C15ull ^ nblocks;
        for (unsigned long long k = nblocks; k > 1; k--) {          // Fisher-Yates with a fixed xorshift: reproducible
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            std::swap(perm[k - 1], perm[st % k]);
        }
    }
    auto worker = [&]() {
        std::vector<char> stacks;
        for (;;) {
            unsigned long long i = next.fetch_add(1);
            if (i >= nblocks) break;
            if (order_mode == 1) i = nblocks - 1 - i;
            else if (order_mode == 2) i = perm[i];
            unsigned bx = (unsigned)(i % grid.x), by = (unsigned)((i / grid.x) % grid.y), bz = (unsigned)(i / ((unsigned long long)grid.x * grid.y));
            F local = body;                      // per-OS-thread copy of the launch closure
            run_block(local, block, bx, by, bz, stacks);
        }
    };
    unsigned nthreads = std::min<unsigned long long>(nblocks, std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
    if (nthreads <= 1) { worker(); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; t++) th.emplace_back(worker);
    for (auto& x : th) x.join();
}
}

#define ASLAM_LDS_BARRIER() __syncthreads()
#define ASLAM_DYN_LDS(name) alignas(16) static thread_local unsigned char name[160 * 1024]
#define ASLAM_RCP_ESTIMATE(x) (1.0 / (x))
#define ASLAM_WAVE_BCAST(v, src) __shfl(v, src)
#define ASLAM_SPIN_UNTIL(cond) do { while (!(cond)) hipemu::yield_lane(); } while (0)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::run_grid(#kernel, dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
