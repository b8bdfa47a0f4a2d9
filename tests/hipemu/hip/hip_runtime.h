// TEST INFRASTRUCTURE ONLY — never part of the product build.
//
// A tiny CPU emulation of the HIP subset used by aruco_slam_amd/csrc/*.hip, selected purely by
// include path (g++ -Itests/hipemu ...) so that the product sources carry no #ifdef.  It lets the
// kernel logic (indexing, control flow, LDS staging, atomics, wave ops) be exercised in this
// GPU-less container before the same sources are run on a real MI355X through gpurun.  It is NOT
// a fallback: the shipped library is always the hipcc/gfx950 build and fails loudly without a GPU.
//
// Model: blocks run one after another; the threads of a block are real OS threads joined by a
// barrier (__syncthreads); a wave is 64 consecutive threads sharing a second barrier for
// __shfl/__ballot (which must be called by every lane of the wave, as in the product code).
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
#include <pthread.h>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

namespace hipemu {
struct WaveCtx {
    pthread_barrier_t bar;
    unsigned long long slot[64];
    int lanes;
};
struct BlockCtx {
    pthread_barrier_t bar;
    std::vector<WaveCtx> waves;
};
inline thread_local uint3_emu t_threadIdx, t_blockIdx;
inline thread_local int t_lane, t_wave;
inline thread_local BlockCtx* t_block;
inline dim3 g_blockDim, g_gridDim;
alignas(16) inline unsigned char g_dynshared[160 * 1024];
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

inline void __syncthreads() { pthread_barrier_wait(&hipemu::t_block->bar); }
inline void __threadfence() { __sync_synchronize(); }
inline void __threadfence_block() { __sync_synchronize(); }

namespace hipemu {
template <class T> inline unsigned long long to_bits(T v) { unsigned long long b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(unsigned long long b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }
inline WaveCtx& wave() { return t_block->waves[t_wave]; }
template <class T> inline T shfl_any(T v, int src) {
    WaveCtx& w = wave();
    w.slot[t_lane] = to_bits(v);
    pthread_barrier_wait(&w.bar);
    T r = from_bits<T>(w.slot[src & 63]);
    pthread_barrier_wait(&w.bar);
    return r;
}
}
template <class T> inline T __shfl(T v, int src, int = 64) { return hipemu::shfl_any(v, src); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) { int s = hipemu::t_lane + (int)d; return hipemu::shfl_any(v, s < 64 ? s : hipemu::t_lane); }
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) { int s = hipemu::t_lane - (int)d; return hipemu::shfl_any(v, s >= 0 ? s : hipemu::t_lane); }
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return hipemu::shfl_any(v, hipemu::t_lane ^ m); }
inline unsigned long long __ballot(int pred) {
    hipemu::WaveCtx& w = hipemu::wave();
    w.slot[hipemu::t_lane] = pred ? 1ull : 0ull;
    pthread_barrier_wait(&w.bar);
    unsigned long long m = 0;
    for (int i = 0; i < w.lanes; i++) m |= (w.slot[i] & 1ull) << i;
    pthread_barrier_wait(&w.bar);
    return m;
}
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }

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
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::malloc(n ? n : 1); return hipSuccess; }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t y = 0; y < h; y++) std::memmove((char*)d + y * dp, (const char*)s + y * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEventRec(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }

#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::g_dynshared);

namespace hipemu {
// kernels without barriers / wave ops listed here run their threads sequentially (fast path for per-pixel kernels)
inline bool is_sequential(const char* name) {
    static const char* seq[] = {"k_render", nullptr};
    for (int i = 0; seq[i]; i++) if (std::strcmp(seq[i], name) == 0) return true;
    return false;
}
template <class F> void run_grid(const char* name, dim3 grid, dim3 block, F&& body) {
    g_blockDim = block;
    g_gridDim = grid;
    const int T = (int)(block.x * block.y * block.z);
    const int nw = (T + 63) / 64;
    const bool seq = is_sequential(name);
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                BlockCtx ctx;
                pthread_barrier_init(&ctx.bar, nullptr, T);
                ctx.waves.resize(nw);
                for (int w = 0; w < nw; w++) {
                    ctx.waves[w].lanes = std::min(64, T - 64 * w);
                    pthread_barrier_init(&ctx.waves[w].bar, nullptr, ctx.waves[w].lanes);
                }
                auto worker = [&](int tid) {
                    t_block = &ctx;
                    t_blockIdx = uint3_emu{bx, by, bz};
                    t_threadIdx = uint3_emu{(unsigned)(tid % block.x), (unsigned)((tid / block.x) % block.y), (unsigned)(tid / (block.x * block.y))};
                    t_lane = tid & 63;
                    t_wave = tid >> 6;
                    body();
                };
                if (T == 1 || seq) {
                    for (int t = 0; t < T; t++) worker(t);
                } else {
                    std::vector<std::thread> th;
                    th.reserve(T);
                    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
                    for (auto& x : th) x.join();
                }
                pthread_barrier_destroy(&ctx.bar);
                for (int w = 0; w < nw; w++) pthread_barrier_destroy(&ctx.waves[w].bar);
            }
}
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::run_grid(#kernel, dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })
