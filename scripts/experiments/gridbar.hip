// micro-benchmark: cost of a software grid barrier (agent-scope release/acquire) between phases that rewrite a 603x603 f64 matrix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(ctr, 1u);
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicOr(err, 1u); ok = false; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(256) void k_loop(double* S, int ld, int N, int iters, int nbar, unsigned* ctr, unsigned* err, int do_work) {
    const int nb = gridDim.x;
    const int tiles = (N + 63) / 64;
    unsigned bar = 0;
    for (int it = 0; it < iters; it++) {
        if (do_work && blockIdx.x < tiles * tiles) {
            int r0 = (blockIdx.x % tiles) * 64, c0 = (blockIdx.x / tiles) * 64;
            for (int e = threadIdx.x; e < 4096; e += 256) {
                int r = r0 + (e & 63), c = c0 + (e >> 6);
                if (r < N && c < N) S[(size_t)c * ld + r] = S[(size_t)c * ld + r] * 0.999 + 1e-3;
            }
        }
        for (int b = 0; b < nbar; b++) { bar++; if (!grid_barrier(ctr, bar * nb, err)) return; }
    }
}
int main() {
    const int N = 603, ld = 640;
    double* S; unsigned* ctr;
    hipMalloc(&S, sizeof(double) * ld * ld); hipMemset(S, 0, sizeof(double) * ld * ld);
    hipMalloc(&ctr, 8); 
    for (int nblocks : {101, 26}) for (int work = 0; work < 2; work++) for (int nbar = 1; nbar <= 3; nbar += 2) {
        hipMemset(ctr, 0, 8);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        const int iters = 2000;
        hipLaunchKernelGGL(k_loop, dim3(nblocks), dim3(256), 0, 0, S, ld, N, iters, nbar, ctr, ctr + 1, work);
        hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        unsigned h[2]; hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost);
        printf("blocks %d work %d barriers/iter %d: %.2f us/iter  (err %u)\n", nblocks, work, nbar, us / iters, h[1]);
    }
    return 0;
}
