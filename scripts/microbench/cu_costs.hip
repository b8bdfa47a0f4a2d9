#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ long long CLK() { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); long long t = __builtin_readcyclecounter(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); return t; }
#define LDSB() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
__global__ __launch_bounds__(512) void k(long long* out, double* sink, int nw) {
    __shared__ double s[8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 8192; e += 512) s[e] = 1.0 + 1e-9 * e;
    __syncthreads();
    long long t[12];
    double x = 1.0 + lane * 1e-9, y = 0.999999;
    if (wave < nw) {
    // (a) 64 dependent f64 fma
    t[0] = CLK();
#pragma unroll
    for (int i = 0; i < 64; i++) x = fma(x, y, 1e-12);
    t[1] = CLK();
    // (b) 64 independent f64 fma (8 chains)
    double a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = x + i;
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = fma(a[i], y, 1e-12);
    t[2] = CLK();
#pragma unroll
    for (int i = 0; i < 8; i++) x += a[i];
    // (c) 16 dependent LDS read chains (pointer chase: index from value)
    int idx = lane;
    t[3] = CLK();
#pragma unroll
    for (int i = 0; i < 16; i++) { double v = s[idx + wave * 64]; idx = (int)(v) + lane - 1 + (i & 1); }
    t[4] = CLK();
    x += idx;
    // (d) 16 independent LDS reads
    double acc = 0; double rv[16];
    t[5] = CLK();
#pragma unroll
    for (int i = 0; i < 16; i++) rv[i] = s[lane + 66 * i + wave * 1000];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[6] = CLK();
#pragma unroll
    for (int i = 0; i < 16; i++) acc += rv[i];
    x += acc;
    // (e) 16 LDS writes b64
#pragma unroll
    for (int i = 0; i < 16; i++) s[lane + 66 * i + wave * 1000] = x + i;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[7] = CLK();
    // (f) 4 mfma dependent? independent 4
    typedef double v4d __attribute__((ext_vector_type(4)));
    v4d g[4];
#pragma unroll
    for (int q = 0; q < 4; q++) g[q] = v4d{x, x, x, x};
    t[8] = CLK();
#pragma unroll
    for (int q = 0; q < 4; q++) g[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, g[q], 0, 0, 0);
    x += g[0][0] + g[1][1] + g[2][2] + g[3][3];
    t[9] = CLK();
    }
    // (g) 16 barriers
    t[10] = CLK();
#pragma unroll
    for (int i = 0; i < 16; i++) LDSB();
    t[11] = CLK();
    if (wave < nw) sink[tid] = x;
    if (lane == 0 && wave < nw) for (int i = 0; i < 12; i++) out[wave * 12 + i] = t[i];
}
int main() {
    long long* d; double* sk; hipMalloc(&d, 8 * 12 * 8); hipMalloc(&sk, 512 * 8);
    long long h[96];
    for (int nw : {1, 4, 8}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, sk, nw); hipDeviceSynchronize(); }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int w = 0; w < nw; w += (nw > 1 ? nw - 1 : 1)) {
            long long* t = h + w * 12;
            printf("nw=%d wave %d: dep fma x64 %lld | indep fma x64 %lld | dep lds x16 %lld | indep lds x16 %lld | lds writes x16 %lld | mfma x4 %lld | barriers x16 %lld\n", nw, w,
                   t[1] - t[0], t[2] - t[1], t[4] - t[3], t[6] - t[5], t[7] - t[6], t[9] - t[8], t[11] - t[10]);
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d, sk, 8); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("100 launches %.3f ms\n", ms);
    return 0;
}
