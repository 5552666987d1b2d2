// Wall-clock rate of back-to-back v_mfma_f32_4x4x1 (4 independent chains, one wave per SIMD) as a function of how
// many CUs run it and for how long: separates the architectural issue rate (8 cycles) from the clock the chip
// sustains under matrix-pipe load.  Also a packed-FMA VALU loop for comparison.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, float seed, int iters) {
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1.f;
    f32x4 a[4]; for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
    f32x2 p[8]; for (int i = 0; i < 8; ++i) p[i] = f32x2{x0 + i, x1};
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 23; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) a[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a[c], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 23; ++r)
#pragma unroll
                for (int c = 0; c < 8; ++c) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[c]) : "v"(p[(c + 1) & 7]), "v"(p[(c + 2) & 7]));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 4; ++i) s += a[i][0]; for (int i = 0; i < 8; ++i) s += p[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int MODE> void run(const char* name, int grid, int iters) {
    float* out; long long* cyc; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(out, cyc, 0.5f, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(out, cyc, 0.5f, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double n_ops = (MODE == 0 ? 92.0 : 184.0) * iters;
    printf("%-10s grid=%4d iters=%6d  event %9.1f us  %6.2f ns/op  counter %6.2f cyc/op  wall_clock64 %8.1f us -> %5.0f MHz\n",
           name, grid, iters, ms * 1e3, ms * 1e6 / n_ops, (double)h[0] / n_ops, h[1] / 100.0, (double)h[0] / (h[1] / 100.0));
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int iters : {30, 300, 30000})
        for (int grid : {16, 64, 128, 256, 512}) run<0>("mfma4x4x1", grid, iters);
    for (int iters : {30, 300, 30000})
        for (int grid : {16, 256}) run<1>("pk_fma", grid, iters);
    return 0;
}
