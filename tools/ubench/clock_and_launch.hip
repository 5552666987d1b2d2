#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, int iters, long long* cyc) {
    float x = threadIdx.x * 1e-9f;
    long long t0 = __builtin_readcyclecounter();
    long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
__global__ void empty(float* out) { if (threadIdx.x == 9999) out[0] = 1; }
int main() {
    float* out; long long* cyc; hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 6; ++rep) {
        int iters = 1 << 20;
        hipEventRecord(a); chain<<<256, 256>>>(out, iters, cyc); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("chain: %.3f ms, %lld shader cycles (%.2f cyc/iter), wall ticks %lld -> shader clk %.0f MHz (wallclk 100MHz assumed)\n", ms, h[0], (double)h[0] / iters, h[1], (double)h[0] / ((double)h[1] / 100.0));
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a); for (int i = 0; i < 1000; ++i) empty<<<1, 64>>>(out); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); printf("1000 empty launches: %.3f ms (%.2f us each)\n", ms, ms);
    }
    return 0;
}
