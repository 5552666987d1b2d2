// How much VALU work fits in the shadow of a v_mfma_f32_16x16x4_f32 (8 passes, 32 cycles)?
//  (a) same wave: NV independent v_fmac after every MFMA;  (b) other wave on the same SIMD doing v_fmac while this
//  wave pads every MFMA with s_nop so the arbiter can pick the other wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int NOPS, int DEP>
__device__ __forceinline__ float mfma_role(int iters, float x0, float x1) {
    f32x4 a[4]; for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = x0 + i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int c = 0; c < 24; ++c) {
            a[DEP ? 0 : (c & 3)] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a[DEP ? 0 : (c & 3)], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NV; ++n) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[n & 7]) : "v"(x0), "v"(x1));
#pragma unroll
            for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 7");
        }
    float s = 0; for (int i = 0; i < 4; ++i) s += a[i][0]; for (int i = 0; i < 8; ++i) s += v[i];
    return s;
}
__device__ __forceinline__ float valu_role(int iters, float x0, float x1, int per) {
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = x0 + i;
    for (int it = 0; it < iters; ++it)
        for (int r = 0; r < per; ++r)
#pragma unroll
            for (int c = 0; c < 24; ++c) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[c & 7]) : "v"(x0), "v"(x1));
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
    return s;
}
template <int NV, int NOPS, int DEP>
__global__ __launch_bounds__(512) void k(float* out, long long* t, float seed, int iters, int other_valu) {
    const int wave = threadIdx.x >> 6;
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1.f;
    long long w0 = wall_clock64();
    float s = 0;
    if (wave < 4) s = mfma_role<NV, NOPS, DEP>(iters, x0, x1);
    else if (other_valu) s = valu_role(iters, x0, x1, other_valu);
    long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) t[wave] = w1 - w0;
}
template <int NV, int NOPS, int DEP> void run(int other) {
    float* out; long long* t; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&t, 64);
    const int iters = 200;
    for (int r = 0; r < 2; ++r) { k<NV, NOPS, DEP><<<256, 512>>>(out, t, 0.5f, iters, other); (void)hipDeviceSynchronize(); }
    long long h[8]; (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    const double n = 200.0 * 24;
    printf("%s MFMA + %d v_fmac + %d s_nop7 each | other wave: %d v_fmac per MFMA   mfma wave %6.2f ns/MFMA (%5.1f cyc @2.3GHz)   other wave done at %6.2f ns/MFMA\n",
           DEP ? "dep  " : "indep", NV, NOPS, other, h[0] * 10.0 / n, h[0] * 10.0 / n * 2.3, h[4] * 10.0 / n);
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    run<0, 0, 0>(0); run<1, 0, 0>(0); run<2, 0, 0>(0); run<4, 0, 0>(0); run<6, 0, 0>(0); run<8, 0, 0>(0);
    run<0, 0, 1>(0); run<2, 0, 1>(0); run<4, 0, 1>(0); run<6, 0, 1>(0);
    run<0, 0, 0>(4); run<0, 1, 0>(4); run<0, 2, 0>(4); run<0, 3, 0>(4); run<0, 3, 0>(6); run<0, 3, 1>(6); run<0, 4, 0>(6);
    return 0;
}
