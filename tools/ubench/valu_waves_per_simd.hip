// f32 VALU throughput vs waves per SIMD: one workgroup of 4 * W waves per CU (W waves per SIMD), every wave runs the
// same v_fmac / v_pk_fma / v_mfma stream; time = the slowest wave of workgroup 0.  If a lone wave already saturates the
// pipe, time grows linearly with W.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ void k(float* out, long long* t, float seed, int iters) {
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = x0 + i;
    f32x2 p[8]; for (int i = 0; i < 8; ++i) p[i] = f32x2{x0 + i, x1};
    f32x4 a[4]; for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 160; ++c) {
            if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[c & 7]) : "v"(p[(c + 1) & 7]), "v"(p[(c + 2) & 7]));
            if (MODE == 2 && c < 24) a[c & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a[c & 3], 0, 0, 0);
            if (MODE == 3) asm volatile("v_exp_f32 %0, %1" : "=v"(v[c & 7]) : "v"(x0));
            if (MODE == 4) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(v[c & 7]) : "v"(x0), "v"(x1));
        }
    }
    long long w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i] + p[i][0]; for (int i = 0; i < 4; ++i) s += a[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // oldest-first issue lets the first wave of a SIMD run at full speed: the LAST wave to finish gives the throughput
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)t, (unsigned long long)(w1 - w0));
}
template <int MODE> void run(const char* name, int per_iter) {
    float* out; long long* t; (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&t, 8);
    const int iters = 200;
    double base = 0;
    for (int w : {1, 2, 4}) {
        for (int r = 0; r < 2; ++r) { (void)hipMemset(t, 0, 8); k<MODE><<<256, 256 * w>>>(out, t, 0.5f, iters); (void)hipDeviceSynchronize(); }
        long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        const double ns = h * 10.0 / (iters * (double)per_iter);
        if (w == 1) base = ns;
        printf("%-14s %d wave(s)/SIMD: %6.2f ns per instruction per wave (x%.2f)  -> SIMD issues one every %5.2f cycles @2.3GHz\n",
               name, w, ns, ns / base, ns * 2.3 / w);
    }
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    run<0>("v_fmac_f32", 160); run<1>("v_pk_fma_f32", 160); run<2>("mfma16x16x4", 24); run<3>("v_exp_f32", 160); run<4>("v_mul_lo_u32", 160);
    return 0;
}
