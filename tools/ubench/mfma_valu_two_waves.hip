// Do VALU instructions of one wave overlap MFMAs of ANOTHER wave on the same SIMD?
// Workgroup = 8 waves (2 per SIMD); waves 0..3 run role A, waves 4..7 role B.  Roles: M = v_mfma_f32_16x16x4
// chain(s), V = v_fmac/v_pk_fma loop, L = ds_read loop, idle.  Time = slowest wave (wall_clock64, 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { IDLE = 0, MFMA16 = 1, VALU = 2, MFMA4 = 3, MFMA16_DEP = 4, PKFMA = 5, LDSR = 6 };
__device__ __forceinline__ float role(int r, int iters, float x0, float x1, float* lds) {
    float s = 0;
    if (r == MFMA16) {
        f32x4 a[4]; for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 24; ++c) a[c & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a[c & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += a[i][0];
    } else if (r == MFMA16_DEP) {
        f32x4 a = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 24; ++c) a = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a, 0, 0, 0);
        s += a[0];
    } else if (r == MFMA4) {
        f32x4 a[4]; for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 96; ++c) a[c & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a[c & 3], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += a[i][0];
    } else if (r == VALU) {
        float v[8]; for (int i = 0; i < 8; ++i) v[i] = x0 + i;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 160; ++c) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[c & 7]) : "v"(x0), "v"(x1));
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if (r == PKFMA) {
        f32x2 p[8]; for (int i = 0; i < 8; ++i) p[i] = f32x2{x0 + i, x1};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 160; ++c) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[c & 7]) : "v"(p[(c + 1) & 7]), "v"(p[(c + 2) & 7]));
        for (int i = 0; i < 8; ++i) s += p[i][0];
    } else if (r == LDSR) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int c = 0; c < 32; ++c) s += lds[(threadIdx.x + 64 * c) & 2047];
    }
    return s;
}
__global__ __launch_bounds__(512) void k(float* out, long long* t, float seed, int iters, int ra, int rb) {
    __shared__ float lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1.f;
    long long w0 = wall_clock64();
    float s = role(wave < 4 ? ra : rb, iters, x0, x1, lds);
    long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) t[wave] = w1 - w0;
}
const char* nm[] = {"idle", "mfma16x16x4(4ch)", "v_fmac", "mfma4x4x1(4ch)", "mfma16x16x4(dep)", "v_pk_fma", "ds_read"};
void run(int ra, int rb, int grid) {
    float* out; long long* t; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&t, 64);
    const int iters = 200;
    k<<<grid, 512>>>(out, t, 0.5f, iters, ra, rb); (void)hipDeviceSynchronize();
    k<<<grid, 512>>>(out, t, 0.5f, iters, ra, rb); (void)hipDeviceSynchronize();
    long long h[8]; (void)hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
    printf("A=%-18s B=%-18s grid=%3d  A: %7.1f us  B: %7.1f us\n", nm[ra], nm[rb], grid, h[0] / 100.0, h[4] / 100.0);
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    for (int grid : {1, 256}) {
        run(MFMA16, IDLE, grid); run(MFMA16_DEP, IDLE, grid); run(MFMA4, IDLE, grid); run(VALU, IDLE, grid); run(PKFMA, IDLE, grid); run(LDSR, IDLE, grid);
        run(MFMA16, MFMA16, grid); run(MFMA16_DEP, MFMA16_DEP, grid);
        run(MFMA16, VALU, grid); run(MFMA16_DEP, VALU, grid); run(MFMA4, VALU, grid); run(MFMA16, PKFMA, grid);
        run(VALU, VALU, grid); run(MFMA16, LDSR, grid); run(VALU, LDSR, grid);
    }
    return 0;
}
