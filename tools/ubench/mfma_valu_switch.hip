// Does VALU work hide under v_mfma_f32_4x4x1 for a single wave per SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 500
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, float seed) {
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = x0 + i;
    f32x4 a[4]; for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 23; ++r) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a[c], 0, 0, 0);
                if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[c]) : "v"(x0), "v"(x1));            // 1 independent VALU per MFMA
                if (MODE == 2) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[c]) : "v"(x0), "v"(x1)); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[4 + c]) : "v"(x0), "v"(x1)); }
                if (MODE == 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[0]) : "v"(x0), "v"(x1));            // dependent VALU chain
                if (MODE == 4) asm volatile("s_nop 0");
            }
        }
        if (MODE == 5) {  // step boundary: results feed the next step's B operand (pipeline drain)
            x1 = a[0][0] * 1e-30f + x1;
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i]; for (int i = 0; i < 4; ++i) s += a[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + x1;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8);
    k<MODE><<<256, 256>>>(out, cyc, 0.5f); (void)hipDeviceSynchronize();
    k<MODE><<<256, 256>>>(out, cyc, 0.5f); (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %8.1f cycles per 92-MFMA step (%.2f per MFMA)\n", name, (double)h / ITERS, (double)h / ITERS / 92);
}
int main() {
    run<0>("92 MFMA, 4 chains");
    run<1>("+1 independent v_fmac per MFMA");
    run<2>("+2 independent v_fmac per MFMA");
    run<3>("+1 dependent-chain v_fmac per MFMA");
    run<4>("+1 s_nop per MFMA");
    run<5>("92 MFMA + step-boundary dependency on result");
    return 0;
}
