// Instruction-rate microbenchmarks for gfx950: cycles per wave-instruction at 1 wave/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2000
template <int MODE> __global__ void k(float* out, long long* cyc, float seed) {
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f32x4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    unsigned u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    unsigned long long w0 = u0, w1 = u1, w2 = u2, w3 = u3;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        if (MODE == 0) {  // 8 independent v_fma_f32
            x0 = __builtin_fmaf(x0, 1.0001f, seed); x1 = __builtin_fmaf(x1, 1.0001f, seed); x2 = __builtin_fmaf(x2, 1.0001f, seed); x3 = __builtin_fmaf(x3, 1.0001f, seed);
            x4 = __builtin_fmaf(x4, 1.0001f, seed); x5 = __builtin_fmaf(x5, 1.0001f, seed); x6 = __builtin_fmaf(x6, 1.0001f, seed); x7 = __builtin_fmaf(x7, 1.0001f, seed);
        } else if (MODE == 1) {  // 8 independent mfma 4x4x1
            a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a6, 0, 0, 0); a7 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a7, 0, 0, 0);
        } else if (MODE == 2) {  // 8 independent mfma 16x16x4
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a3, 0, 0, 0);
            a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a4, 0, 0, 0); a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a5, 0, 0, 0);
            a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a6, 0, 0, 0); a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, a7, 0, 0, 0);
        } else if (MODE == 3) {  // 4 independent 32x32->64 multiplies (v_mad_u64_u32)
            w0 = (unsigned long long)(unsigned)w0 * 0xD2511F53u + (w0 >> 32); w1 = (unsigned long long)(unsigned)w1 * 0xD2511F53u + (w1 >> 32);
            w2 = (unsigned long long)(unsigned)w2 * 0xD2511F53u + (w2 >> 32); w3 = (unsigned long long)(unsigned)w3 * 0xD2511F53u + (w3 >> 32);
        } else if (MODE == 4) {  // 4 independent v_mul_lo_u32 + 4 v_mul_hi_u32
            unsigned h0 = __umulhi(u0, 0xD2511F53u), h1 = __umulhi(u1, 0xD2511F53u), h2 = __umulhi(u2, 0xD2511F53u), h3 = __umulhi(u3, 0xD2511F53u);
            u0 = u0 * 0xCD9E8D57u ^ h0; u1 = u1 * 0xCD9E8D57u ^ h1; u2 = u2 * 0xCD9E8D57u ^ h2; u3 = u3 * 0xCD9E8D57u ^ h3;
        } else if (MODE == 5) {  // 4x (log2, sqrt, sin, cos)
            x0 = __builtin_amdgcn_logf(x0 + 2.f); x1 = __builtin_sqrtf(x1 + 2.f); x2 = __builtin_amdgcn_sinf(x2); x3 = __builtin_amdgcn_cosf(x3);
            x4 = __builtin_amdgcn_logf(x4 + 2.f); x5 = __builtin_sqrtf(x5 + 2.f); x6 = __builtin_amdgcn_sinf(x6); x7 = __builtin_amdgcn_cosf(x7);
        } else if (MODE == 6) {  // 8 dependent-chain mfma 4x4x1 on one accumulator
            for (int r = 0; r < 8; ++r) a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a0, 0, 0, 0);
        } else if (MODE == 7) {  // 8 independent xor/add (cheap int)
            u0 = (u0 ^ u1) + 0x9E3779B9u; u1 = (u1 ^ u2) + 0x9E3779B9u; u2 = (u2 ^ u3) + 0x9E3779B9u; u3 = (u3 ^ u0) + 0x9E3779B9u;
            u0 = (u0 ^ u2) + 0xBB67AE85u; u1 = (u1 ^ u3) + 0xBB67AE85u; u2 = (u2 ^ u0) + 0xBB67AE85u; u3 = (u3 ^ u1) + 0xBB67AE85u;
        } else if (MODE == 8) {  // 5 interleaved dependent mfma chains (the rollout pattern)
            for (int r = 0; r < 4; ++r) {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a3, 0, 0, 0);
                a4 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0, x1, a4, 0, 0, 0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[0] + a6[0] + a7[0] + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(w0 ^ w1 ^ w2 ^ w3);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name, int instrs_per_iter, int threads) {
    float* out; long long* cyc; (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 8);
    k<MODE><<<256, threads>>>(out, cyc, 0.5f); (void)hipDeviceSynchronize();
    k<MODE><<<256, threads>>>(out, cyc, 0.5f); (void)hipDeviceSynchronize();
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD=%d : %7.2f cycles per instr-slot (%lld cycles, %d instrs/iter)\n", name, threads / 256, (double)h / ITERS / instrs_per_iter, h, instrs_per_iter);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int th : {256, 512}) {
        run<0>("v_fma_f32 x8 indep", 8, th);
        run<1>("v_mfma_f32_4x4x1_16b x8 indep", 8, th);
        run<6>("v_mfma_f32_4x4x1_16b x8 dependent", 8, th);
        run<8>("v_mfma_f32_4x4x1_16b 5 chains x4", 20, th);
        run<2>("v_mfma_f32_16x16x4 x8 indep", 8, th);
        run<3>("v_mad_u64_u32 x4 indep", 4, th);
        run<4>("v_mul_hi_u32+v_mul_lo_u32 x4 pairs", 8, th);
        run<5>("transcendental x8 (log/sqrt/sin/cos)", 8, th);
        run<7>("xor+add x8 pairs (16 int ops)", 16, th);
    }
    return 0;
}
