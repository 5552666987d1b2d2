// v_mfma_f32_16x16x32_bf16 on one wave per SIMD: cycles per MFMA (six independent accumulators, back to back) alone, with
// ONE / TWO independent VALU instructions behind every MFMA (v_cvt_pk_bf16_f32, v_sub_f32, v_lshlrev_b32, v_pk_add_f32), and
// with the same VALU count as one burst behind 36 MFMAs.  Answers: does a lone wave's VALU hide in its own bf16 MFMAs' shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define ITERS 2000
template <int MODE, int WAVES> __global__ __launch_bounds__(64 * WAVES) void k(float* out, long long* cyc) {
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a = {threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, threadIdx.x * 3u, 1u, 2u};
    float x0 = threadIdx.x * 1e-3f, x1 = 1.5f, x2 = 2.5f, x3 = 3.5f;
    unsigned u0 = threadIdx.x, u1 = 77u;
    // MODE 8 / 9: operands with random bit patterns, six different A and three different B rotating -- the data toggling of a
    // real GEMM (the chip clocks to its power budget: constant operands flatter the rate)
    u32x4 ar[6], br[3];
    for (int i = 0; i < 6; ++i) {
        unsigned h = (threadIdx.x * 2654435761u) ^ (i * 0x9E3779B9u) ^ (blockIdx.x * 40503u);
        for (int k = 0; k < 4; ++k) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; ar[i][k] = (h & 0x7FFF7FFFu) % 0x3F803F80u; }
    }
    for (int i = 0; i < 3; ++i) {
        unsigned h = (threadIdx.x * 40503u) ^ (i * 0x85EBCA6Bu) ^ 12345u;
        for (int k = 0; k < 4; ++k) { h ^= h << 13; h ^= h >> 17; h ^= h << 5; br[i][k] = (h & 0x7FFF7FFFu) % 0x3F803F80u; }
    }
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int q = 0; q < 36; ++q) {
            if (MODE >= 8) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[q % 6]) : "v"(ar[q % 6]), "v"(br[(q / 6) % 3]));
                if (MODE == 9) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));
                continue;
            }
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[q % 6]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(x0), "v"(x1));
            if (MODE == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));
            if (MODE == 3) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u0) : "v"(u1));
            if (MODE == 4) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*(double*)&x2) : "v"(*(double*)&x0), "v"(*(double*)&x0));
            if (MODE == 5) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(x1)); asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u0) : "v"(u1)); }
        }
        if (MODE == 6) {
#pragma unroll
            for (int q = 0; q < 36; ++q) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x0) : "v"(x1));
        }
        if (MODE == 7) {
#pragma unroll
            for (int q = 0; q < 36; ++q) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(q & 1 ? x0 : x2) : "v"(x1));
        }
    }
    const long long t1 = clock64();
    float s = x0 + x1 + x2 + x3 + (float)u0;
    for (int i = 0; i < 6; ++i) s += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int WAVES> void run(const char* name) {
    float* out; long long* cyc; (void)hipMalloc(&out, 1 << 24); (void)hipHostMalloc(&cyc, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE, WAVES><<<256, 64 * WAVES>>>(out, cyc); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k<MODE, WAVES><<<256, 64 * WAVES>>>(out, cyc); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("%-64s %d wave(s)/SIMD: %7.3f ms, %6.2f shader cycles (s_memtime) and %6.2f ns per MFMA of one wave\n", name, WAVES / 4, ms,
           (double)*cyc / (ITERS * 36.0), ms * 1e6 / (ITERS * 36.0));
}
int main() {
    run<0, 4>("MFMA only");
    run<1, 4>("+ 1 v_cvt_pk_bf16_f32 per MFMA");
    run<2, 4>("+ 1 dependent v_sub_f32 per MFMA");
    run<3, 4>("+ 1 v_lshlrev_b32 per MFMA");
    run<4, 4>("+ 1 v_pk_add_f32 per MFMA");
    run<5, 4>("+ 2 VALU per MFMA");
    run<6, 4>("+ 36 dependent v_sub_f32 behind the 36 MFMAs");
    run<7, 4>("+ 36 v_sub_f32 (two chains) behind the 36 MFMAs");
    run<8, 4>("MFMA only, random operands rotating");
    run<9, 4>("... + 1 v_sub_f32 per MFMA");
    run<8, 8>("MFMA only, random operands rotating");
    run<0, 8>("MFMA only");
    run<5, 8>("+ 2 VALU per MFMA");
    run<6, 8>("+ 36 dependent v_sub_f32 behind the 36 MFMAs");
    return 0;
}
