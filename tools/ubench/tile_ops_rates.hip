// Issue rates of the instructions Tile16H's step is made of (EXPERIMENTS R5.1), per SIMD, 1 / 2 / 4 waves per SIMD: one
// workgroup of 4 * W waves per CU, every wave runs the same stream of ONE instruction over 8 independent registers; time =
// the slowest wave of workgroup 0.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tile_ops_rates.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE> __global__ void k(float* out, long long* t, float seed, int iters) {
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = x0 + i;
    unsigned u[8]; for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    f32x4 a[4]; for (int i = 0; i < 4; ++i) a[i] = f32x4{0, 0, 0, 0};
    f16x8 ha, hb; for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(x0 + i); hb[i] = (_Float16)(x1 - i); }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[4]; for (int i = 0; i < 4; ++i) pk[i] = f32x2{x0 + i, x1};
    const unsigned long long mask = 0x5555555555555555ull ^ (unsigned long long)iters;
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 160; ++c) {
            if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(v[c & 7]) : "v"(x0), "v"(x1), "v"(u[0]));
            if (MODE == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(u[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 4) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[c & 7]), "+v"(v[(c + 4) & 7]));
            if (MODE == 6 && c < 48) a[c & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, a[c & 3], 0, 0, 0);   // four independent accumulators
            if (MODE == 7 && c < 48) a[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, a[0], 0, 0, 0);           // one dependent chain
            if (MODE == 8) asm volatile("v_cmp_gt_f32 vcc, |%0|, %1" : : "v"(v[c & 7]), "v"(x1) : "vcc");
            if (MODE == 9) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 10) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(v[c & 7]) : "v"(x0));
            if (MODE == 12) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v[c & 7]) : "v"(x0), "v"(x1), "s"(mask));
            if (MODE == 13) asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(v[c & 7]) : "v"(x1), "s"(mask));
            if (MODE == 14) asm volatile("v_and_b32 %0, %1, %2" : "=v"(u[c & 7]) : "v"(u[(c + 1) & 7]), "v"(u[(c + 2) & 7]));
            if (MODE == 15) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 16) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(v[c & 7]) : "v"(x0), "v"(x1), "v"(v[(c + 1) & 7]));
            if (MODE == 17) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(u[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 18) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(u[c & 7]) : "v"(x0));
            if (MODE == 19) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[c & 7]) : "v"(u[(c + 1) & 7]), "v"(u[(c + 2) & 7]), "v"(u[(c + 3) & 7]));
            if (MODE == 20) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 21) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pk[c & 3]) : "v"(pk[(c + 1) & 3]), "v"(pk[(c + 2) & 3]));
            if (MODE == 22) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[c & 7]), "+v"(v[(c + 4) & 7]));
            if (MODE == 23) asm volatile("v_add_f32_dpp %0, %1, %2 row_ror:4 row_mask:0xf bank_mask:0xf" : "=v"(v[c & 7]) : "v"(x0), "v"(x1));
            if (MODE == 11) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[0]) : "v"(x0), "v"(x1));   // one dependent chain of fmacs
        }
    }
    long long w1 = wall_clock64();
    float s = pk[0][0] + pk[1][1] + pk[2][0] + pk[3][1]; for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i]; for (int i = 0; i < 4; ++i) s += a[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)t, (unsigned long long)(w1 - w0));
}
template <int MODE> void run(const char* name, int per_iter) {
    float* out; long long* t; (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&t, 8);
    const int iters = 200;
    for (int w : {1, 2, 4}) {
        for (int r = 0; r < 2; ++r) { (void)hipMemset(t, 0, 8); k<MODE><<<256, 256 * w>>>(out, t, 0.5f, iters); (void)hipDeviceSynchronize(); }
        long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
        const double ns = h * 10.0 / (iters * (double)per_iter);
        printf("%-28s %d wave(s)/SIMD: %6.2f ns per instruction per wave -> the SIMD issues one every %5.2f ns\n", name, w, ns, ns / w);
    }
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    run<0>("v_fmac_f32", 160); run<11>("v_fmac_f32 (dependent)", 160); run<9>("v_mul_f32", 160); run<1>("v_cvt_pk_f16_f32", 160); run<2>("v_fma_mix_f32", 160);
    run<3>("v_fma_mixlo_f16", 160); run<4>("v_cndmask_b32", 160); run<8>("v_cmp_gt_f32 |x|", 160); run<5>("v_permlane32_swap", 160); run<10>("v_mov_b32_dpp row_ror", 160);
    run<12>("v_cndmask_b32_e64 sgpr mask", 160); run<13>("v_cndmask_b32_e64 0, v, s", 160); run<14>("v_and_b32", 160); run<15>("v_add_f32", 160);
    run<16>("v_med3_f32", 160); run<17>("v_fma_mixhi_f16", 160); run<18>("v_cvt_f16_f32", 160); run<19>("v_perm_b32", 160); run<20>("v_max_f32", 160);
    run<21>("v_pk_mul_f32", 160); run<22>("v_permlane16_swap", 160); run<23>("v_add_f32_dpp row_ror", 160);
    run<6>("mfma_f32_16x16x32_f16 x4 acc", 48); run<7>("mfma_f32_16x16x32_f16 chain", 48);
    return 0;
}
