// A LONE wave per SIMD: ns per vector instruction against the number of independent chains it interleaves (1 = every
// instruction waits for the one in front of it).  What the model step of the single-launch kernel is bound by at one tile per CU:
// if a dependent instruction costs more than an independent one, interleaving the step's chains (cost / 17th column / planes)
// in the SOURCE is worth an issue slot each; if not, only the instruction count matters.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lone_wave_ilp tools/ubench/lone_wave_ilp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int NACC> __global__ void k(float* out, long long* t, float seed, int iters) {
    float x0 = threadIdx.x * 1e-3f + seed, x1 = x0 + 1.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = x0 + i;
    unsigned u[8]; for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
    long long w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 160; ++c) {
            const int a = c % NACC;
            if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[a]) : "v"(x0), "v"(x1));
            if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[a]) : "v"(x0), "v"(x1));           // VOP3
            if (MODE == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(u[a]) : "v"(x1));
            if (MODE == 3) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(v[a]) : "v"(x0), "v"(x1));
            if (MODE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[a]) : "v"(x1));
            if (MODE == 5) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[a]) : "v"(u[7]));
        }
    }
    long long w1 = wall_clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) atomicMax((unsigned long long*)t, (unsigned long long)(w1 - w0));
}
template <int MODE, int NACC> double one(int waves) {
    float* out; long long* t; (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&t, 8);
    const int iters = 200;
    for (int r = 0; r < 2; ++r) { (void)hipMemset(t, 0, 8); k<MODE, NACC><<<256, 64 * waves>>>(out, t, 0.5f, iters); (void)hipDeviceSynchronize(); }
    long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    (void)hipFree(out); (void)hipFree(t);
    return h * 10.0 / (iters * 160.0);
}
template <int MODE> void run(const char* name) {
    // one wave per CU (SIMD 0 only), then four (one per SIMD)
    printf("%-18s lone wave, chains 1 / 2 / 4 / 8: %5.2f %5.2f %5.2f %5.2f ns per instruction   (one wave per SIMD, 8 chains: %5.2f)\n", name,
           one<MODE, 1>(1), one<MODE, 2>(1), one<MODE, 4>(1), one<MODE, 8>(1), one<MODE, 8>(4));
}
int main() {
    run<0>("v_fmac_f32"); run<1>("v_fma_f32 (VOP3)"); run<2>("v_cvt_pk_f16_f32"); run<3>("v_fma_mix_f32"); run<4>("v_cndmask_b32"); run<5>("v_xor_b32");
    return 0;
}
