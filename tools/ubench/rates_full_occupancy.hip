// Throughput microbenchmarks at high occupancy: SIMD cycles per wave-instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4000
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, float seed) {
    float x[8]; unsigned u[8]; unsigned long long w[4];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + seed + i; u[i] = threadIdx.x * (2 * i + 1) + 12345u; }
    for (int i = 0; i < 4; ++i) w[i] = u[i];
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) { for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(1.0001f), "v"(seed)); }
        else if (MODE == 1) { for (int i = 0; i < 8; i += 2) { float2 v = {x[i], x[i+1]}; asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(v)); x[i] = v.x; x[i+1] = v.y; } }
        else if (MODE == 2) { for (int i = 0; i < 4; ++i) w[i] = (unsigned long long)(unsigned)w[i] * 0xD2511F53u + (w[i] >> 32); }
        else if (MODE == 3) { for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_logf(x[i]); }
        else if (MODE == 4) { for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_sinf(x[i]); }
        else if (MODE == 5) { for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_sqrtf(x[i]); }
        else if (MODE == 6) { for (int i = 0; i < 8; ++i) u[i] = (u[i] ^ u[(i + 1) & 7]) + 0x9E3779B9u; }
        else if (MODE == 7) { for (int i = 0; i < 8; ++i) u[i] = u[i] * 0xCD9E8D57u; }
        else if (MODE == 8) { for (int i = 0; i < 8; ++i) u[i] = __umulhi(u[i], 0xCD9E8D57u); }
        else if (MODE == 9) { for (int i = 0; i < 8; ++i) u[i] = (u[i] << 7) | (u[i] >> 25); }
        else if (MODE == 10) { for (int i = 0; i < 8; ++i) x[i] = (float)u[i] * x[i]; }
        else if (MODE == 11) { for (int i = 0; i < 8; ++i) u[i] = ((u[i] & 0xFFFFFFu) * 0x9E3779u) + u[(i+1)&7]; }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i]; for (int i = 0; i < 4; ++i) s += (float)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int n_instr) {
    float* out; (void)hipMalloc(&out, 1 << 24);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int wgs = 256 * 8;  // 8 waves per SIMD
    k<MODE><<<wgs, 256>>>(out, 0.5f); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k<MODE><<<wgs, 256>>>(out, 0.5f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double waves_per_simd = wgs * 4.0 / 1024.0;
    double cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * ITERS * n_instr);
    printf("%-28s %8.3f ms -> %6.2f SIMD cycles per wave-instruction (assuming 2.4 GHz)\n", name, ms, cyc);
    (void)hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 8); run<1>("v_pk_fma_f32", 4); run<2>("v_mad_u64_u32", 4); run<3>("v_log_f32", 8); run<4>("v_sin_f32", 8);
    run<5>("v_sqrt_f32", 8); run<6>("xor+add (v_xad_u32)", 8); run<7>("v_mul_lo_u32", 8); run<8>("v_mul_hi_u32", 8);
    run<9>("rotate (v_alignbit)", 8); run<10>("cvt_f32_u32+mul", 16); run<11>("v_mad_u32_u24", 8);
    return 0;
}
