// refit.h -- the one definition of the elite refit arithmetic (icem/controllers/icem.py:207-211),
// shared by every kernel that performs it so that all paths (stateless op, sharded merge, single-GPU
// merge) produce bit-identical mean/std.  Floating-point contraction is pinned off: every fused
// multiply-add is written explicitly.
#pragma once
#include <hip/hip_runtime.h>

namespace icem {

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float sqrt_t(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_t(double x) { return sqrt(x); }

// mean over K (sequential sum / K), population std (two-pass, ddof = 0), momentum alpha.
template <typename T, typename GetX>
__device__ __forceinline__ void refit_element(int K, T alpha, T old_mean, T old_std, GetX x, T& new_mean, T& new_std) {
#pragma clang fp contract(off)
    T s = (T)0;
    for (int r = 0; r < K; ++r) s = s + x(r);
    const T m = s / (T)K;
    T v = (T)0;
    for (int r = 0; r < K; ++r) {
        const T dx = x(r) - m;
        v = fma_t(dx, dx, v);
    }
    const T sd = sqrt_t(v / (T)K);
    const T one_m = (T)1 - alpha;
    new_mean = fma_t(one_m, m, alpha * old_mean);
    new_std = fma_t(one_m, sd, alpha * old_std);
}

// Same arithmetic, same order, for K values already in registers (x[r], r < K <= KMAX): every loop is
// static so nothing is indexed dynamically.
template <typename T, int KMAX>
__device__ __forceinline__ void refit_element_regs(int K, T alpha, T old_mean, T old_std, const T (&x)[KMAX], T& new_mean,
                                                   T& new_std) {
#pragma clang fp contract(off)
    T s = (T)0;
#pragma unroll
    for (int r = 0; r < KMAX; ++r)
        if (r < K) s = s + x[r];
    const T m = s / (T)K;
    T v = (T)0;
#pragma unroll
    for (int r = 0; r < KMAX; ++r) {
        if (r < K) {
            const T dx = x[r] - m;
            v = fma_t(dx, dx, v);
        }
    }
    const T sd = sqrt_t(v / (T)K);
    const T one_m = (T)1 - alpha;
    new_mean = fma_t(one_m, m, alpha * old_mean);
    new_std = fma_t(one_m, sd, alpha * old_std);
}

}  // namespace icem
