// cost_terms_dev.h -- the parametric step cost beyond icem_cost_spec (icem_cost_terms, include/icem_hip.h) as device
// code shared by the general rollout kernel, icem_trajectory_cost (generic_kernels.hip) and the wide rollout kernels
// (k_rollout_wide.hip).
#pragma once
#include <cfloat>
#include "cost_args.h"
#include "host_common.h"

namespace icem {

// __builtin_fma is the DOUBLE fma: route by type so the f32 kernels stay in f32.
__device__ __forceinline__ float fmad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }

__device__ __forceinline__ bool finite_val(float x) { return fabsf(x) <= FLT_MAX; }    // false for NaN / inf
__device__ __forceinline__ bool finite_val(double x) { return fabs(x) <= DBL_MAX; }
__device__ __forceinline__ float sqrt_val(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_val(double x) { return sqrt(x); }

// one entry of the term list: weight * f(obs) (times its gate)
template <typename T, typename Obs>
__device__ __forceinline__ T cost_term_value(const typename CostArgs<T>::Term& tm, Obs obs) {
    T f;
    if (tm.kind == ICEM_TERM_STEP_GT) {
        f = obs(tm.a) > tm.th ? (T)1 : (T)0;
    } else if (tm.kind == ICEM_TERM_SQ_OFFSET) {
        const T v = obs(tm.a) - tm.th;
        f = v * v;
    } else {
        T acc = (T)0;
        for (int m = 0; m < tm.len; ++m) {
            T v = obs(tm.a + m);
            if (tm.b >= 0) v -= obs(tm.b + m);
            acc = fmad(v, v, acc);
        }
        if (tm.kind == ICEM_TERM_SUMSQ) {
            f = acc;
        } else {
            const T r = sqrt_val(acc);
            f = tm.kind == ICEM_TERM_NORM ? r : tm.kind == ICEM_TERM_NORM_GT ? (r > tm.th ? (T)1 : (T)0) : (r < tm.th ? (T)1 : (T)0);
        }
    }
    if (tm.gate_idx >= 0) f *= obs(tm.gate_idx) > tm.gate_th ? (T)1 : (T)0;  // a product, as in the reference (NaN * 0 = NaN)
    return tm.w * f;
}

// The extra terms of one step given accessors for the pre- and post-action observation; `bad` = some observation
// entry is non-finite or outside Hopper's state box (computed by the caller, who owns the sweep over the row).
// WITH_DIFF = false leaves the difference term to the caller (a kernel that overwrites the observation in place adds
// diff_w * (next - saved) once the step is done).
template <typename T, bool WITH_DIFF = true, typename Obs, typename Nxt>
__device__ __forceinline__ T cost_terms(const CostArgs<T>& cs, bool bad, Obs obs, Nxt nxt) {
    T c = (T)0;
    if (WITH_DIFF && cs.diff_idx >= 0) c += cs.diff_w * (nxt(cs.diff_idx) - obs(cs.diff_idx));
    if (cs.health_idx >= 0) {
        const T z = obs(cs.health_idx);
        const bool in = cs.health_closed ? (cs.health_lo <= z && z <= cs.health_hi) : (cs.health_lo < z && z < cs.health_hi);
        c += (in && !bad) ? (T)0 : cs.health_pen;
    }
    // static term indices: a runtime index into the by-value argument block would move it to scratch
#pragma unroll
    for (int j = 0; j < ICEM_MAX_COST_TERMS; ++j) {
        if (j >= cs.n_terms) break;
        c += cost_term_value<T>(cs.terms[j], obs);
    }
    return c;
}

// host side (generic_kernels.hip): the handle's cost spec + terms in kernel-argument form
void fill_cost_args_f32(const icem_handle* h, CostArgs<float>& cs);

}  // namespace icem
