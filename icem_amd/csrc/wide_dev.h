// wide_dev.h -- device helpers shared by the wide-observation rollout kernels (k_rollout_wide.hip: exact f32 matrix pipe;
// k_rollout_wide_split.hip: 3-way bf16 split): the step cost of a trajectory read from its contraction vector in LDS.
// Anonymous namespace: every translation unit gets its own copy (same bits).
#pragma once
#include "cost_terms_dev.h"

namespace icem {
namespace {

// The step cost of one trajectory from its PRE-action contraction vector x = [obs (o) | action (d)], shared by the tile
// kernel and the row kernel (the same expression, so a row costs the same bits whichever of them scores it):
// icem_cost_spec, then -- cs.ext -- the icem_cost_terms that read the pre-action observation (health, the term list;
// `bad` = some entry of the observation is non-finite or outside the state box: the caller owns that sweep).  The
// difference term reads the NEXT observation, which these kernels write over x: `dold` takes obs[diff_idx] and the
// caller adds wide_diff_cost() once the step is done, before it accumulates.  The icem_cost_spec part travels by value
// (WideCost); the terms sit in device memory behind a pointer that is NULL when none is on and are copied to LDS once per
// workgroup.  By value in the argument block they cost the tile kernel 200 spilled scalar registers, and read through
// the pointer inside the step loop they left vector-memory loads pending on one path into the model loop, whose
// two-blocks-ahead requests then sat behind a vmcnt(0): 9 % of a launch either way, terms on or not (measured).
__device__ __forceinline__ void wide_stage_terms(CostArgs<float>& dst, const CostArgs<float>* src, int tid, int nthr) {
    if (src != nullptr)
        for (int e = tid; e < (int)(sizeof(CostArgs<float>) / 4); e += nthr)
            reinterpret_cast<int*>(&dst)[e] = reinterpret_cast<const int*>(src)[e];
    __syncthreads();
}
struct WideCost {
    int lin_idx, flip_idx;
    float ctrl_w, lin_w, flip_pen, flip_th;
};
__device__ __forceinline__ float wide_step_cost(const WideCost& b, bool ext, const CostArgs<float>& cs, const float* x, int o, int d,
                                                bool bad, float& dold) {
    float c = 0.f;
    if (b.flip_idx >= 0) {
        const float ang = x[b.flip_idx];
        c += (ang > b.flip_th) ? b.flip_pen : 0.f;
        c += (ang < -b.flip_th) ? b.flip_pen : 0.f;
    }
    float u = 0.f;
    for (int e = 0; e < d; ++e) u = __builtin_fmaf(x[o + e], x[o + e], u);
    c = __builtin_fmaf(u, b.ctrl_w, c);
    if (b.lin_w != 0.f) c = __builtin_fmaf(b.lin_w, x[b.lin_idx], c);  // a zero weight drops the term (icem_cost_spec)
    if (ext) {
        c += cost_terms<float, false>(cs, bad, [&](int idx) { return x[idx]; }, [&](int idx) { return x[idx]; });
        dold = cs.diff_idx >= 0 ? x[cs.diff_idx] : 0.f;
    }
    return c;
}
// The same cost with the icem_cost_terms spread over the FOUR lanes (j, g) of a trajectory (rollout_wide_kernel's B-operand
// layout): the icem_cost_spec part, the health term and `dold` in lane g = 0, term t of the list in lane t % 4 -- the
// caller sums the four shares (reduce_groups).  One lane walking the whole list is a chain of dependent LDS reads: 1.5 us
// of a 2.9 us step at o = 28 with two norm terms (FetchPickAndPlace; EXPERIMENTS.md R4.8).  Same terms, summed in another
// order: an f32 rounding apart from wide_step_cost.
__device__ __forceinline__ float wide_step_cost_lanes(const WideCost& b, const CostArgs<float>& cs, const float* x, int o, int d,
                                                      bool bad, int g, float& dold) {
    float c = 0.f;
    auto obs = [&](int idx) { return x[idx]; };
    if (g == 0) {
        if (b.flip_idx >= 0) {
            const float ang = x[b.flip_idx];
            c += (ang > b.flip_th) ? b.flip_pen : 0.f;
            c += (ang < -b.flip_th) ? b.flip_pen : 0.f;
        }
        float u = 0.f;
        for (int e = 0; e < d; ++e) u = __builtin_fmaf(x[o + e], x[o + e], u);
        c = __builtin_fmaf(u, b.ctrl_w, c);
        if (b.lin_w != 0.f) c = __builtin_fmaf(b.lin_w, x[b.lin_idx], c);
        if (cs.health_idx >= 0) {
            const float z = x[cs.health_idx];
            const bool in = cs.health_closed ? (cs.health_lo <= z && z <= cs.health_hi) : (cs.health_lo < z && z < cs.health_hi);
            c += (in && !bad) ? 0.f : cs.health_pen;
        }
        dold = cs.diff_idx >= 0 ? x[cs.diff_idx] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < ICEM_MAX_COST_TERMS / 4; ++t) {
        const int j = 4 * t + g;
        if (j < cs.n_terms) c += cost_term_value<float>(cs.terms[j], obs);
    }
    return c;
}
__device__ __forceinline__ float wide_diff_cost(const CostArgs<float>& cs, float next, float dold) {
    return cs.diff_w * (next - dold);
}
__device__ __forceinline__ bool wide_bad_entry(const CostArgs<float>& cs, float v, int k) {
    bool bad = !finite_val(v);
    if (cs.box_from >= 0 && k >= cs.box_from) bad |= !(cs.box_lo < v && v < cs.box_hi);
    return bad;
}
// sum / best / final over the steps (np.amin: a NaN step cost makes the trajectory's cost NaN)
__device__ __forceinline__ float wide_accumulate(float acc, float c, int t, int cost_mode) {
    if (t == 0 || cost_mode == 2) return c;
    if (cost_mode == 0) return acc + c;
    return (c < acc || c != c) ? c : acc;
}

}  // namespace
}  // namespace icem
