// icem_kernels.hip -- gfx950 kernels of the iCEM inner planning loop + the C ABI (include/icem_hip.h).
//
// Data layout in HBM (all C-contiguous, T = float or double):
//   actions [n, h, d]   the reference's `action_sequences` (icem/controllers/icem.py:73-79)
//   costs   [n]
//   mean/std [h, d], low/high [d]
//   W [h, HMAX]         colored-noise synthesis table, row t holds the h coefficients that turn the
//                       h white draws of one (trajectory, action-dim) row into sample t (zero padded)
//   records [world*K, 2+h*d]   {cost, gidx, actions[h*d]} -- what the ranks all-gather
//
// Kernels (one section each): sample_clip (K1), rollout_cost (K2), block top-k (K3),
// local_pack / merge_refit (K3+K4 of the fused step), small epilogue kernels.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/icem_hip.h"
#include "philox.h"
#include "icem_fused.h"
#include "icem_rssm.h"
#include "refit.h"
#include <type_traits>

namespace icem {

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define ICEM_HIP_TRY(expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ICEM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
    } while (0)

constexpr int WG = 256;          // 4 wavefronts of 64
constexpr int TOPK_CHUNK = 1024; // costs per workgroup in the block-level top-k

// __builtin_fma is the DOUBLE fma: route by type so the f32 kernels stay in f32.
__device__ __forceinline__ float fmad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <typename T>
__device__ __forceinline__ T inf_v() {
    return (T)INFINITY;
}

// (cost, index) lexicographic order; the index breaks ties (np.argmin / stable argsort semantics).
template <typename T>
__device__ __forceinline__ bool key_less(T ca, int ia, T cb, int ib) {
    return ca < cb || (ca == cb && ia < ib);
}

// ---------------------------------------------------------------------------------------------
// K1  colored-noise sampling + affine + clip          (icem.py:61-82 + colorednoise)
// ---------------------------------------------------------------------------------------------
// One thread per (trajectory, action-dim) row: it owns the h white draws of that row, applies the
// [h x h] synthesis (inverse real DFT with f^(-beta/2)/sigma folded in) with the table row as a
// wave-uniform (scalar) operand, and parks the h samples in an LDS tile laid out like the output,
// so the workgroup's slab of `actions` (tpw consecutive trajectories = one contiguous span) goes
// out as coalesced stores.  The reference's transpose([0,2,1]) is absorbed by the tile indexing.

template <typename T>
struct SampleArgs {
    int n, h, d, F, tpw;
    long long first_index;
    const T* W;
    const T* mean;
    const T* std;
    const T* low;
    const T* high;
    const T* zr;
    const T* zi;
    uint32_t seed_lo, seed_hi, off_lo, off_hi;
    int t_begin, row0_mean;
    int white;  // noise_beta <= 0 (icem.py:77): zr is randn[n, h, d], zi unused; W is the identity
    T* out;
};

template <typename T, int HMAX, int ROUNDS>
__device__ __forceinline__ void white_row(const SampleArgs<T>& a, int row_local, long long gi, int j, T (&g)[HMAX]) {
    if (a.zr != nullptr && a.white) {
#pragma unroll
        for (int m = 0; m < HMAX; ++m) g[m] = m < a.h ? a.zr[((size_t)row_local * a.h + m) * a.d + j] : (T)0;
    } else if (a.zr != nullptr) {
        const size_t base = ((size_t)row_local * a.d + j) * a.F;
#pragma unroll
        for (int m = 0; m < HMAX; ++m) {
            T v = (T)0;
            if (m < a.F)
                v = a.zr[base + m];
            else if (m < a.h)
                v = a.zi[base + (m - a.F + 1)];
            g[m] = v;
        }
    } else {
        Xoshiro128pp rng = row_stream<ROUNDS>((uint32_t)gi, (uint32_t)j, a.off_lo, a.off_hi, a.seed_lo, a.seed_hi);
#pragma unroll
        for (int m = 0; m < HMAX; m += 2) {
            if (m < a.h) {
                const uint32_t xa = rng.next();
                const uint32_t xb = rng.next();
                box_muller(xa, xb, g[m], g[m + 1]);
            } else {
                g[m] = g[m + 1] = (T)0;
            }
        }
    }
}

template <typename T, int HMAX, int ROUNDS>
__global__ __launch_bounds__(WG) void sample_clip_kernel(SampleArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    const int tid = threadIdx.x;
    const int hd = a.h * a.d;
    const int n_base = blockIdx.x * a.tpw;
    const int n_here = min(a.tpw, a.n - n_base);
    const int rows = n_here * a.d;
    if (tid < rows) {
        const int nl = tid / a.d;
        const int j = tid - nl * a.d;
        T g[HMAX];
        white_row<T, HMAX, ROUNDS>(a, n_base + nl, a.first_index + n_base + nl, j, g);
        const T lo = a.low[j], hi = a.high[j];
        for (int t = a.t_begin; t < a.h; ++t) {
            const T* __restrict__ w = a.W + (size_t)t * HMAX;
            T acc = (T)0;
#pragma unroll
            for (int m = 0; m < HMAX; ++m) acc = fmad(g[m], w[m], acc);
            T v = fmad(acc, a.std[t * a.d + j], a.mean[t * a.d + j]);
            v = v < lo ? lo : v;
            v = v > hi ? hi : v;
            tile[nl * hd + t * a.d + j] = v;
        }
    }
    __syncthreads();
    if (a.row0_mean && a.first_index + n_base == 0) {  // icem.py:87-88
        for (int e = tid; e < hd; e += WG) tile[e] = a.mean[e];
        __syncthreads();
    }
    const size_t base = (size_t)n_base * hd;
    const int total = n_here * hd;
    const int e_begin = a.t_begin * a.d;
    for (int e = tid; e < total; e += WG) {
        if (e_begin == 0 || (e % hd) >= e_begin) a.out[base + e] = tile[e];
    }
}

// ---------------------------------------------------------------------------------------------
// MpcCemStd (the CEM baseline, icem/controllers/mpc.py:142-327): truncated-normal sampling and its bounds
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double std_normal_cdf(double x) { return normcdf(x); }
__device__ __forceinline__ float std_normal_cdf(float x) { return normcdff(x); }
__device__ __forceinline__ double std_normal_icdf(double p) { return normcdfinv(p); }
__device__ __forceinline__ float std_normal_icdf(float p) { return normcdfinvf(p); }

// actions[i, t, j] = mean[t, j] + std[t, j] * ppf(u; lower[t, j], upper[t, j]) with the truncated standard normal's
// inverse CDF ppf(u; a, b) = Phi^-1(Phi(a) + u (Phi(b) - Phi(a)))  (scipy.stats.truncnorm.rvs, mpc.py:188-198).
// u: the caller's uniforms [n, h, d] (parity: scipy draws exactly that array), or, if null, word t of row (i, j)'s
// Philox / xoshiro stream mapped to (x + 0.5) * 2^-32.  One thread per (trajectory, dim) row.
template <typename T, int ROUNDS>
__global__ __launch_bounds__(WG) void sample_truncnorm_kernel(int n, int h, int d, long long first_index, const T* mean,
                                                             const T* std, const T* lower, const T* upper, const T* u,
                                                             uint32_t seed_lo, uint32_t seed_hi, uint32_t off_lo,
                                                             uint32_t off_hi, T* out) {
    const int row = blockIdx.x * WG + threadIdx.x;
    if (row >= n * d) return;
    const int i = row / d, j = row - i * d;
    Xoshiro128pp rng = row_stream<ROUNDS>((uint32_t)(first_index + i), (uint32_t)j, off_lo, off_hi, seed_lo, seed_hi);
    for (int t = 0; t < h; ++t) {
        const size_t e = ((size_t)i * h + t) * d + j;
        const uint32_t x = rng.next();
        const T uu = u ? u[e] : ((T)x + (T)0.5) * (T)2.3283064365386963e-10;
        const T pa = std_normal_cdf(lower[t * d + j]), pb = std_normal_cdf(upper[t * d + j]);
        const T z = std_normal_icdf(fmad(uu, pb - pa, pa));
        out[e] = fmad(z, std[t * d + j], mean[t * d + j]);
    }
}

// MpcRandom.sample_action_sequences (mpc.py:96-109): uniform actions held for a number of consecutive calls of
// sample() -- a call counter that runs over (trajectory, step) pairs and on across MPC steps.  Call c uses block
// 0 (the action drawn at construction) while c < freq, then block 1 + (c - freq) / (freq + 1).  u: the caller's
// uniforms [*, d] for blocks first_block.. (parity), or null: word 0 of block (b, j)'s Philox / xoshiro stream.
template <typename T, int ROUNDS>
__global__ __launch_bounds__(WG) void sample_piecewise_kernel(long long total, int d, long long call_offset, int freq,
                                                             long long first_block, const T* low, const T* high, const T* u,
                                                             uint32_t seed_lo, uint32_t seed_hi, T* out) {
    const long long e = (long long)blockIdx.x * WG + threadIdx.x;
    if (e >= total) return;
    const long long call = call_offset + e / d;
    const int j = (int)(e % d);
    const long long b = call < freq ? 0 : 1 + (call - freq) / (freq + 1);
    T uu;
    if (u) {
        uu = u[(b - first_block) * d + j];
    } else {
        Xoshiro128pp rng = row_stream<ROUNDS>((uint32_t)b, (uint32_t)j, (uint32_t)((unsigned long long)b >> 32), 0x52414E44u /* "RAND" */,
                                              seed_lo, seed_hi);
        uu = ((T)rng.next() + (T)0.5) * (T)2.3283064365386963e-10;
    }
    out[e] = fmad(high[j] - low[j], uu, low[j]);
}

// MpcCemStd._update_bounds (mpc.py:290-301), in place on std (like_levine) and into lower / upper [h, d]
template <typename T>
__global__ __launch_bounds__(WG) void cem_bounds_kernel(int hd, int d, int like_levine, const T* mean, T* std, const T* low,
                                                       const T* high, T* lower, T* upper) {
    const int e = blockIdx.x * WG + threadIdx.x;
    if (e >= hd) return;
    const int j = e % d;
    if (like_levine) {
        const T lb = (mean[e] - low[j]) / (T)2, ub = (high[j] - mean[e]) / (T)2;
        T s = lb < ub ? lb : ub;
        s = s < std[e] ? s : std[e];
        std[e] = s > (T)1e-8 ? s : (T)1e-8;
        lower[e] = (T)-2;
        upper[e] = (T)2;
    } else {
        lower[e] = (low[j] - mean[e]) / (std[e] + (T)1e-8);
        upper[e] = (high[j] - mean[e]) / (std[e] + (T)1e-8);
    }
}

// Raw Philox white noise in the reference's [n, d, F] x 2 layout (RNG known-answer tests).
template <typename T, int HMAX, int ROUNDS>
__global__ __launch_bounds__(WG) void philox_normals_kernel(SampleArgs<T> a, T* zr_out, T* zi_out) {
    const int row = blockIdx.x * WG + threadIdx.x;
    if (row >= a.n * a.d) return;
    const int nl = row / a.d;
    const int j = row - nl * a.d;
    T g[HMAX];
    white_row<T, HMAX, ROUNDS>(a, nl, a.first_index + nl, j, g);
    const size_t base = (size_t)row * a.F;
#pragma unroll
    for (int m = 0; m < HMAX; ++m) {
        if (m < a.F) {
            zr_out[base + m] = g[m];
            zi_out[base + m] = (T)0;
        }
    }
#pragma unroll
    for (int m = 0; m < HMAX; ++m) {
        if (m >= a.F && m < a.h) zi_out[base + (m - a.F + 1)] = g[m];
    }
}

// ---------------------------------------------------------------------------------------------
// K2  batched open-loop rollout + per-trajectory cost   (abstract_models.py:17-53,
//     abstract_controller.py:74-91, environments/mujoco.py:67-99 / 259-277)
// ---------------------------------------------------------------------------------------------
// One thread per trajectory; the observation lives in registers (O compile-time, zero padded),
// the model matrices are wave-uniform operands.  Cost is scored on the PRE-action observation.

template <typename T>
struct CostArgs {
    T ctrl_w, lin_w, flip_pen, flip_th;
    int lin_idx, flip_idx;
    // icem_cost_terms (include/icem_hip.h); ext = any of them on
    T diff_w, health_pen, health_lo, health_hi, box_lo, box_hi;
    int ext, diff_idx, health_idx, health_closed, box_from, n_terms;
    struct Term {
        T w, th, gate_th;
        int kind, a, b, len, gate_idx;
    } terms[ICEM_MAX_COST_TERMS];
};

__device__ __forceinline__ bool finite_val(float x) { return fabsf(x) <= FLT_MAX; }    // false for NaN / inf
__device__ __forceinline__ bool finite_val(double x) { return fabs(x) <= DBL_MAX; }
__device__ __forceinline__ float sqrt_val(float x) { return sqrtf(x); }
__device__ __forceinline__ double sqrt_val(double x) { return sqrt(x); }

// The extra terms of one step given accessors for the pre- and post-action observation; `bad` = some observation
// entry is non-finite or outside Hopper's state box (computed by the caller, who owns the sweep over the row).
template <typename T, typename Obs, typename Nxt>
__device__ __forceinline__ T cost_terms(const CostArgs<T>& cs, bool bad, Obs obs, Nxt nxt) {
    T c = (T)0;
    if (cs.diff_idx >= 0) c += cs.diff_w * (nxt(cs.diff_idx) - obs(cs.diff_idx));
    if (cs.health_idx >= 0) {
        const T z = obs(cs.health_idx);
        const bool in = cs.health_closed ? (cs.health_lo <= z && z <= cs.health_hi) : (cs.health_lo < z && z < cs.health_hi);
        c += (in && !bad) ? (T)0 : cs.health_pen;
    }
    // static term indices: a runtime index into the by-value argument block would move it to scratch
#pragma unroll
    for (int j = 0; j < ICEM_MAX_COST_TERMS; ++j) {
        if (j >= cs.n_terms) break;
        const typename CostArgs<T>::Term& tm = cs.terms[j];
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
        c += tm.w * f;
    }
    return c;
}

template <typename T>
struct RolloutArgs {
    int n, h, d, o;
    const T* A;  // [O, O] padded
    const T* B;  // [d, O] padded
    const T* obs0;
    const T* actions;
    T* costs;
    T* observations;  // nullable [n, h, o]
    CostArgs<T> cs;
    int cost_mode;
};

__device__ __forceinline__ float act_tanh(float x) { return tanhf(x); }
__device__ __forceinline__ double act_tanh(double x) { return tanh(x); }

template <typename T, int O, int KIND>
__global__ __launch_bounds__(WG) void rollout_cost_kernel(RolloutArgs<T> a) {
    const int n = blockIdx.x * WG + threadIdx.x;
    if (n >= a.n) return;
    T obs[O];
#pragma unroll
    for (int k = 0; k < O; ++k) obs[k] = k < a.o ? a.obs0[k] : (T)0;
    const T* __restrict__ act = a.actions + (size_t)n * a.h * a.d;
    const T* __restrict__ A = a.A;
    const T* __restrict__ B = a.B;
    T acc = (T)0;
    for (int t = 0; t < a.h; ++t) {
        T nxt[O];
#pragma unroll
        for (int i = 0; i < O; ++i) nxt[i] = (T)0;
#pragma unroll
        for (int k = 0; k < O; ++k) {
            const T ok = obs[k];
#pragma unroll
            for (int i = 0; i < O; ++i) nxt[i] = fmad(ok, A[k * O + i], nxt[i]);
        }
        T ctrl = (T)0;
        for (int j = 0; j < a.d; ++j) {
            const T aj = act[t * a.d + j];
            ctrl = fmad(aj, aj, ctrl);
#pragma unroll
            for (int i = 0; i < O; ++i) nxt[i] = fmad(aj, B[j * O + i], nxt[i]);
        }
        T lin = (T)0, ang = (T)0;
#pragma unroll
        for (int k = 0; k < O; ++k) {
            lin = (k == a.cs.lin_idx) ? obs[k] : lin;
            ang = (k == a.cs.flip_idx) ? obs[k] : ang;
        }
        T c = (T)0;
        if (a.cs.flip_idx >= 0) {
            c += (ang > a.cs.flip_th) ? a.cs.flip_pen : (T)0;
            c += (ang < -a.cs.flip_th) ? a.cs.flip_pen : (T)0;
        }
        c += a.cs.ctrl_w * ctrl;
        if (a.cs.lin_w != (T)0) c += a.cs.lin_w * lin;
        if (a.cs.ext) {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < O; ++k) {
                if (k >= a.o) continue;
                bad |= !finite_val(obs[k]);
                if (a.cs.box_from >= 0 && k >= a.cs.box_from) bad |= !(a.cs.box_lo < obs[k] && obs[k] < a.cs.box_hi);
            }
            auto pick = [&](const T* v, int idx) {
                T r = (T)0;
#pragma unroll
                for (int k = 0; k < O; ++k) r = (k == idx) ? v[k] : r;
                return r;
            };
            T post[O];
#pragma unroll
            for (int i = 0; i < O; ++i) post[i] = (KIND == ICEM_MODEL_TANH) ? act_tanh(nxt[i]) : nxt[i];
            c += cost_terms<T>(a.cs, bad, [&](int idx) { return pick(obs, idx); }, [&](int idx) { return pick(post, idx); });
        }
        if (t == 0 || a.cost_mode == ICEM_COST_FINAL)
            acc = c;
        else if (a.cost_mode == ICEM_COST_SUM)
            acc += c;
        else
            acc = (c < acc || c != c) ? c : acc;  // np.amin: a NaN step cost makes the trajectory's cost NaN
        if (a.observations != nullptr) {
            T* dst = a.observations + ((size_t)n * a.h + t) * a.o;
#pragma unroll
            for (int k = 0; k < O; ++k)
                if (k < a.o) dst[k] = obs[k];
        }
#pragma unroll
        for (int i = 0; i < O; ++i) obs[i] = (KIND == ICEM_MODEL_TANH) ? act_tanh(nxt[i]) : nxt[i];
    }
    a.costs[n] = acc;
}

template <typename T>
__global__ __launch_bounds__(WG) void cost_reduce_kernel(int n, int h, int mode, const T* step, T* costs) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const T* row = step + (size_t)i * h;
    T acc = row[0];
    for (int t = 1; t < h; ++t) {
        const T c = row[t];
        if (mode == ICEM_COST_SUM)
            acc += c;
        else if (mode == ICEM_COST_BEST)
            acc = (c < acc || c != c) ? c : acc;  // np.amin: a NaN step cost makes the trajectory's cost NaN
        else
            acc = c;
    }
    costs[i] = acc;
}

// trajectory_cost_fn (abstract_controller.py:74-91) over rollouts an external model left in HBM: one wavefront
// per trajectory.  Phase A, only when a term needs every entry of the observation (finite check / state box):
// the rows are swept coalesced (lanes across the observation), one ballot per step leaves a bit mask of the bad
// steps.  Phase B: lane t scores step t (its actions and the handful of observation entries the terms read).
// The step costs are then reduced in t order.
template <typename T>
struct TrajCostArgs {
    int n, h, d, o;
    const T* obs;
    const T* nxt;     // nullable
    long long ts, ss;
    const T* actions;
    T* costs;
    CostArgs<T> cs;
    int cost_mode, sweep;
};

template <typename T>
__device__ __forceinline__ bool bad_entry(const TrajCostArgs<T>& a, T v, int k) {
    bool bad = !finite_val(v);
    if (a.cs.box_from >= 0 && k >= a.cs.box_from) bad |= !(a.cs.box_lo < v && v < a.cs.box_hi);
    return bad;
}

template <typename T>
__global__ __launch_bounds__(WG) void trajectory_cost_kernel(TrajCostArgs<T> a) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
    if (n >= a.n) return;
    const T* __restrict__ traj = a.obs + (long long)n * a.ts;
    unsigned long long bad_steps = 0;   // h <= 64 (icem_create)
    if (a.sweep == 1) {
        // four rows x two 64-entry columns = eight unconditional loads in flight per lane; indices past the end
        // of a row / of the trajectory are clamped (a repeated entry changes nothing)
        for (int t = 0; t < a.h; t += 4) {
            const T* __restrict__ r0 = traj + (long long)min(t + 0, a.h - 1) * a.ss;
            const T* __restrict__ r1 = traj + (long long)min(t + 1, a.h - 1) * a.ss;
            const T* __restrict__ r2 = traj + (long long)min(t + 2, a.h - 1) * a.ss;
            const T* __restrict__ r3 = traj + (long long)min(t + 3, a.h - 1) * a.ss;
            bool b0 = false, b1 = false, b2 = false, b3 = false;
            for (int base = 0; base < a.o; base += 128) {
                const int k0 = min(base + lane, a.o - 1), k1 = min(base + 64 + lane, a.o - 1);
                const T v00 = r0[k0], v01 = r0[k1], v10 = r1[k0], v11 = r1[k1];
                const T v20 = r2[k0], v21 = r2[k1], v30 = r3[k0], v31 = r3[k1];
                b0 |= bad_entry(a, v00, k0);
                b0 |= bad_entry(a, v01, k1);
                b1 |= bad_entry(a, v10, k0);
                b1 |= bad_entry(a, v11, k1);
                b2 |= bad_entry(a, v20, k0);
                b2 |= bad_entry(a, v21, k1);
                b3 |= bad_entry(a, v30, k0);
                b3 |= bad_entry(a, v31, k1);
            }
            bad_steps |= (unsigned long long)(__ballot(b0) != 0) << (t & 63);
            bad_steps |= (unsigned long long)(__ballot(b1) != 0) << ((t + 1) & 63);   // rows past h repeat row h-1:
            bad_steps |= (unsigned long long)(__ballot(b2) != 0) << ((t + 2) & 63);   // their bits are never read
            bad_steps |= (unsigned long long)(__ballot(b3) != 0) << ((t + 3) & 63);
        }
    }
    T c = (T)0;
    if (lane < a.h) {
        const int t = lane;
        const T* __restrict__ row = traj + (long long)t * a.ss;
        const T* __restrict__ act = a.actions + ((long long)n * a.h + t) * a.d;
        T ctrl = (T)0;
        for (int j = 0; j < a.d; ++j) ctrl = fmad(act[j], act[j], ctrl);
        if (a.cs.flip_idx >= 0) {
            const T ang = row[a.cs.flip_idx];
            c += (ang > a.cs.flip_th) ? a.cs.flip_pen : (T)0;
            c += (ang < -a.cs.flip_th) ? a.cs.flip_pen : (T)0;
        }
        c += a.cs.ctrl_w * ctrl;
        if (a.cs.lin_w != (T)0) c += a.cs.lin_w * row[a.cs.lin_idx];
        if (a.cs.ext) {
            const T* __restrict__ nrow = a.nxt ? a.nxt + (long long)n * a.ts + (long long)t * a.ss : row;
            bool bad = (bad_steps >> t) & 1ull;
            if (a.sweep == 2)   // narrow observations: each lane checks its own row
                for (int k = 0; k < a.o; ++k) bad |= bad_entry(a, row[k], k);
            c += cost_terms<T>(a.cs, bad, [&](int idx) { return row[idx]; },
                               [&](int idx) { return nrow[idx]; });
        }
    }
    T acc = __shfl(c, 0);
    for (int t = 1; t < a.h; ++t) {
        const T ct = __shfl(c, t);
        if (a.cost_mode == ICEM_COST_SUM)
            acc += ct;
        else if (a.cost_mode == ICEM_COST_BEST)
            acc = (ct < acc || ct != ct) ? ct : acc;  // np.amin: NaN propagates
        else
            acc = ct;
    }
    if (lane == 0) a.costs[n] = acc;
}

// ---------------------------------------------------------------------------------------------
// K3  sorted top-k                                       (icem.py:199 argsort()[:K], :149 argmin)
// ---------------------------------------------------------------------------------------------
// Threshold selection: round r takes the smallest (cost, idx) key strictly greater than round
// r-1's winner, so nothing is mutated and the K winners come out already sorted.  Per round: a
// strided scan of the keys, a 64-lane butterfly, and one LDS hop across the 4 waves.

template <typename T>
__device__ __forceinline__ void wave_min_key(T& c, int& i) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const T oc = __shfl_xor(c, s, 64);
        const int oi = __shfl_xor(i, s, 64);
        if (key_less(oc, oi, c, i)) {
            c = oc;
            i = oi;
        }
    }
}

// getc(e)/geti(e) expose `cnt` keys; winners go to out_c/out_i[0..K) (any address space),
// `slot(e)` is returned through out_e (position of the winner in the key array) when non-null.
template <typename T, typename GetC, typename GetI>
__device__ __forceinline__ void block_select_sorted(int cnt, int K, GetC getc, GetI geti, T* out_c, int* out_i,
                                                    int* out_e, T* red_c, int* red_i, int* red_e) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    T pc = -inf_v<T>();
    int pi = -1;
    for (int r = 0; r < K; ++r) {
        T bc = inf_v<T>();
        int bi = INT_MAX, be = -1;
        for (int e = tid; e < cnt; e += WG) {
            const T c = getc(e);
            const int i = geti(e);
            const bool after_prev = c > pc || (c == pc && i > pi);
            if (after_prev && key_less(c, i, bc, bi)) {
                bc = c;
                bi = i;
                be = e;
            }
        }
        // reduce (bc, bi); carry `be` along with the winner
        T wc = bc;
        int wi = bi;
        wave_min_key(wc, wi);
        const bool mine = (wc == bc && wi == bi);
        // several lanes can hold the sentinel; the lowest such lane reports
        const unsigned long long m = __ballot(mine);
        if (mine && lane == __ffsll((long long)m) - 1) {
            red_c[wave] = bc;
            red_i[wave] = bi;
            red_e[wave] = be;
        }
        __syncthreads();
        T fc = red_c[0];
        int fi = red_i[0], fe = red_e[0];
#pragma unroll
        for (int w = 1; w < WG / 64; ++w) {
            if (key_less(red_c[w], red_i[w], fc, fi)) {
                fc = red_c[w];
                fi = red_i[w];
                fe = red_e[w];
            }
        }
        if (tid == 0) {
            out_c[r] = fc;
            out_i[r] = fi;
            if (out_e != nullptr) out_e[r] = fe;
        }
        pc = fc;
        pi = fi;
        __syncthreads();
    }
}

template <typename T>
__device__ __forceinline__ T nan_to_inf(T c) {
    return c != c ? inf_v<T>() : c;
}

// Stage 1: each workgroup reduces TOPK_CHUNK costs to its K best -> part_c/part_i[block*K + r].
template <typename T>
__global__ __launch_bounds__(WG) void topk_partial_kernel(int n, int K, const T* costs, T* part_c, int* part_i) {
    __shared__ T keys[TOPK_CHUNK];
    __shared__ T red_c[WG / 64];
    __shared__ int red_i[WG / 64];
    __shared__ int red_e[WG / 64];
    const int base = blockIdx.x * TOPK_CHUNK;
    const int cnt = min(TOPK_CHUNK, n - base);
    for (int e = threadIdx.x; e < cnt; e += WG) keys[e] = nan_to_inf(costs[base + e]);
    __syncthreads();
    block_select_sorted<T>(
        cnt, K, [&](int e) { return keys[e]; }, [&](int e) { return base + e; }, part_c + (size_t)blockIdx.x * K,
        part_i + (size_t)blockIdx.x * K, nullptr, red_c, red_i, red_e);
}

// Stage 2 (stand-alone API): one workgroup merges the partial lists.
template <typename T>
__global__ __launch_bounds__(WG) void topk_final_kernel(int cnt, int K, const T* part_c, const int* part_i, T* out_c,
                                                        int* out_i) {
    __shared__ T red_c[WG / 64];
    __shared__ int red_i[WG / 64];
    __shared__ int red_e[WG / 64];
    block_select_sorted<T>(
        cnt, K, [&](int e) { return part_c[e]; }, [&](int e) { return part_i[e]; }, out_c, out_i, nullptr, red_c,
        red_i, red_e);
}

// ---------------------------------------------------------------------------------------------
// K4  gather + refit                                     (icem.py:201-211)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(WG) void gather_refit_kernel(int hd, int K, T alpha, const T* actions, const int* idx,
                                                          T* mean, T* std, T* elites_out) {
    for (int e = blockIdx.x * WG + threadIdx.x; e < hd; e += gridDim.x * WG) {
        if (elites_out != nullptr)
            for (int r = 0; r < K; ++r) elites_out[(size_t)r * hd + e] = actions[(size_t)idx[r] * hd + e];
        T nm, ns;
        refit_element<T>(K, alpha, mean[e], std[e], [&](int r) { return actions[(size_t)idx[r] * hd + e]; }, nm, ns);
        mean[e] = nm;
        std[e] = ns;
    }
}

// get_action epilogue (icem.py:167-175) and beginning_of_rollout (icem.py:48-59).
template <typename T>
__global__ __launch_bounds__(WG) void shift_kernel(int h, int d, T init_std, T* mean, T* std, const T* low,
                                                   const T* high) {
    // single workgroup: read every element before any is overwritten
    const int hd = h * d;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tmp = reinterpret_cast<T*>(smem_raw);
    for (int e = threadIdx.x; e < hd; e += WG) tmp[e] = mean[e];
    __syncthreads();
    for (int e = threadIdx.x; e < hd; e += WG) {
        const int j = e % d;
        mean[e] = (e + d < hd) ? tmp[e + d] : tmp[e];
        std[e] = (high[j] - low[j]) / (T)2 * init_std;
    }
}

template <typename T>
__global__ __launch_bounds__(WG) void reset_kernel(int h, int d, T init_std, T* mean, T* std, const T* low,
                                                   const T* high) {
    const int hd = h * d;
    for (int e = blockIdx.x * WG + threadIdx.x; e < hd; e += gridDim.x * WG) {
        const int j = e % d;
        mean[e] = (high[j] + low[j]) / (T)2;
        std[e] = (high[j] - low[j]) / (T)2 * init_std;
    }
}

// ---------------------------------------------------------------------------------------------
// fused-step glue: shifted elites, local candidate packing, global merge + refit
// ---------------------------------------------------------------------------------------------

// icem.py:97-100: rows [0, n_reuse) of dst <- elites[e, 1:, :] (the last time step is sampled after).
template <typename T>
__global__ __launch_bounds__(WG) void shift_elites_kernel(int n_reuse, int h, int d, const T* elites, T* dst) {
    const int hd = h * d;
    const int total = n_reuse * (hd - d);
    for (int x = blockIdx.x * WG + threadIdx.x; x < total; x += gridDim.x * WG) {
        const int e = x / (hd - d);
        const int r = x - e * (hd - d);
        dst[(size_t)e * hd + r] = elites[(size_t)e * hd + d + r];
    }
}

template <typename T>
__device__ __forceinline__ int rec_gidx(const T* rec) {
    return reinterpret_cast<const int*>(rec + 1)[0];
}
template <typename T>
__device__ __forceinline__ void rec_set(T* rec, T cost, int gidx) {
    rec[0] = cost;
    rec[1] = (T)0;
    reinterpret_cast<int*>(rec + 1)[0] = gidx;
}

// One workgroup: pick this rank's K best among the block partials (sorted), translate local pool
// indices to global trajectory indices, and pack {cost, gidx, actions row} records.
//   local idx < n_loc           -> gidx = shard_lo + idx
//   local idx >= n_loc (shifted elites simulated at iteration 0) -> gidx = n_global + (idx - n_loc)
template <typename T>
__global__ __launch_bounds__(WG) void local_pack_kernel(int cnt, int K, int hd, int n_loc, int shard_lo, int n_global,
                                                        const T* part_c, const int* part_i, const T* actions,
                                                        T* records) {
    __shared__ T red_c[WG / 64];
    __shared__ int red_i[WG / 64];
    __shared__ int red_e[WG / 64];
    __shared__ T sel_c[ICEM_MAX_ELITES];
    __shared__ int sel_i[ICEM_MAX_ELITES];
    block_select_sorted<T>(
        cnt, K, [&](int e) { return part_c[e]; }, [&](int e) { return part_i[e]; }, sel_c, sel_i, nullptr, red_c,
        red_i, red_e);
    __syncthreads();
    const int rs = hd + 2;
    for (int r = 0; r < K; ++r) {
        const int li = sel_i[r];
        T* rec = records + (size_t)r * rs;
        if (li == INT_MAX) {  // fewer than K candidates on this rank
            if (threadIdx.x == 0) rec_set(rec, inf_v<T>(), INT_MAX);
            for (int e = threadIdx.x; e < hd; e += WG) rec[2 + e] = (T)0;
        } else {
            const int g = li < n_loc ? shard_lo + li : n_global + (li - n_loc);
            if (threadIdx.x == 0) rec_set(rec, sel_c[r], g);
            const T* src = actions + (size_t)li * hd;
            for (int e = threadIdx.x; e < hd; e += WG) rec[2 + e] = src[e];
        }
    }
}

template <typename T>
struct MergeArgs {
    int n_rec;        // world*K candidate records
    int n_keep;       // kept elites appended as candidates (icem.py:143-145)
    int K, h, d;
    int n_global;     // N_it: kept elite e gets gidx = n_global + e
    int last;         // last CEM iteration of the MPC step
    T alpha, init_std;
    const T* records;
    const T* elites_cur;       // [K, hd]
    const T* elites_cost_cur;  // [K]
    T* elites_next;
    T* elites_cost_next;
    const T* mean_in;  // distribution before the refit (momentum term)
    const T* std_in;
    T* mean;           // ... and where the new one goes (may alias)
    T* std;
    const T* low;
    const T* high;
    T* executed;
    T* best_cost;
};

// One workgroup: global sorted top-K over the gathered records (+ kept elites), new elite set,
// mean/std refit with momentum (icem.py:199-211); on the last iteration also the executed action,
// min cost, time shift of the mean and std reset (icem.py:163-177).
template <typename T>
__global__ __launch_bounds__(WG) void merge_refit_kernel(MergeArgs<T> a) {
    __shared__ T red_c[WG / 64];
    __shared__ int red_i[WG / 64];
    __shared__ int red_e[WG / 64];
    __shared__ T sel_c[ICEM_MAX_ELITES];
    __shared__ int sel_i[ICEM_MAX_ELITES];
    __shared__ int sel_e[ICEM_MAX_ELITES];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* new_mean = reinterpret_cast<T*>(smem_raw);  // [hd]
    const int hd = a.h * a.d;
    const int rs = hd + 2;
    const int cnt = a.n_rec + a.n_keep;
    block_select_sorted<T>(
        cnt, a.K,
        [&](int e) { return e < a.n_rec ? nan_to_inf(a.records[(size_t)e * rs]) : a.elites_cost_cur[e - a.n_rec]; },
        [&](int e) { return e < a.n_rec ? rec_gidx(a.records + (size_t)e * rs) : a.n_global + (e - a.n_rec); }, sel_c,
        sel_i, sel_e, red_c, red_i, red_e);
    __syncthreads();
    auto src_row = [&](int r) -> const T* {
        const int e = sel_e[r];
        return e < a.n_rec ? a.records + (size_t)e * rs + 2 : a.elites_cur + (size_t)(e - a.n_rec) * hd;
    };
    for (int e = threadIdx.x; e < hd; e += WG) {
        for (int r = 0; r < a.K; ++r) a.elites_next[(size_t)r * hd + e] = src_row(r)[e];
        T nm, ns;
        refit_element<T>(a.K, a.alpha, a.mean_in[e], a.std_in[e], [&](int r) { return src_row(r)[e]; }, nm, ns);
        if (!a.last) {
            a.mean[e] = nm;
            a.std[e] = ns;
        } else {
            new_mean[e] = nm;
        }
    }
    if ((int)threadIdx.x < a.K) a.elites_cost_next[threadIdx.x] = sel_c[threadIdx.x];
    if (a.last) {
        __syncthreads();
        for (int e = threadIdx.x; e < hd; e += WG) {
            const int j = e % a.d;
            a.mean[e] = (e + a.d < hd) ? new_mean[e + a.d] : new_mean[e];
            a.std[e] = (a.high[j] - a.low[j]) / (T)2 * a.init_std;
        }
        if ((int)threadIdx.x < a.d) a.executed[threadIdx.x] = src_row(0)[threadIdx.x];
        if (threadIdx.x == 0) a.best_cost[0] = sel_c[0];
    }
}

}  // namespace icem

// =============================================================================================
// host side: handle + C ABI
// =============================================================================================

using namespace icem;

// icem_get_action: executed action + best cost -> the host-mapped block, then the sequence flag (system scope)
__global__ void publish_result_kernel(const float* executed, const float* best_cost, int d, float* host_out, unsigned* flag,
                                      unsigned seq) {
    const int j = threadIdx.x;
    if (j < d) host_out[j] = executed[j];
    if (j == d) host_out[d] = best_cost[0];
    __threadfence_system();
    __syncthreads();
    if (j == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct icem_handle {
    icem_config cfg;
    int F = 0, HMAX = 0, hd = 0;
    size_t tsize = 4;
    void* W_dev = nullptr;
    int model_kind = 0, obs_dim = 0, O = 0;
    void* A_dev = nullptr;
    void* B_dev = nullptr;
    bool has_model = false, has_cost = false;
    icem_cost_spec cost;
    icem_cost_terms terms;
    bool has_terms = false;  // any term of icem_cost_terms switched on
    void* host_stage = nullptr;  // pinned, device-mapped block of icem_get_action [obs | action, best cost | flag]
    void* host_stage_dev = nullptr;
    unsigned io_seq = 0;
    std::vector<int> pop;
    int n_reuse = 0;
    int n_local_max = 0;
    // optional per-kernel timing with HIP events on the caller's stream (bench.py roofline leg)
    bool profiling = false;
    struct Span {
        int kind;
        long long units;
        hipEvent_t a, b;
    };
    bool use_fast = true;
    long long* dbg = nullptr;
    int fast_lists = 0;  // candidate lists written by the last matrix-pipe rollout (0 = generic path ran)
    // icem_plan_step (world == 1): an iteration's merge can ride in the prologue of the next iteration's launch,
    // which then reads the previous pool / lists / distribution while writing new ones -> ping-pong partners of
    // the caller's actions / workspace buffers and of mean | std, owned by the handle
    void* actions_alt = nullptr;
    void* ws_alt = nullptr;
    float* pp_stats = nullptr;          // [2][2 * hd]
    bool defer_merge = false;           // plan_iter_merge: stash the merge instead of launching it
    bool pm_pending = false;            // a stashed merge waits for the next local launch
    MergeSingleArgs pm_args;
    float* merge_mean_out = nullptr;    // where the next merge writes mean / std (nullptr: in place)
    float* merge_std_out = nullptr;
    // world > 1 with icem_set_merge_deferral(on): the same folding across the split calls; the distribution of the
    // running MPC step lives at cur_mean / cur_std (the caller's buffers or pp_stats)
    bool deferral = false;
    float* cur_mean = nullptr;
    float* cur_std = nullptr;
    // permuted, padded model of the matrix-pipe rollout: column 0 = obs[lin_idx], column 1 = obs[flip_idx]
    void* Mp_dev = nullptr;
    void* perm_dev = nullptr;
    int flip_col = -1;
    bool fast_model_ready = false;
    std::vector<double> A_host, B_host;
    std::vector<Span> spans;
    std::vector<hipEvent_t> free_events;
};

namespace {
struct ProfScope {
    icem_handle* h;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    int kind;
    long long units;
    static hipEvent_t get(icem_handle* h) {
        if (!h->free_events.empty()) {
            hipEvent_t e = h->free_events.back();
            h->free_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    ProfScope(const icem_handle* hc, int kind_, long long units_, hipStream_t st_)
        : h(const_cast<icem_handle*>(hc)), st(st_), kind(kind_), units(units_) {
        if (!h->profiling) return;
        a = get(h);
        b = get(h);
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        h->spans.push_back({kind, units, a, b});
    }
};
}  // namespace

namespace {

void psd_scale_host(int h, double beta, std::vector<double>& s, double& sigma) {
    // colorednoise.powerlaw_psd_gaussian (third-party, call site icem.py:73): f = rfftfreq(h),
    // DC takes the first bin's value, s = f^(-beta/2), sigma = 2*sqrt(sum w^2)/h.
    const int F = h / 2 + 1;
    s.resize(F);
    for (int k = 0; k < F; ++k) s[k] = (double)k * (1.0 / (double)h);
    const double fmin = 1.0 / (double)h;
    int ix = 0;
    for (int k = 0; k < F; ++k) ix += s[k] < fmin ? 1 : 0;
    if (ix && ix < F)
        for (int k = 0; k < ix; ++k) s[k] = s[ix];
    for (int k = 0; k < F; ++k) s[k] = std::pow(s[k], -beta / 2.0);
    double acc = 0.0;
    for (int k = 1; k < F; ++k) {
        double w = s[k];
        if (k == F - 1) w *= (1 + (h % 2)) / 2.0;
        acc += w * w;
    }
    sigma = 2.0 * std::sqrt(acc) / (double)h;
}

void noise_tables(int h, double beta, std::vector<double>& cr, std::vector<double>& ci) {
    std::vector<double> s;
    double sigma;
    psd_scale_host(h, beta, s, sigma);
    const int F = h / 2 + 1;
    cr.assign((size_t)F * h, 0.0);
    ci.assign((size_t)F * h, 0.0);
    for (int k = 0; k < F; ++k) {
        double mult = 2.0;
        if (k == 0 || (h % 2 == 0 && k == F - 1)) mult = 1.0;
        const double amp = mult * s[k] / ((double)h * sigma);
        const bool imag_dropped = (k == 0) || (h % 2 == 0 && k == F - 1);
        for (int t = 0; t < h; ++t) {
            const double ang = 2.0 * M_PI * (double)k * (double)t / (double)h;
            cr[(size_t)k * h + t] = amp * std::cos(ang);
            ci[(size_t)k * h + t] = imag_dropped ? 0.0 : -amp * std::sin(ang);
        }
    }
}

std::vector<int> population_sizes(const icem_config& c) {
    std::vector<int> out;
    int n = c.num_traj;
    for (int i = 0; i < c.opt_iters; ++i) {
        if (i > 0) n = std::max(c.elites_size * 2, (int)((double)n / c.factor_decrease));
        out.push_back(n);
    }
    return out;
}

template <typename T>
int upload(void** dev, const std::vector<double>& host) {
    std::vector<T> tmp(host.size());
    for (size_t i = 0; i < host.size(); ++i) tmp[i] = (T)host[i];
    if (*dev) {
        (void)hipFree(*dev);
        *dev = nullptr;
    }
    ICEM_HIP_TRY(hipMalloc(dev, tmp.size() * sizeof(T)));
    ICEM_HIP_TRY(hipMemcpy(*dev, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return ICEM_OK;
}

int pick_O(int o) {
    const int sizes[] = {8, 16, 17, 18, 24, 32};
    for (int s : sizes)
        if (o <= s) return s;
    return -1;
}

inline int shard_chunk(int n_global, int world) { return (n_global + world - 1) / world; }

// ---- typed launchers -------------------------------------------------------------------------

template <typename T>
SampleArgs<T> make_sample_args(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                               const void* low, const void* high, const void* zr, const void* zi, uint64_t offset,
                               int t_begin, int row0_mean, void* out) {
    SampleArgs<T> a;
    a.n = n;
    a.h = h->cfg.horizon;
    a.d = h->cfg.act_dim;
    a.F = h->F;
    a.tpw = std::max(1, WG / a.d);
    a.first_index = first_index;
    a.W = (const T*)h->W_dev;
    a.mean = (const T*)mean;
    a.std = (const T*)std;
    a.low = (const T*)low;
    a.high = (const T*)high;
    a.zr = (const T*)zr;
    a.zi = (const T*)zi;
    a.seed_lo = (uint32_t)h->cfg.seed;
    a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    a.off_lo = (uint32_t)offset;
    a.off_hi = (uint32_t)(offset >> 32);
    a.t_begin = t_begin;
    a.row0_mean = row0_mean;
    a.white = h->cfg.noise_beta <= 0 ? 1 : 0;
    a.out = (T*)out;
    return a;
}

template <typename T>
int launch_sample(const icem_handle* h, const SampleArgs<T>& a, hipStream_t st) {
    if (a.n <= 0) return ICEM_OK;
    const int grid = (a.n + a.tpw - 1) / a.tpw;
    const size_t lds = (size_t)a.tpw * a.h * a.d * sizeof(T);
    ProfScope prof(h, ICEM_K_SAMPLE, (long long)a.n * (a.h - a.t_begin), st);
    const bool r7 = h->cfg.rng_rounds == 7;
    if (h->HMAX == 32) {
        if (r7)
            hipLaunchKernelGGL((sample_clip_kernel<T, 32, 7>), dim3(grid), dim3(WG), lds, st, a);
        else
            hipLaunchKernelGGL((sample_clip_kernel<T, 32, 10>), dim3(grid), dim3(WG), lds, st, a);
    } else {
        if (r7)
            hipLaunchKernelGGL((sample_clip_kernel<T, 64, 7>), dim3(grid), dim3(WG), lds, st, a);
        else
            hipLaunchKernelGGL((sample_clip_kernel<T, 64, 10>), dim3(grid), dim3(WG), lds, st, a);
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

template <typename T, int KIND>
int launch_rollout_k(const icem_handle* h, const RolloutArgs<T>& a, hipStream_t st) {
    const int grid = (a.n + WG - 1) / WG;
    ProfScope prof(h, ICEM_K_ROLLOUT, (long long)a.n * a.h, st);
    switch (h->O) {
#define ICEM_CASE(OV)                                                                                  \
    case OV:                                                                                           \
        hipLaunchKernelGGL((rollout_cost_kernel<T, OV, KIND>), dim3(grid), dim3(WG), 0, st, a);        \
        break;
        ICEM_CASE(8)
        ICEM_CASE(16)
        ICEM_CASE(17)
        ICEM_CASE(18)
        ICEM_CASE(24)
        ICEM_CASE(32)
#undef ICEM_CASE
        default:
            return fail(ICEM_E_UNSUPPORTED, "obs_dim not compiled");
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

template <typename T>
void fill_cost_args(const icem_handle* h, CostArgs<T>& cs) {
    cs.ctrl_w = (T)h->cost.ctrl_weight;
    cs.lin_w = (T)h->cost.lin_weight;
    cs.flip_pen = (T)h->cost.flip_penalty;
    cs.flip_th = (T)h->cost.flip_thresh;
    cs.lin_idx = h->cost.lin_idx;
    cs.flip_idx = h->cost.flip_idx;
    const icem_cost_terms& t = h->terms;
    cs.ext = h->has_terms ? 1 : 0;
    cs.diff_w = (T)t.diff_weight;
    cs.health_pen = (T)t.health_penalty;
    cs.health_lo = (T)t.health_lo;
    cs.health_hi = (T)t.health_hi;
    cs.box_lo = (T)t.box_lo;
    cs.box_hi = (T)t.box_hi;
    cs.diff_idx = h->has_terms ? t.diff_idx : -1;
    cs.health_idx = h->has_terms ? t.health_idx : -1;
    cs.health_closed = t.health_closed;
    cs.box_from = (h->has_terms && t.health_idx >= 0) ? t.box_from : -1;
    cs.n_terms = h->has_terms ? t.n_terms : 0;
    for (int j = 0; j < ICEM_MAX_COST_TERMS; ++j) {
        const icem_cost_term& tm = t.terms[j];
        cs.terms[j].w = (T)tm.weight;
        cs.terms[j].th = (T)tm.thresh;
        cs.terms[j].gate_th = (T)tm.gate_thresh;
        cs.terms[j].kind = tm.kind;
        cs.terms[j].a = tm.a;
        cs.terms[j].b = tm.b;
        cs.terms[j].len = tm.len;
        cs.terms[j].gate_idx = tm.gate_idx;
    }
}

// every index a cost term reads lies inside an observation of width o
const char* cost_indices_error(const icem_handle* h, int o) {
    if (h->cost.lin_idx < 0 || h->cost.lin_idx >= o || h->cost.flip_idx >= o) return "cost index outside the observation";
    if (!h->has_terms) return nullptr;
    const icem_cost_terms& t = h->terms;
    if (t.diff_idx >= o || t.health_idx >= o || t.box_from >= o) return "cost term index outside the observation";
    for (int j = 0; j < t.n_terms; ++j) {
        const icem_cost_term& tm = t.terms[j];
        if (tm.a < 0 || tm.a + tm.len > o || (tm.b >= 0 && tm.b + tm.len > o) || tm.gate_idx >= o)
            return "cost term slice outside the observation";
    }
    return nullptr;
}

template <typename T>
int launch_trajectory_cost(const icem_handle* h, int n, int o, const void* obs, const void* nxt, long long ts,
                           long long ss, const void* actions, void* costs, hipStream_t st) {
    TrajCostArgs<T> a;
    a.n = n;
    a.h = h->cfg.horizon;
    a.d = h->cfg.act_dim;
    a.o = o;
    a.obs = (const T*)obs;
    a.nxt = (const T*)nxt;
    a.ts = ts;
    a.ss = ss;
    a.actions = (const T*)actions;
    a.costs = (T*)costs;
    fill_cost_args<T>(h, a.cs);
    a.cost_mode = h->cfg.cost_mode;
    // all_finite(obs) / the state box are part of `unhealthy` only; 1: coalesced sweep, 2: per-lane rows (narrow obs)
    a.sweep = a.cs.health_idx < 0 ? 0 : (o <= 32 ? 2 : 1);
    hipLaunchKernelGGL((trajectory_cost_kernel<T>), dim3((n + WG / 64 - 1) / (WG / 64)), dim3(WG), 0, st, a);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

template <typename T>
int launch_rollout(const icem_handle* h, int n, const void* obs0, const void* actions, void* costs, void* observations,
                   hipStream_t st) {
    if (n <= 0) return ICEM_OK;
    RolloutArgs<T> a;
    a.n = n;
    a.h = h->cfg.horizon;
    a.d = h->cfg.act_dim;
    a.o = h->obs_dim;
    a.A = (const T*)h->A_dev;
    a.B = (const T*)h->B_dev;
    a.obs0 = (const T*)obs0;
    a.actions = (const T*)actions;
    a.costs = (T*)costs;
    a.observations = (T*)observations;
    fill_cost_args<T>(h, a.cs);
    a.cost_mode = h->cfg.cost_mode;
    return h->model_kind == ICEM_MODEL_TANH ? launch_rollout_k<T, ICEM_MODEL_TANH>(h, a, st)
                                            : launch_rollout_k<T, ICEM_MODEL_LINEAR>(h, a, st);
}

inline int topk_blocks(int n) { return (n + TOPK_CHUNK - 1) / TOPK_CHUNK; }

template <typename T>
void split_partial_ws(void* ws, int nblk, int K, T** pc, int** pi) {
    *pc = (T*)ws;
    *pi = (int*)((unsigned char*)ws + (size_t)nblk * K * sizeof(T));
}

template <typename T>
int launch_topk(int n, int K, const void* costs, void* out_c, int* out_i, void* ws, hipStream_t st) {
    const int nblk = topk_blocks(n);
    T* pc;
    int* pi;
    split_partial_ws<T>(ws, nblk, K, &pc, &pi);
    hipLaunchKernelGGL((topk_partial_kernel<T>), dim3(nblk), dim3(WG), 0, st, n, K, (const T*)costs, pc, pi);
    hipLaunchKernelGGL((topk_final_kernel<T>), dim3(1), dim3(WG), 0, st, nblk * K, K, (const T*)pc, (const int*)pi,
                       (T*)out_c, out_i);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

// Build the permuted, zero-padded [A ; B] operand of the matrix-pipe rollout (lazily: it depends on
// both icem_set_model and icem_set_cost).  Observation entries are reordered so that the linear cost
// term reads column 0 and the flip term column 0 or 1 -- static registers in the kernel.
int ensure_fast_model(icem_handle* h) {
    if (h->fast_model_ready) return ICEM_OK;
    const int O = h->O, o = h->obs_dim, d = h->cfg.act_dim;
    // layout of Tile16 (icem_fused.hip): O <= 20 -> one 16-column matrix-pipe tile + extra columns, Mp [O + d + 1, ceil4(O)];
    // O > 20 -> two tiles, observation block padded to 32 rows / columns, Mp [32 + d + 1, 32]
    const bool two = O > 20;
    const int OP = two ? 32 : O;
    const int CT4 = two ? 32 : ((O + 3) / 4) * 4;
    std::vector<int> perm;
    perm.push_back(h->cost.lin_idx);
    h->flip_col = -1;
    if (h->cost.flip_idx >= 0) {
        if (h->cost.flip_idx == h->cost.lin_idx) {
            h->flip_col = 0;
        } else {
            perm.push_back(h->cost.flip_idx);
            h->flip_col = 1;
        }
    }
    for (int k = 0; k < O; ++k)
        if (std::find(perm.begin(), perm.end(), k) == perm.end()) perm.push_back(k);
    std::vector<float> Mp((size_t)(OP + d + 1) * CT4, 0.f);  // + one zero row (contraction slots without an entry)
    auto Aat = [&](int r, int c) { return (r < o && c < o) ? h->A_host[(size_t)r * o + c] : 0.0; };
    auto Bat = [&](int j, int c) { return c < o ? h->B_host[(size_t)j * o + c] : 0.0; };
    for (int k = 0; k < O; ++k)
        for (int c = 0; c < O; ++c) Mp[(size_t)k * CT4 + c] = (float)Aat(perm[k], perm[c]);
    for (int j = 0; j < d; ++j)
        for (int c = 0; c < O; ++c) Mp[(size_t)(OP + j) * CT4 + c] = (float)Bat(j, perm[c]);
    for (int k = 0; k < (int)perm.size(); ++k)
        if (k >= o || perm[k] >= o) perm[k] = 31;  // padding columns start from the zero slot of the staged observation
    perm.resize(32, 31);
    if (h->Mp_dev) (void)hipFree(h->Mp_dev);
    if (h->perm_dev) (void)hipFree(h->perm_dev);
    ICEM_HIP_TRY(hipMalloc(&h->Mp_dev, Mp.size() * sizeof(float)));
    ICEM_HIP_TRY(hipMalloc(&h->perm_dev, perm.size() * sizeof(int)));
    ICEM_HIP_TRY(hipMemcpy(h->Mp_dev, Mp.data(), Mp.size() * sizeof(float), hipMemcpyHostToDevice));
    ICEM_HIP_TRY(hipMemcpy(h->perm_dev, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice));
    h->fast_model_ready = true;
    return ICEM_OK;
}

bool fast_rollout_ok(const icem_handle* h, int K) {
    if (h->has_terms) return false;  // the extra cost terms live in the general kernel
    if (h->cost.lin_weight == 0.0) return false;  // ... and so does a cost without the linear term (dropped, not 0 * obs)
    return h->use_fast && h->cfg.dtype == ICEM_F32 && h->has_model && h->has_cost &&
           fast_rollout_supported(h->cfg.horizon, h->cfg.act_dim, h->O, K);
}

FastRolloutArgs fast_rollout_args(const icem_handle* h, int n_rows, int n_cand, int K, const void* obs0,
                                  const void* actions, void* costs, float* part_c, int* part_i) {
    FastRolloutArgs a;
    a.n_rows = n_rows;
    a.n_cand = n_cand;
    a.K = K;
    a.o = h->obs_dim;
    a.cost_mode = h->cfg.cost_mode;
    a.Mp = (const float*)h->Mp_dev;
    a.perm = (const int*)h->perm_dev;
    a.obs0 = (const float*)obs0;
    a.ctrl_w = (float)h->cost.ctrl_weight;
    a.lin_w = (float)h->cost.lin_weight;
    a.flip_pen = (float)h->cost.flip_penalty;
    a.flip_th = (float)h->cost.flip_thresh;
    a.flip_col = h->flip_col;
    a.actions = (const float*)actions;
    a.costs = (float*)costs;
    a.part_c = part_c;
    a.part_i = part_i;
    a.part_k = nullptr;
    a.dbg = h->dbg;
    return a;
}

// rows -> costs (+ one sorted candidate list per workgroup when K > 0); returns the number of candidate lists
int launch_fast_rollout(icem_handle* h, int n_rows, int n_cand, int K, const void* obs0, const void* actions,
                        void* costs, float* part_c, int* part_i, hipStream_t st, int* lists_out,
                        unsigned long long* part_k = nullptr) {
    int rc = ensure_fast_model(h);
    if (rc) return rc;
    FastRolloutArgs a = fast_rollout_args(h, n_rows, n_cand, K, obs0, actions, costs, part_c, part_i);
    a.part_k = part_k;
    const int grid = rollout_lists(h->cfg.horizon, h->cfg.act_dim, h->O, n_rows);
    {
        ProfScope prof(h, ICEM_K_ROLLOUT, (long long)n_rows * h->cfg.horizon, st);
        launch_rollout16(a, h->cfg.horizon, h->cfg.act_dim, h->O, h->model_kind, st);
    }
    ICEM_HIP_TRY(hipGetLastError());
    if (lists_out) *lists_out = grid;
    return ICEM_OK;
}

bool fast_sample_ok(const icem_handle* h) {
    return h->use_fast && h->cfg.dtype == ICEM_F32 && fast_sample_supported(h->cfg.horizon, h->cfg.act_dim);
}

FastSampleArgs fast_sample_args(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                                const void* low, const void* high, uint64_t offset, int row0_mean, void* out,
                                int n_shift, const void* elites_src, uint64_t offset2) {
    FastSampleArgs a;
    a.n = n;
    a.h = h->cfg.horizon;
    a.d = h->cfg.act_dim;
    a.first_index = first_index;
    a.W = (const float*)h->W_dev;
    a.mean = (const float*)mean;
    a.std = (const float*)std;
    a.low = (const float*)low;
    a.high = (const float*)high;
    a.seed_lo = (uint32_t)h->cfg.seed;
    a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    a.off_lo = (uint32_t)offset;
    a.off_hi = (uint32_t)(offset >> 32);
    a.row0_mean = row0_mean;
    a.out = (float*)out;
    a.n_shift = n_shift;
    a.elites_src = (const float*)elites_src;
    a.off2_lo = (uint32_t)offset2;
    a.off2_hi = (uint32_t)(offset2 >> 32);
    a.white = h->cfg.noise_beta <= 0 ? 1 : 0;
    return a;
}

int launch_fast_sample(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                       const void* low, const void* high, uint64_t offset, int row0_mean, void* out, hipStream_t st,
                       int n_shift = 0, const void* elites_src = nullptr, uint64_t offset2 = 0) {
    if (n <= 0 && n_shift <= 0) return ICEM_OK;
    const FastSampleArgs a = fast_sample_args(h, n, first_index, mean, std, low, high, offset, row0_mean, out, n_shift,
                                              elites_src, offset2);
    {
        ProfScope prof(h, ICEM_K_SAMPLE, (long long)n * a.h, st);
        launch_sample_folded(a, h->cfg.rng_rounds, st);
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

// Can the f32 launch of an iteration with n_rows local rows (no shifted elites) carry a merge in its prologue?
// (single-launch kernel with <= 4 rollout waves, or the sampler of the two-kernel path)
bool prologue_possible(const icem_handle* h, int n_rows) {
    const icem_config& c = h->cfg;
    const int K = c.num_elites;
    if (c.dtype != ICEM_F32 || !fast_rollout_ok(h, K) || !fast_sample_ok(h) || n_rows <= 0) return false;
    if (sample_rollout_lists(c.horizon, c.act_dim, h->O, c.rng_rounds, n_rows) > 0)
        return sample_rollout_merge_ok(c.horizon, c.act_dim, h->O, c.rng_rounds, n_rows, K);
    return sample_folded_merge_ok(c.horizon, c.act_dim, c.rng_rounds, K);
}

// a stashed merge that found no launch to ride in
int launch_pending_merge(icem_handle* h, hipStream_t st) {
    const MergeSingleArgs& m = h->pm_args;
    if (m.records == nullptr) {
        ProfScope prof(h, ICEM_K_MERGE_REFIT, m.n_lists * m.K + m.n_keep, st);
        launch_merge_single(m, st);
    } else {
        MergeArgs<float> a;
        a.n_rec = m.n_rec;
        a.n_keep = m.n_keep;
        a.K = m.K;
        a.h = m.h;
        a.d = m.d;
        a.n_global = m.n_global;
        a.last = 0;
        a.alpha = m.alpha;
        a.init_std = m.init_std;
        a.records = m.records;
        a.elites_cur = m.elites_cur;
        a.elites_cost_cur = m.elites_cost_cur;
        a.elites_next = m.elites_next;
        a.elites_cost_next = m.elites_cost_next;
        a.mean_in = m.mean;
        a.std_in = m.std;
        a.mean = m.mean_out;
        a.std = m.std_out;
        a.low = m.low;
        a.high = m.high;
        a.executed = m.executed;
        a.best_cost = m.best_cost;
        ProfScope prof(h, ICEM_K_MERGE_REFIT, a.n_rec + a.n_keep, st);
        hipLaunchKernelGGL((merge_refit_kernel<float>), dim3(1), dim3(WG), (size_t)m.h * m.d * sizeof(float), st, a);
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

template <typename T>
int plan_iter_local_t(icem_handle* h, const icem_plan_buffers* b, int mpc_step, int it, hipStream_t st) {
    const icem_config& c = h->cfg;
    const int hd = h->hd, K = c.num_elites;
    const int n_global = h->pop[it];
    const int chunk = shard_chunk(n_global, c.world);
    const int lo = std::min(n_global, c.rank * chunk);
    const int n_loc = std::max(0, std::min(n_global - lo, chunk));
    const uint64_t call_base = (uint64_t)mpc_step * (uint64_t)(c.opt_iters + 1);
    const bool last = it == c.opt_iters - 1;
    T* actions = (T*)b->actions;
    // shifted elites, simulated at iteration 0 of every MPC step but the first (icem.py:131-137)
    int n_extra = 0;
    const T* shift_src = nullptr;
    bool shift_in_sampler = false;
    if (it == 0 && c.shift_elites && mpc_step > 0 && h->n_reuse > 0) {
        n_extra = h->n_reuse;
        const int g = (int)(((long long)mpc_step * c.opt_iters) & 1);  // elite buffer holding the previous step's set
        shift_src = (const T*)b->elites + (size_t)g * K * hd;
        // the fast sampler prepares them in an extra workgroup of its own launch
        shift_in_sampler = std::is_same<T, float>::value && b->z_r == nullptr && b->z_r_shift == nullptr &&
                           fast_rollout_ok(h, K) && fast_sample_ok(h) &&
                           n_extra * c.act_dim <= 256;
        if (!shift_in_sampler) {
            T* dst = actions + (size_t)n_loc * hd;
            hipLaunchKernelGGL((shift_elites_kernel<T>), dim3(1), dim3(WG), 0, st, n_extra, c.horizon, c.act_dim, shift_src, dst);
            SampleArgs<T> a = make_sample_args<T>(h, n_extra, 0, b->mean, b->std, b->low, b->high, b->z_r_shift, b->z_i_shift,
                                                  call_base + (uint64_t)c.opt_iters, c.horizon - 1, 0, dst);
            int rc = launch_sample<T>(h, a, st);
            if (rc) return rc;
        }
    }
    // candidates: the shard, plus the shifted elites on rank 0 only (they are replicated)
    const int n_cand = n_loc + (c.rank == 0 ? n_extra : 0);
    const int row0 = (last && c.use_mean_actions) ? 1 : 0;
    T* rec = (T*)b->records + (size_t)c.rank * K * (hd + 2);
    h->fast_lists = 0;
    if constexpr (std::is_same<T, float>::value) {
        if (b->z_r == nullptr && fast_rollout_ok(h, K)) {
            // f32 throughput path
            const uint64_t off = call_base + (uint64_t)it;
            int rc = ensure_fast_model(h);
            if (rc) return rc;
            int lists = 0;
            float* pc;
            int* pi;
            const int n_rows = n_loc + n_extra;
            const int one = (fast_sample_ok(h) && (n_extra == 0 || shift_in_sampler))
                                ? sample_rollout_lists(c.horizon, c.act_dim, h->O, c.rng_rounds, n_rows) : 0;
            // the merge finds the lists' indices behind `lists * K` costs
            split_partial_ws<float>(b->workspace, one > 0 ? one : rollout_lists(c.horizon, c.act_dim, h->O, n_rows), K, &pc, &pi);
            bool prologue = false;
            if (h->pm_pending) {
                prologue = n_extra == 0 && prologue_possible(h, n_rows);
                if (!prologue) {  // cannot ride along after all: run it now
                    rc = launch_pending_merge(h, st);
                    if (rc) return rc;
                }
                h->pm_pending = false;
            }
            if (one > 0) {
                // small populations: sample + rollout + top-K in one launch
                FastIterArgs fa;
                if (prologue) fa.m = h->pm_args;
                fa.s = fast_sample_args(h, n_loc, lo, b->mean, b->std, b->low, b->high, off, row0, actions,
                                        shift_in_sampler ? n_extra : 0, shift_src, call_base + (uint64_t)c.opt_iters);
                fa.r = fast_rollout_args(h, n_rows, n_cand, K, b->obs0, actions, b->costs, pc, pi);
                fa.r.part_k = (unsigned long long*)b->workspace;  // read by merge_single_kernel / pack_records_kernel
                {
                    ProfScope prof(h, ICEM_K_SAMPLE_ROLLOUT, (long long)n_rows * c.horizon, st);
                    launch_sample_rollout(fa, c.horizon, c.act_dim, h->O, h->model_kind, prologue, st);
                }
                ICEM_HIP_TRY(hipGetLastError());
                lists = one;
            }
            if (one == 0) {
                if (prologue) {
                    FastSampleMergeArgs sm;
                    sm.s = fast_sample_args(h, n_loc, lo, b->mean, b->std, b->low, b->high, off, row0, actions, 0, nullptr, 0);
                    sm.m = h->pm_args;
                    {
                        ProfScope prof(h, ICEM_K_SAMPLE, (long long)n_loc * c.horizon, st);
                        launch_sample_folded_merge(sm, st);
                    }
                    ICEM_HIP_TRY(hipGetLastError());
                    rc = ICEM_OK;
                } else if (fast_sample_ok(h)) {
                    rc = launch_fast_sample(h, n_loc, lo, b->mean, b->std, b->low, b->high, off, row0, actions, st,
                                            shift_in_sampler ? n_extra : 0, shift_src, call_base + (uint64_t)c.opt_iters);
                } else {
                    SampleArgs<T> a = make_sample_args<T>(h, n_loc, lo, b->mean, b->std, b->low, b->high, nullptr, nullptr,
                                                          off, 0, row0, actions);
                    rc = launch_sample<T>(h, a, st);
                }
                if (rc) return rc;
                rc = launch_fast_rollout(h, n_rows, n_cand, K, b->obs0, actions, b->costs, pc, pi, st, &lists,
                                         (unsigned long long*)b->workspace);
                if (rc) return rc;
            }
            h->fast_lists = lists;
            if (c.world > 1) {
                // this rank's K best -> records for the all-gather (same selection code as the merge)
                MergeSingleArgs pk{};
                pk.n_lists = lists;
                pk.n_keep = 0;
                pk.n_pool = n_rows;
                pk.n_global = n_global;
                pk.K = K;
                pk.h = c.horizon;
                pk.d = c.act_dim;
                pk.part_k = (const unsigned long long*)b->workspace;
                pk.actions = (const float*)actions;
                ProfScope prof(h, ICEM_K_LOCAL_PACK, lists * K, st);
                launch_pack_records(pk, n_loc, lo, (float*)rec, st);
            }
            ICEM_HIP_TRY(hipGetLastError());
            return ICEM_OK;
        }
    }
    // generic path (f64, external noise, shapes outside the fast list): one kernel per stage
    {
        SampleArgs<T> a = make_sample_args<T>(h, n_loc, lo, b->mean, b->std, b->low, b->high, b->z_r, b->z_i,
                                              call_base + (uint64_t)it, 0, row0, actions);
        int rc = launch_sample<T>(h, a, st);
        if (rc) return rc;
    }
    int rc = launch_rollout<T>(h, n_loc + n_extra, b->obs0, actions, b->costs, nullptr, st);
    if (rc) return rc;
    const int nblk = std::max(1, topk_blocks(n_cand));
    T* pc;
    int* pi;
    split_partial_ws<T>(b->workspace, nblk, K, &pc, &pi);
    {
        ProfScope prof(h, ICEM_K_TOPK_PARTIAL, n_cand, st);
        hipLaunchKernelGGL((topk_partial_kernel<T>), dim3(nblk), dim3(WG), 0, st, n_cand, K, (const T*)b->costs, pc, pi);
    }
    {
        ProfScope prof(h, ICEM_K_LOCAL_PACK, nblk * K, st);
        hipLaunchKernelGGL((local_pack_kernel<T>), dim3(1), dim3(WG), 0, st, nblk * K, K, hd, n_loc, lo, n_global,
                           (const T*)pc, (const int*)pi, (const T*)actions, rec);
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

template <typename T>
int plan_iter_merge_t(icem_handle* h, const icem_plan_buffers* b, int mpc_step, int it, hipStream_t st) {
    const icem_config& c = h->cfg;
    const int hd = h->hd, K = c.num_elites;
    const long long g = (long long)mpc_step * c.opt_iters + it;  // global iteration number
    const int cur = (int)(g & 1), nxt = cur ^ 1;
    T* el = (T*)b->elites;
    T* elc = el + (size_t)2 * K * hd;
    if constexpr (std::is_same<T, float>::value) {
        if (c.world == 1 && h->fast_lists > 0) {
            MergeSingleArgs m;
            m.n_lists = h->fast_lists;
            m.n_keep = (it > 0 && c.keep_previous_elites) ? h->n_reuse : 0;
            const int n_extra = (it == 0 && c.shift_elites && mpc_step > 0) ? h->n_reuse : 0;
            m.n_pool = h->pop[it] + n_extra;
            m.n_global = h->pop[it];
            m.K = K;
            m.h = c.horizon;
            m.d = c.act_dim;
            m.last = it == c.opt_iters - 1;
            m.alpha = (float)c.alpha;
            m.init_std = (float)c.init_std;
            m.part_k = (const unsigned long long*)b->workspace;
            m.records = nullptr;
            m.n_rec = 0;
            m.actions = (const float*)b->actions;
            m.elites_cur = (const float*)el + (size_t)cur * K * hd;
            m.elites_cost_cur = (const float*)elc + (size_t)cur * K;
            m.elites_next = (float*)el + (size_t)nxt * K * hd;
            m.elites_cost_next = (float*)elc + (size_t)nxt * K;
            m.mean = (const float*)b->mean;
            m.std = (const float*)b->std;
            m.mean_out = h->merge_mean_out ? h->merge_mean_out : (float*)b->mean;
            m.std_out = h->merge_std_out ? h->merge_std_out : (float*)b->std;
            m.low = (const float*)b->low;
            m.high = (const float*)b->high;
            m.executed = (float*)b->executed;
            m.best_cost = (float*)b->best_cost;
            m.dbg = h->dbg;
            if (h->defer_merge && !m.last) {  // rides in the next iteration's launch (icem_plan_step)
                h->pm_args = m;
                h->pm_pending = true;
                return ICEM_OK;
            }
            ProfScope prof(h, ICEM_K_MERGE_REFIT, h->fast_lists * K + m.n_keep, st);
            launch_merge_single(m, st);
            ICEM_HIP_TRY(hipGetLastError());
            return ICEM_OK;
        }
    }
    MergeArgs<T> a;
    a.n_rec = c.world * K;
    a.n_keep = (it > 0 && c.keep_previous_elites) ? h->n_reuse : 0;
    a.K = K;
    a.h = c.horizon;
    a.d = c.act_dim;
    a.n_global = h->pop[it];
    a.last = it == c.opt_iters - 1;
    a.alpha = (T)c.alpha;
    a.init_std = (T)c.init_std;
    a.records = (const T*)b->records;
    a.elites_cur = el + (size_t)cur * K * hd;
    a.elites_cost_cur = elc + (size_t)cur * K;
    a.elites_next = el + (size_t)nxt * K * hd;
    a.elites_cost_next = elc + (size_t)nxt * K;
    a.mean_in = (const T*)b->mean;
    a.std_in = (const T*)b->std;
    a.mean = h->merge_mean_out ? (T*)h->merge_mean_out : (T*)b->mean;
    a.std = h->merge_std_out ? (T*)h->merge_std_out : (T*)b->std;
    a.low = (const T*)b->low;
    a.high = (const T*)b->high;
    a.executed = (T*)b->executed;
    a.best_cost = (T*)b->best_cost;
    if constexpr (std::is_same<T, float>::value) {
        const bool fast_records = h->fast_lists > 0 && a.n_rec <= 128 && K <= 32;
        if (fast_records) {  // f32 throughput path: the selection / refit code of the single-GPU merge on the records
            MergeSingleArgs m{};
            m.n_lists = 0;
            m.n_keep = a.n_keep;
            m.n_pool = 0;
            m.n_global = a.n_global;
            m.K = K;
            m.h = a.h;
            m.d = a.d;
            m.last = 0;
            m.alpha = a.alpha;
            m.init_std = a.init_std;
            m.part_k = nullptr;
            m.records = a.records;
            m.n_rec = a.n_rec;
            m.actions = nullptr;
            m.elites_cur = a.elites_cur;
            m.elites_cost_cur = a.elites_cost_cur;
            m.elites_next = a.elites_next;
            m.elites_cost_next = a.elites_cost_next;
            m.mean = a.mean_in;
            m.std = a.std_in;
            m.mean_out = a.mean;
            m.std_out = a.std;
            m.low = a.low;
            m.high = a.high;
            m.executed = a.executed;
            m.best_cost = a.best_cost;
            m.dbg = nullptr;
            if (h->defer_merge && !a.last) {  // rides in the next iteration's launch
                h->pm_args = m;
                h->pm_pending = true;
                return ICEM_OK;
            }
            m.last = a.last;
            ProfScope prof(h, ICEM_K_MERGE_REFIT, a.n_rec + a.n_keep, st);
            launch_merge_single(m, st);
            ICEM_HIP_TRY(hipGetLastError());
            return ICEM_OK;
        }
    }
    {
        ProfScope prof(h, ICEM_K_MERGE_REFIT, a.n_rec + a.n_keep, st);
        hipLaunchKernelGGL((merge_refit_kernel<T>), dim3(1), dim3(WG), (size_t)hd * sizeof(T), st, a);
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int check_handle(const icem_handle* h) {
    if (!h) return fail(ICEM_E_INVALID, "null handle");
    return ICEM_OK;
}

}  // namespace

#define ICEM_DISPATCH(h, expr_f32, expr_f64) ((h)->cfg.dtype == ICEM_F64 ? (expr_f64) : (expr_f32))

extern "C" {

int icem_abi_version(void) { return ICEM_ABI_VERSION; }

const char* icem_last_error(void) { return g_err.c_str(); }

int icem_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int icem_noise_tables_host(int32_t horizon, double beta, double* cr_host, double* ci_host) {
    if (horizon < 2 || !cr_host || !ci_host) return fail(ICEM_E_INVALID, "bad horizon / null output");
    std::vector<double> cr, ci;
    noise_tables(horizon, beta, cr, ci);
    std::memcpy(cr_host, cr.data(), cr.size() * sizeof(double));
    std::memcpy(ci_host, ci.data(), ci.size() * sizeof(double));
    return ICEM_OK;
}

int icem_create(const icem_config* cfg, icem_handle** out) {
    if (!cfg || !out) return fail(ICEM_E_INVALID, "null argument");
    const icem_config& c = *cfg;
    if (c.num_traj < 2) return fail(ICEM_E_INVALID, "At least two trajectories needed!");  // mpc.py:30-31
    if (c.horizon < 2 || c.horizon > ICEM_MAX_HORIZON) return fail(ICEM_E_UNSUPPORTED, "horizon must be in [2, 64]");
    if (c.act_dim < 1 || c.act_dim > ICEM_MAX_ACT_DIM) return fail(ICEM_E_UNSUPPORTED, "act_dim must be in [1, 64]");
    if (c.num_elites < 1 || c.num_elites > ICEM_MAX_ELITES) return fail(ICEM_E_UNSUPPORTED, "num_elites must be in [1, 64]");
    if (c.opt_iters < 1) return fail(ICEM_E_INVALID, "opt_iters < 1");
    if (c.dtype != ICEM_F32 && c.dtype != ICEM_F64) return fail(ICEM_E_INVALID, "dtype");
    if (c.rng_rounds != 10 && c.rng_rounds != 7) return fail(ICEM_E_INVALID, "rng_rounds must be 10 or 7");
    if (c.world < 1 || c.rank < 0 || c.rank >= c.world) return fail(ICEM_E_INVALID, "rank/world");
    if (c.noise_beta != c.noise_beta) return fail(ICEM_E_INVALID, "noise_beta is NaN");  // <= 0: white branch, icem.py:77
    if (!(c.factor_decrease >= 1.0)) return fail(ICEM_E_INVALID, "factor_decrease must be >= 1");
    if (c.cost_mode < 0 || c.cost_mode > 2)
        return fail(ICEM_E_UNSUPPORTED, "Implement method to compute cost along trajectory");  // abstract_controller.py:88-91
    if (icem_device_count() < 1) return fail(ICEM_E_NO_DEVICE, "no HIP device visible");
    icem_handle* h = new icem_handle();
    h->cfg = c;
    h->F = c.horizon / 2 + 1;
    h->HMAX = c.horizon <= 32 ? 32 : 64;
    h->hd = c.horizon * c.act_dim;
    h->tsize = c.dtype == ICEM_F64 ? 8 : 4;
    h->pop = population_sizes(c);
    h->n_reuse = (int)((double)c.num_elites * c.fraction_reused);  // int(len(elites)*xi), icem.py:98,145
    // the population can GROW after iteration 0 when N < 2*elites_size (icem.py:127 floors N_i at 2*elites_size)
    h->n_local_max = 0;
    for (int n_it : h->pop) h->n_local_max = std::max(h->n_local_max, shard_chunk(n_it, c.world));
    if (const char* e = getenv("ICEM_DISABLE_FAST")) h->use_fast = !(e[0] == '1');
    // synthesis table W[t][m]: m < F real part of bin m, F <= m < h imaginary part of bin m-F+1
    // noise_beta <= 0 is the reference's white branch (np.random.randn(N, h, d), icem.py:77): draw t of a row is
    // its sample at step t, i.e. the identity table
    std::vector<double> cr, ci, W((size_t)c.horizon * h->HMAX, 0.0);
    if (c.noise_beta > 0) {
        noise_tables(c.horizon, c.noise_beta, cr, ci);
        for (int t = 0; t < c.horizon; ++t)
            for (int m = 0; m < c.horizon; ++m)
                W[(size_t)t * h->HMAX + m] = m < h->F ? cr[(size_t)m * c.horizon + t] : ci[(size_t)(m - h->F + 1) * c.horizon + t];
    } else {
        for (int t = 0; t < c.horizon; ++t) W[(size_t)t * h->HMAX + t] = 1.0;
    }
    int rc = c.dtype == ICEM_F64 ? upload<double>(&h->W_dev, W) : upload<float>(&h->W_dev, W);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return ICEM_OK;
}

int icem_destroy(icem_handle* h) {
    if (!h) return ICEM_OK;
    if (h->W_dev) (void)hipFree(h->W_dev);
    if (h->actions_alt) (void)hipFree(h->actions_alt);
    if (h->host_stage) (void)hipHostFree(h->host_stage);
    if (h->ws_alt) (void)hipFree(h->ws_alt);
    if (h->pp_stats) (void)hipFree(h->pp_stats);
    if (h->A_dev) (void)hipFree(h->A_dev);
    if (h->B_dev) (void)hipFree(h->B_dev);
    if (h->Mp_dev) (void)hipFree(h->Mp_dev);
    if (h->perm_dev) (void)hipFree(h->perm_dev);
    for (auto& sp : h->spans) {
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    for (auto e : h->free_events) (void)hipEventDestroy(e);
    delete h;
    return ICEM_OK;
}

int icem_population_sizes(const icem_handle* h, int32_t* out_host) {
    if (!h || !out_host) return fail(ICEM_E_INVALID, "null argument");
    for (size_t i = 0; i < h->pop.size(); ++i) out_host[i] = h->pop[i];
    return ICEM_OK;
}

int icem_set_model(icem_handle* h, int32_t kind, int32_t obs_dim, const double* A_host, const double* B_host) {
    if (!h || !A_host || !B_host) return fail(ICEM_E_INVALID, "null argument");
    if (kind != ICEM_MODEL_LINEAR && kind != ICEM_MODEL_TANH) return fail(ICEM_E_INVALID, "model kind");
    const int O = pick_O(obs_dim);
    if (obs_dim < 1 || O < 0) return fail(ICEM_E_UNSUPPORTED, "obs_dim must be in [1, 32] for the built-in models");
    const int d = h->cfg.act_dim;
    std::vector<double> A((size_t)O * O, 0.0), B((size_t)d * O, 0.0);
    for (int k = 0; k < obs_dim; ++k)
        for (int i = 0; i < obs_dim; ++i) A[(size_t)k * O + i] = A_host[(size_t)k * obs_dim + i];
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < obs_dim; ++i) B[(size_t)j * O + i] = B_host[(size_t)j * obs_dim + i];
    int rc = h->cfg.dtype == ICEM_F64 ? upload<double>(&h->A_dev, A) : upload<float>(&h->A_dev, A);
    if (rc) return rc;
    rc = h->cfg.dtype == ICEM_F64 ? upload<double>(&h->B_dev, B) : upload<float>(&h->B_dev, B);
    if (rc) return rc;
    h->model_kind = kind;
    h->obs_dim = obs_dim;
    h->O = O;
    h->has_model = true;
    h->A_host.assign(A_host, A_host + (size_t)obs_dim * obs_dim);
    h->B_host.assign(B_host, B_host + (size_t)d * obs_dim);
    h->fast_model_ready = false;
    return ICEM_OK;
}

int icem_set_cost(icem_handle* h, const icem_cost_spec* spec) {
    if (!h || !spec) return fail(ICEM_E_INVALID, "null argument");
    h->cost = *spec;
    h->has_cost = true;
    h->fast_model_ready = false;
    return ICEM_OK;
}

int icem_set_cost_terms(icem_handle* h, const icem_cost_terms* terms) {
    if (!h) return fail(ICEM_E_INVALID, "null argument");
    if (terms == nullptr) {
        h->has_terms = false;
        return ICEM_OK;
    }
    if (terms->n_terms < 0 || terms->n_terms > ICEM_MAX_COST_TERMS) return fail(ICEM_E_INVALID, "n_terms must be in [0, 8]");
    for (int j = 0; j < terms->n_terms; ++j) {
        const icem_cost_term& tm = terms->terms[j];
        if (tm.kind < ICEM_TERM_NORM || tm.kind > ICEM_TERM_STEP_GT) return fail(ICEM_E_INVALID, "unknown cost term kind");
        if (tm.len < 1 || tm.len > ICEM_MAX_TERM_LEN) return fail(ICEM_E_INVALID, "cost term len must be in [1, 64]");
    }
    if (terms->box_from >= 0 && terms->health_idx < 0)
        return fail(ICEM_E_INVALID, "box_from is part of the health term: health_idx must be set");
    h->terms = *terms;
    h->has_terms = terms->diff_idx >= 0 || terms->health_idx >= 0 || terms->n_terms > 0;
    return ICEM_OK;
}

int icem_trajectory_cost(icem_handle* h, int32_t n, int32_t obs_dim, const void* observations,
                         const void* next_observations, int64_t traj_stride, int64_t step_stride, const void* actions,
                         void* costs, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!h->has_cost) return fail(ICEM_E_STATE, "icem_set_cost must be called first");
    if (n < 0 || obs_dim < 1 || !observations || !actions || !costs) return fail(ICEM_E_INVALID, "null tensor / bad n or obs_dim");
    if (const char* e = cost_indices_error(h, obs_dim)) return fail(ICEM_E_INVALID, e);
    if (h->has_terms && h->terms.diff_idx >= 0 && !next_observations)
        return fail(ICEM_E_INVALID, "the difference term needs next_observations");
    if (n == 0) return ICEM_OK;
    hipStream_t st = (hipStream_t)stream;
    return ICEM_DISPATCH(h, launch_trajectory_cost<float>(h, n, obs_dim, observations, next_observations, traj_stride, step_stride, actions, costs, st),
                         launch_trajectory_cost<double>(h, n, obs_dim, observations, next_observations, traj_stride, step_stride, actions, costs, st));
}

int icem_sample_clip(icem_handle* h, int32_t n, int64_t first_index, const void* mean, const void* std,
                     const void* low, const void* high, const void* z_r, const void* z_i, uint64_t offset,
                     int32_t t_begin, int32_t row0_mean, void* actions, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || !mean || !std || !low || !high || !actions) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (h->cfg.noise_beta > 0 && (z_r == nullptr) != (z_i == nullptr))
        return fail(ICEM_E_INVALID, "z_r and z_i must both be given or both NULL");
    if (t_begin < 0 || t_begin >= h->cfg.horizon) return fail(ICEM_E_INVALID, "t_begin out of range");
    hipStream_t st = (hipStream_t)stream;
    if (z_r == nullptr && t_begin == 0 && fast_sample_ok(h))
        return launch_fast_sample(h, n, first_index, mean, std, low, high, offset, row0_mean, actions, st);
    return ICEM_DISPATCH(h,
                         launch_sample<float>(h, make_sample_args<float>(h, n, first_index, mean, std, low, high, z_r, z_i, offset, t_begin, row0_mean, actions), st),
                         launch_sample<double>(h, make_sample_args<double>(h, n, first_index, mean, std, low, high, z_r, z_i, offset, t_begin, row0_mean, actions), st));
}

int icem_sample_truncnorm(icem_handle* h, int32_t n, int64_t first_index, const void* mean, const void* std,
                          const void* lower, const void* upper, const void* u, uint64_t offset, void* actions,
                          void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || !mean || !std || !lower || !upper || !actions) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (n == 0) return ICEM_OK;
    hipStream_t st = (hipStream_t)stream;
    const icem_config& c = h->cfg;
    const int grid = (n * c.act_dim + WG - 1) / WG;
    const uint32_t sl = (uint32_t)c.seed, sh = (uint32_t)(c.seed >> 32), ol = (uint32_t)offset, oh = (uint32_t)(offset >> 32);
#define ICEM_TN(T, R)                                                                                                    \
    hipLaunchKernelGGL((sample_truncnorm_kernel<T, R>), dim3(grid), dim3(WG), 0, st, n, c.horizon, c.act_dim,            \
                       (long long)first_index, (const T*)mean, (const T*)std, (const T*)lower, (const T*)upper,          \
                       (const T*)u, sl, sh, ol, oh, (T*)actions)
    if (c.dtype == ICEM_F64) {
        if (c.rng_rounds == 7) ICEM_TN(double, 7); else ICEM_TN(double, 10);
    } else {
        if (c.rng_rounds == 7) ICEM_TN(float, 7); else ICEM_TN(float, 10);
    }
#undef ICEM_TN
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_sample_piecewise(icem_handle* h, int32_t n, int64_t call_offset, int32_t change_freq, int64_t first_block,
                          const void* low, const void* high, const void* u, void* actions, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || call_offset < 0 || change_freq < 0 || !low || !high || !actions)
        return fail(ICEM_E_INVALID, "null tensor / negative n, call_offset or change_freq");
    if (n == 0) return ICEM_OK;
    const icem_config& c = h->cfg;
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)n * c.horizon * c.act_dim;
    const int grid = (int)((total + WG - 1) / WG);
    const uint32_t sl = (uint32_t)c.seed, sh = (uint32_t)(c.seed >> 32);
#define ICEM_PW(T, R)                                                                                              \
    hipLaunchKernelGGL((sample_piecewise_kernel<T, R>), dim3(grid), dim3(WG), 0, st, total, c.act_dim,             \
                       (long long)call_offset, change_freq, (long long)first_block, (const T*)low, (const T*)high, \
                       (const T*)u, sl, sh, (T*)actions)
    if (c.dtype == ICEM_F64) {
        if (c.rng_rounds == 7) ICEM_PW(double, 7); else ICEM_PW(double, 10);
    } else {
        if (c.rng_rounds == 7) ICEM_PW(float, 7); else ICEM_PW(float, 10);
    }
#undef ICEM_PW
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_cem_bounds(icem_handle* h, int32_t like_levine, const void* mean, void* std, const void* low, const void* high,
                    void* lower, void* upper, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!mean || !std || !low || !high || !lower || !upper) return fail(ICEM_E_INVALID, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (h->hd + WG - 1) / WG;
    if (h->cfg.dtype == ICEM_F64)
        hipLaunchKernelGGL((cem_bounds_kernel<double>), dim3(grid), dim3(WG), 0, st, h->hd, h->cfg.act_dim, like_levine,
                           (const double*)mean, (double*)std, (const double*)low, (const double*)high, (double*)lower, (double*)upper);
    else
        hipLaunchKernelGGL((cem_bounds_kernel<float>), dim3(grid), dim3(WG), 0, st, h->hd, h->cfg.act_dim, like_levine,
                           (const float*)mean, (float*)std, (const float*)low, (const float*)high, (float*)lower, (float*)upper);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_philox_normals(icem_handle* h, int32_t n, int64_t first_index, uint64_t offset, void* z_r, void* z_i,
                        void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n <= 0 || !z_r || !z_i) return fail(ICEM_E_INVALID, "null tensor / n <= 0");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (n * h->cfg.act_dim + WG - 1) / WG;
    const bool r7 = h->cfg.rng_rounds == 7;
#define ICEM_PN(T, HM, R)                                                                                           \
    hipLaunchKernelGGL((philox_normals_kernel<T, HM, R>), dim3(grid), dim3(WG), 0, st,                              \
                       make_sample_args<T>(h, n, first_index, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, \
                                           offset, 0, 0, nullptr),                                                  \
                       (T*)z_r, (T*)z_i)
    if (h->cfg.dtype == ICEM_F64) {
        if (h->HMAX == 32) { if (r7) ICEM_PN(double, 32, 7); else ICEM_PN(double, 32, 10); }
        else { if (r7) ICEM_PN(double, 64, 7); else ICEM_PN(double, 64, 10); }
    } else {
        if (h->HMAX == 32) { if (r7) ICEM_PN(float, 32, 7); else ICEM_PN(float, 32, 10); }
        else { if (r7) ICEM_PN(float, 64, 7); else ICEM_PN(float, 64, 10); }
    }
#undef ICEM_PN
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_rollout_cost(icem_handle* h, int32_t n, const void* obs0, const void* actions, void* costs,
                      void* observations, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!h->has_model || !h->has_cost) return fail(ICEM_E_STATE, "icem_set_model / icem_set_cost must be called first");
    if (n < 0 || !obs0 || !actions || !costs) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (const char* e = cost_indices_error(h, h->obs_dim)) return fail(ICEM_E_INVALID, e);
    hipStream_t st = (hipStream_t)stream;
    if (observations == nullptr && n > 0 && fast_rollout_ok(h, 0))
        return launch_fast_rollout(h, n, 0, 0, obs0, actions, costs, nullptr, nullptr, st, nullptr);
    return ICEM_DISPATCH(h, launch_rollout<float>(h, n, obs0, actions, costs, observations, st),
                         launch_rollout<double>(h, n, obs0, actions, costs, observations, st));
}

int icem_cost_reduce(icem_handle* h, int32_t n, const void* step_costs, void* costs, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || !step_costs || !costs) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (n == 0) return ICEM_OK;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (n + WG - 1) / WG;
    if (h->cfg.dtype == ICEM_F64)
        hipLaunchKernelGGL((cost_reduce_kernel<double>), dim3(grid), dim3(WG), 0, st, n, h->cfg.horizon, h->cfg.cost_mode,
                           (const double*)step_costs, (double*)costs);
    else
        hipLaunchKernelGGL((cost_reduce_kernel<float>), dim3(grid), dim3(WG), 0, st, n, h->cfg.horizon, h->cfg.cost_mode,
                           (const float*)step_costs, (float*)costs);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

size_t icem_topk_workspace_bytes(const icem_handle* h, int32_t n, int32_t k) {
    if (!h || n < 1 || k < 1) return 0;
    return (size_t)topk_blocks(n) * (size_t)k * (h->tsize + sizeof(int));
}

int icem_topk_sorted(icem_handle* h, int32_t n, const void* costs, int32_t k, void* out_cost, int32_t* out_idx,
                     void* workspace, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 1 || k < 1 || k > ICEM_MAX_ELITES || !costs || !out_cost || !out_idx || !workspace)
        return fail(ICEM_E_INVALID, "bad n/k or null tensor");
    hipStream_t st = (hipStream_t)stream;
    return ICEM_DISPATCH(h, launch_topk<float>(n, k, costs, out_cost, out_idx, workspace, st),
                         launch_topk<double>(n, k, costs, out_cost, out_idx, workspace, st));
}

int icem_gather_refit(icem_handle* h, const void* actions, const int32_t* idx, int32_t k, void* mean, void* std,
                      void* elites_out, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (k < 1 || !actions || !idx || !mean || !std) return fail(ICEM_E_INVALID, "bad k or null tensor");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (h->hd + WG - 1) / WG;
    if (h->cfg.dtype == ICEM_F64)
        hipLaunchKernelGGL((gather_refit_kernel<double>), dim3(grid), dim3(WG), 0, st, h->hd, k, (double)h->cfg.alpha,
                           (const double*)actions, idx, (double*)mean, (double*)std, (double*)elites_out);
    else
        hipLaunchKernelGGL((gather_refit_kernel<float>), dim3(grid), dim3(WG), 0, st, h->hd, k, (float)h->cfg.alpha,
                           (const float*)actions, idx, (float*)mean, (float*)std, (float*)elites_out);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_shift(icem_handle* h, void* mean, void* std, const void* low, const void* high, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!mean || !std || !low || !high) return fail(ICEM_E_INVALID, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    if (h->cfg.dtype == ICEM_F64)
        hipLaunchKernelGGL((shift_kernel<double>), dim3(1), dim3(WG), (size_t)h->hd * 8, st, h->cfg.horizon, h->cfg.act_dim,
                           (double)h->cfg.init_std, (double*)mean, (double*)std, (const double*)low, (const double*)high);
    else
        hipLaunchKernelGGL((shift_kernel<float>), dim3(1), dim3(WG), (size_t)h->hd * 4, st, h->cfg.horizon, h->cfg.act_dim,
                           (float)h->cfg.init_std, (float*)mean, (float*)std, (const float*)low, (const float*)high);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_reset_distribution(icem_handle* h, void* mean, void* std, const void* low, const void* high, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!mean || !std || !low || !high) return fail(ICEM_E_INVALID, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    const int grid = (h->hd + WG - 1) / WG;
    if (h->cfg.dtype == ICEM_F64)
        hipLaunchKernelGGL((reset_kernel<double>), dim3(grid), dim3(WG), 0, st, h->cfg.horizon, h->cfg.act_dim,
                           (double)h->cfg.init_std, (double*)mean, (double*)std, (const double*)low, (const double*)high);
    else
        hipLaunchKernelGGL((reset_kernel<float>), dim3(grid), dim3(WG), 0, st, h->cfg.horizon, h->cfg.act_dim,
                           (float)h->cfg.init_std, (float*)mean, (float*)std, (const float*)low, (const float*)high);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_debug_stamps(icem_handle* h, void* dev_ptr) {
    if (check_handle(h)) return ICEM_E_INVALID;
    h->dbg = (long long*)dev_ptr;
    return ICEM_OK;
}

int icem_profile_enable(icem_handle* h, int32_t on) {
    if (check_handle(h)) return ICEM_E_INVALID;
    h->profiling = on != 0;
    return ICEM_OK;
}

int icem_profile_read(icem_handle* h, double* total_ms, int64_t* launches, int64_t* units) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!total_ms || !launches || !units) return fail(ICEM_E_INVALID, "null output");
    for (int k = 0; k < ICEM_K_COUNT; ++k) {
        total_ms[k] = 0.0;
        launches[k] = 0;
        units[k] = 0;
    }
    for (auto& sp : h->spans) {
        ICEM_HIP_TRY(hipEventSynchronize(sp.b));
        float ms = 0.f;
        ICEM_HIP_TRY(hipEventElapsedTime(&ms, sp.a, sp.b));
        total_ms[sp.kind] += ms;
        launches[sp.kind] += 1;
        units[sp.kind] += sp.units;
        h->free_events.push_back(sp.a);
        h->free_events.push_back(sp.b);
    }
    h->spans.clear();
    return ICEM_OK;
}

size_t icem_record_bytes(const icem_handle* h) { return h ? (size_t)(h->hd + 2) * h->tsize : 0; }

size_t icem_plan_buffer_bytes(const icem_handle* h, int32_t which) {
    if (!h) return 0;
    const size_t ts = h->tsize, hd = (size_t)h->hd, K = (size_t)h->cfg.num_elites;
    const size_t rows = (size_t)h->n_local_max + (size_t)h->n_reuse;
    switch (which) {
        case ICEM_BUF_MEAN:
        case ICEM_BUF_STD:
            return hd * ts;
        case ICEM_BUF_LOW:
        case ICEM_BUF_HIGH:
        case ICEM_BUF_EXECUTED:
            return (size_t)h->cfg.act_dim * ts;
        case ICEM_BUF_OBS0:
            return (size_t)std::max(1, h->obs_dim) * ts;
        case ICEM_BUF_ACTIONS:
            return rows * hd * ts;
        case ICEM_BUF_COSTS:
            return rows * ts;
        case ICEM_BUF_ELITES:
            return 2 * K * hd * ts + 2 * K * ts;
        case ICEM_BUF_RECORDS:
            return (size_t)h->cfg.world * K * (hd + 2) * ts;
        case ICEM_BUF_WORKSPACE:
            return (size_t)std::max(topk_blocks((int)rows), 1024) * K * (ts + sizeof(int));
        case ICEM_BUF_BEST_COST:
            return ts;
        default:
            return 0;
    }
}

static int check_plan(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!b) return fail(ICEM_E_INVALID, "null buffers");
    if (!h->has_model || !h->has_cost) return fail(ICEM_E_STATE, "icem_set_model / icem_set_cost must be called first");
    if (mpc_step < 0 || it < 0 || it >= h->cfg.opt_iters) return fail(ICEM_E_INVALID, "mpc_step / iteration out of range");
    if (const char* e = cost_indices_error(h, h->obs_dim)) return fail(ICEM_E_INVALID, e);
    if (!b->mean || !b->std || !b->low || !b->high || !b->obs0 || !b->actions || !b->costs || !b->elites || !b->records ||
        !b->workspace || !b->executed || !b->best_cost)
        return fail(ICEM_E_INVALID, "null plan buffer");
    if (h->cfg.noise_beta > 0 &&
        ((b->z_r == nullptr) != (b->z_i == nullptr) || (b->z_r_shift == nullptr) != (b->z_i_shift == nullptr)))
        return fail(ICEM_E_INVALID, "z_r/z_i must be given in pairs");
    return ICEM_OK;
}

// rows of this rank's shard at iteration `it`
static int local_rows(const icem_handle* h, int it) {
    const int n_global = h->pop[it];
    const int chunk = shard_chunk(n_global, h->cfg.world);
    const int lo = std::min(n_global, h->cfg.rank * chunk);
    return std::max(0, std::min(n_global - lo, chunk));
}

static int ensure_pp_stats(icem_handle* h) {
    if (!h->pp_stats) ICEM_HIP_TRY(hipMalloc((void**)&h->pp_stats, (size_t)4 * h->hd * sizeof(float)));
    return ICEM_OK;
}

// world > 1 with merge deferral: the running step's distribution is at cur_mean / cur_std
static bool deferral_active(const icem_handle* h, const icem_plan_buffers* b) {
    return h->deferral && h->cfg.world > 1 && h->cfg.dtype == ICEM_F32 && b->z_r == nullptr && h->use_fast;
}

int icem_set_merge_deferral(icem_handle* h, int32_t on) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (h->pm_pending) return fail(ICEM_E_STATE, "a deferred merge is pending: finish the MPC step first");
    h->deferral = on != 0;
    return ICEM_OK;
}

int icem_plan_iter_local(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it, void* stream) {
    int rc = check_plan(h, b, mpc_step, it);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    icem_plan_buffers bb = *b;
    if (deferral_active(h, b)) {
        if (it == 0 || !h->cur_mean) {
            h->cur_mean = (float*)b->mean;
            h->cur_std = (float*)b->std;
        }
        bb.mean = h->cur_mean;
        bb.std = h->cur_std;
    }
    return ICEM_DISPATCH(h, plan_iter_local_t<float>(h, &bb, mpc_step, it, st), plan_iter_local_t<double>(h, &bb, mpc_step, it, st));
}

int icem_plan_iter_merge(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it, void* stream) {
    int rc = check_plan(h, b, mpc_step, it);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!deferral_active(h, b) || !h->cur_mean)
        return ICEM_DISPATCH(h, plan_iter_merge_t<float>(h, b, mpc_step, it, st), plan_iter_merge_t<double>(h, b, mpc_step, it, st));
    // sharded, deferral on: a non-last merge may ride in the next icem_plan_iter_local launch (mean / std / elites in
    // the caller's buffers are then current again only after that launch; the last merge always runs here)
    rc = ensure_pp_stats(h);
    if (rc) return rc;
    const bool last = it == h->cfg.opt_iters - 1;
    const bool fold = !last && h->fast_lists > 0 && prologue_possible(h, local_rows(h, it + 1));
    icem_plan_buffers bb = *b;
    bb.mean = h->cur_mean;
    bb.std = h->cur_std;
    float* pp = h->pp_stats + (size_t)(it & 1) * 2 * h->hd;
    h->defer_merge = fold;
    if (last) {
        h->merge_mean_out = (float*)b->mean;
        h->merge_std_out = (float*)b->std;
    } else if (fold) {
        h->merge_mean_out = pp;
        h->merge_std_out = pp + h->hd;
    } else {
        h->merge_mean_out = h->merge_std_out = nullptr;
    }
    rc = plan_iter_merge_t<float>(h, &bb, mpc_step, it, st);
    h->defer_merge = false;
    h->merge_mean_out = h->merge_std_out = nullptr;
    if (rc) return rc;
    if (fold && h->pm_pending) {
        h->cur_mean = pp;
        h->cur_std = pp + h->hd;
    }
    if (last) h->cur_mean = h->cur_std = nullptr;
    return ICEM_OK;
}

int icem_plan_step(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (h->cfg.world != 1) return fail(ICEM_E_INVALID, "icem_plan_step is the world == 1 path; use iter_local/iter_merge");
    int rc = check_plan(h, b, mpc_step, 0);
    if (rc) return rc;
    const icem_config& c = h->cfg;
    const int iters = c.opt_iters;
    // f32, device noise: iteration it's merge may ride in the prologue of iteration it+1's launch.  That launch
    // reads pool / lists / distribution of iteration it while writing its own, so consecutive iterations alternate
    // between the caller's buffers and the handle's partners (the last iteration always uses the caller's).
    const bool pingpong = c.dtype == ICEM_F32 && b->z_r == nullptr && h->use_fast && iters > 1;
    if (pingpong && !h->actions_alt) {
        ICEM_HIP_TRY(hipMalloc(&h->actions_alt, icem_plan_buffer_bytes(h, ICEM_BUF_ACTIONS)));
        ICEM_HIP_TRY(hipMalloc(&h->ws_alt, icem_plan_buffer_bytes(h, ICEM_BUF_WORKSPACE)));
    }
    if (pingpong) {
        rc = ensure_pp_stats(h);
        if (rc) return rc;
    }
    float* cur_mean = (float*)b->mean;  // where the current distribution lives
    float* cur_std = (float*)b->std;
    for (int it = 0; it < iters; ++it) {
        icem_plan_buffers bb = *b;
        if (pingpong) {
            if ((iters - 1 - it) & 1) bb.actions = h->actions_alt;
            if (it & 1) bb.workspace = h->ws_alt;
            bb.mean = cur_mean;
            bb.std = cur_std;
        }
        rc = icem_plan_iter_local(h, &bb, mpc_step, it, stream);
        if (rc) return rc;
        const bool last = it == iters - 1;
        bool fold = false;
        if (pingpong && !last && h->fast_lists > 0) fold = prologue_possible(h, h->pop[it + 1]);
        h->defer_merge = fold;
        float* pp = pingpong ? h->pp_stats + (size_t)(it & 1) * 2 * h->hd : nullptr;
        if (last) {  // the final distribution goes to the caller's buffers
            h->merge_mean_out = (float*)b->mean;
            h->merge_std_out = (float*)b->std;
        } else if (fold) {
            h->merge_mean_out = pp;
            h->merge_std_out = pp + h->hd;
        } else {
            h->merge_mean_out = h->merge_std_out = nullptr;  // in place
        }
        rc = icem_plan_iter_merge(h, &bb, mpc_step, it, stream);
        h->defer_merge = false;
        h->merge_mean_out = h->merge_std_out = nullptr;
        if (rc) return rc;
        if (fold && h->pm_pending) {
            cur_mean = pp;
            cur_std = pp + h->hd;
        }
    }
    return ICEM_OK;
}

size_t icem_rssm_param_elems(void) { return rssm::TOTAL; }

int icem_rssm_rollout_cost(int32_t n, int32_t horizon, int32_t cost_mode, const void* params, const void* obs0,
                           const void* actions, void* costs, void* stream) {
    if (n < 0 || horizon < 1 || cost_mode < ICEM_COST_SUM || cost_mode > ICEM_COST_FINAL || !params || !obs0 || !actions || !costs)
        return fail(ICEM_E_INVALID, "null tensor / bad n, horizon or cost_mode");
    ICEM_HIP_TRY(launch_rssm_rollout(n, horizon, cost_mode, (const unsigned short*)params, (const float*)obs0,
                                     (const float*)actions, (float*)costs, (hipStream_t)stream));
    return ICEM_OK;
}

// MpcICem.get_action as one call for a host caller: observation in, executed action (+ its pool's best cost) out.
// The handle owns a small pinned, device-mapped block [obs | action, best cost | flag].  On the f32 fast path the first
// launch reads the observation straight from it (no H2D copy command in front of the step) and a one-thread kernel
// behind the last merge writes the result and a sequence flag into it, which the host polls (no D2H copy commands,
// no stream synchronisation wake-up).  Other configurations stage through the same block with copy commands.
int icem_get_action(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, const double* obs_host,
                    double* action_host, double* best_cost_host, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!b || !obs_host || !action_host) return fail(ICEM_E_INVALID, "null argument");
    if (!h->has_model) return fail(ICEM_E_STATE, "icem_set_model / icem_set_cost must be called first");
    hipStream_t st = (hipStream_t)stream;
    const int o = h->obs_dim, d = h->cfg.act_dim;
    const size_t ts = h->tsize;
    constexpr size_t OUT_OFF = ICEM_MAX_OBS_DIM * sizeof(double), FLAG_OFF = OUT_OFF + (ICEM_MAX_ACT_DIM + 1) * sizeof(double);
    if (!h->host_stage) {
        ICEM_HIP_TRY(hipHostMalloc(&h->host_stage, FLAG_OFF + 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->host_stage, 0, FLAG_OFF + 64);
        ICEM_HIP_TRY(hipHostGetDevicePointer(&h->host_stage_dev, h->host_stage, 0));
    }
    unsigned char* stage = (unsigned char*)h->host_stage;
    for (int k = 0; k < o; ++k) {
        if (h->cfg.dtype == ICEM_F64) ((double*)stage)[k] = obs_host[k];
        else ((float*)stage)[k] = (float)obs_host[k];
    }
    unsigned char* out = stage + OUT_OFF;
    volatile unsigned* flag = (volatile unsigned*)(stage + FLAG_OFF);
    const bool mapped = h->cfg.dtype == ICEM_F32 && h->use_fast && b->z_r == nullptr && fast_rollout_ok(h, h->cfg.num_elites) &&
                        fast_sample_ok(h);
    icem_plan_buffers bb = *b;
    if (mapped) {
        std::atomic_thread_fence(std::memory_order_release);
        bb.obs0 = h->host_stage_dev;
    } else {
        ICEM_HIP_TRY(hipMemcpyAsync(b->obs0, stage, o * ts, hipMemcpyHostToDevice, st));
    }
    const int rc = icem_plan_step(h, &bb, mpc_step, stream);
    if (rc) return rc;
    if (mapped) {
        // (publishing from inside the last merge kernel instead was tried: no faster than this one-wave launch)
        const unsigned seq = ++h->io_seq;
        unsigned char* dev = (unsigned char*)h->host_stage_dev;
        hipLaunchKernelGGL(publish_result_kernel, dim3(1), dim3(64), 0, st, (const float*)b->executed, (const float*)b->best_cost, d,
                           (float*)(dev + OUT_OFF), (unsigned*)(dev + FLAG_OFF), seq);
        ICEM_HIP_TRY(hipGetLastError());
        long long spins = 0;
        while (*flag != seq) {
            __builtin_ia32_pause();
            if (++spins > (1ll << 26)) {  // ~ a second: something is wrong on the stream -- let the runtime report it
                ICEM_HIP_TRY(hipStreamSynchronize(st));
                if (*flag != seq) return fail(ICEM_E_HIP, "result flag never arrived");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        ICEM_HIP_TRY(hipMemcpyAsync(out, b->executed, d * ts, hipMemcpyDeviceToHost, st));
        ICEM_HIP_TRY(hipMemcpyAsync(out + (size_t)d * ts, b->best_cost, ts, hipMemcpyDeviceToHost, st));
        ICEM_HIP_TRY(hipStreamSynchronize(st));
    }
    for (int j = 0; j <= d; ++j) {
        const double v = h->cfg.dtype == ICEM_F64 ? ((const double*)out)[j] : (double)((const float*)out)[j];
        if (j < d) action_host[j] = v;
        else if (best_cost_host) *best_cost_host = v;
    }
    return ICEM_OK;
}

}  // extern "C"
