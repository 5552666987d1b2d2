// generic_kernels.hip -- the generic gfx950 kernels of the iCEM inner planning loop (f32 / f64, any h <= 64, d <= 64,
// o <= 32, external or device white noise) and their launchers (gk_*, host_common.h).  They serve the strict-parity
// f64 mode, shapes outside the compiled matrix-pipe list, and the stand-alone operators of the ABI.
//
// Data layout in HBM (all C-contiguous, T = float or double):
//   actions [n, h, d]   the reference's `action_sequences` (icem/controllers/icem.py:73-79)
//   costs   [n]
//   mean/std [h, d], low/high [d]
//   W [h, HMAX]         colored-noise synthesis table, row t holds the h coefficients that turn the
//                       h white draws of one (trajectory, action-dim) row into sample t (zero padded)
//   records [world*K, 2+h*d]   {cost, gidx, actions[h*d]} -- what the ranks exchange
//
// Kernels (one section each): sample_clip (K1), rollout_cost (K2), block top-k (K3),
// local_pack / merge_refit (K3+K4 of the fused step), small epilogue kernels.
#include "host_common.h"
#include "cost_terms_dev.h"
#include "exchange_dev.h"
#include "philox.h"
#include "refit.h"

namespace icem {


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

// The same sampler with FOUR lanes per (trajectory, dim) row (WG / 4 rows per workgroup): in float64 a row is one thread's chain of
// HMAX / 2 libm-grade Box-Muller transforms and h x HMAX dependent fused multiply-adds -- 19 us per launch at N = 4096 with the
// chip all but empty.  Lane q of a row's quad runs the row's generator like the others (integer work: cheap), transforms only
// the pairs p = q, q + 4, .. of its words (or loads only those entries of the caller's z), multiplies them into partial sums
// over ITS table columns (W staged in LDS) and the quad adds the four partial sums (two DPP swaps).  Not the thread form's
// summation order: results agree with it to rounding (a few 1e-16 relative), inside the strict-parity bar of 1e-10.
template <int CTRL>
__device__ __forceinline__ float quad_swap(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double quad_swap(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}

template <typename T, int HMAX, int ROUNDS>
__global__ __launch_bounds__(WG) void sample_clip_quad_kernel(SampleArgs<T> a) {
    constexpr int PAIRS = HMAX / 8;   // Box-Muller pairs per lane
    typedef T T2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);              // [tpw][h * d]
    const int hd = a.h * a.d;
    T* Wl = tile + (((size_t)a.tpw * hd + 1) & ~(size_t)1);   // [h][HMAX]
    const int tid = threadIdx.x;
    const int q = tid & 3, rowi = tid >> 2;
    const int n_base = blockIdx.x * a.tpw;
    const int n_here = min(a.tpw, a.n - n_base);
    const int rows = n_here * a.d;
    for (int e = tid; e < a.h * HMAX; e += WG) Wl[e] = a.W[e];
    const bool on = rowi < rows;
    const int nl = on ? rowi / a.d : 0;
    const int j = on ? rowi - nl * a.d : 0;
    T g[2 * PAIRS];   // entries m = 8 i + 2 q, 8 i + 2 q + 1 of the row's white draws
    {
        const int row_local = n_base + nl;
        if (a.zr != nullptr && a.white) {
#pragma unroll
            for (int i = 0; i < PAIRS; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int m = 8 * i + 2 * q + u;
                    g[2 * i + u] = m < a.h ? a.zr[((size_t)row_local * a.h + m) * a.d + j] : (T)0;
                }
        } else if (a.zr != nullptr) {
            const size_t base = ((size_t)row_local * a.d + j) * a.F;
#pragma unroll
            for (int i = 0; i < PAIRS; ++i)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int m = 8 * i + 2 * q + u;
                    T v = (T)0;
                    if (m < a.F)
                        v = a.zr[base + m];
                    else if (m < a.h)
                        v = a.zi[base + (m - a.F + 1)];
                    g[2 * i + u] = v;
                }
        } else {
            Xoshiro128pp rng = row_stream<ROUNDS>((uint32_t)(a.first_index + row_local), (uint32_t)j, a.off_lo, a.off_hi, a.seed_lo, a.seed_hi);
            // every lane of the quad walks the row's whole word stream (integer work) and KEEPS the words of its own pairs
            // (selects, no branch); the transforms -- the float64 work -- then run on all lanes at once, a quarter each
            uint32_t wa[PAIRS], wb[PAIRS];
#pragma unroll
            for (int p = 0; p < HMAX / 2; ++p) {
                const uint32_t xa = rng.next();
                const uint32_t xb = rng.next();
                if ((p & 3) == 0) {   // (p is a constant of the unrolled loop) the group's first pair: lane 0's, a placeholder for the others
                    wa[p >> 2] = xa;
                    wb[p >> 2] = xb;
                } else {
                    const bool mine = (p & 3) == q;
                    wa[p >> 2] = mine ? xa : wa[p >> 2];
                    wb[p >> 2] = mine ? xb : wb[p >> 2];
                }
            }
#pragma unroll
            for (int i = 0; i < PAIRS; ++i) {
                if (2 * (4 * i) < a.h)   // (wave-uniform up to the quad's offset: pairs past the horizon are zero)
                    box_muller(wa[i], wb[i], g[2 * i], g[2 * i + 1]);
                if (!(2 * (4 * i + q) < a.h)) g[2 * i] = g[2 * i + 1] = (T)0;
            }
        }
    }
    __syncthreads();
    const T lo = a.low[j], hi = a.high[j];
    for (int t = a.t_begin; t < a.h; ++t) {
        const T* __restrict__ w = Wl + (size_t)t * HMAX + 2 * q;
        T acc = (T)0;
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const T2 wv = *reinterpret_cast<const T2*>(w + 8 * i);
            acc = fmad(g[2 * i], wv[0], acc);
            acc = fmad(g[2 * i + 1], wv[1], acc);
        }
        acc = acc + quad_swap<0xB1>(acc);   // quad_perm [1, 0, 3, 2]
        acc = acc + quad_swap<0x4E>(acc);   // quad_perm [2, 3, 0, 1]
        if (on && (t & 3) == q) {
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
        // the quantile lies in [lower, upper] by definition; in f32 a uniform that rounds to 1 or an interval deep in
        // one tail (pa == pb) would otherwise come back as +-inf
        T z = std_normal_icdf(fmad(uu, pb - pa, pa));
        const T lo_z = lower[t * d + j], hi_z = upper[t * d + j];
        z = z < lo_z ? lo_z : z;
        z = z > hi_z ? hi_z : z;
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
    int ch;  // (rows kernel) steps of actions staged in LDS at a time
    long long* dbg;  // icem_debug_stamps: phase stamps [24..29] of workgroup 0 (tools/dbg/f64_stamps.py), else null
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

// The same rollout with a trajectory spread over LPT lanes (lane = observation column) instead of one thread per trajectory:
// in float64 a thread's horizon is 30 x (O + d) x O dependent 8-cycle FMAs -- 214 us per launch at N = 4096, three quarters
// of a strict-parity MPC step -- while 250 of the chip's 256 CUs hold one wave each.  Here lane c of a trajectory's row keeps
// column c of A in registers and accumulates output c over the SAME chain (k = 0 .. O - 1, then the actions, fused
// multiply-adds in that order: bit-identical to the thread form), the state goes from step to step through two LDS rows per
// trajectory (written by its lanes, read back as broadcasts by the same wave: no workgroup barrier in the loop), B sits in
// LDS, a chunk of the row's actions is staged in LDS.  Costs (icem_cost_spec's form; a term list keeps the thread form) are
// computed redundantly by every lane of the row (same instructions); lane 0 stores.  ICEM_GK_ROLLOUT=thread brings the thread
// form back (A/B, tests).
template <typename T, int O, int KIND>
__global__ __launch_bounds__(WG) void rollout_cost_rows_kernel(RolloutArgs<T> a) {
    constexpr int LPT = O <= 16 ? 16 : 32;   // lanes per trajectory
    constexpr int TPW = WG / LPT;            // trajectories per workgroup
    constexpr int OS = (O + 1) & ~1;         // LDS row stride: pairs of entries are read as one 2 x T vector
    constexpr int BREG = 8;                  // action dims whose row of B stays in registers
    typedef T T2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);  // [2][TPW][OS]: the state before / behind the running step
    T* Bs = xs + 2 * TPW * OS;               // [d][OS]
    const int ds = a.d <= BREG ? BREG : ((a.d + 1) & ~1);   // a step's actions in LDS, zero padded (to BREG: read whole, unconditionally)
    T* As = Bs + a.d * OS;                   // [TPW][ch * ds]: the next `ch` steps' actions of every trajectory of the workgroup
    const int tid = threadIdx.x, col = tid % LPT, tl = tid / LPT;
    const bool live_col = col < O;
    const int cc = live_col ? col : 0;
    const int n = blockIdx.x * TPW + tl;
    const bool live = n < a.n;
    long long* const dbg = (blockIdx.x == 0 && tid == 0) ? a.dbg : nullptr;
    if (dbg) dbg[24] = wall_clock64();
    for (int e = tid; e < a.d * O; e += WG) Bs[(e / O) * OS + e % O] = a.B[e];
    T Acol[O];
#pragma unroll
    for (int k = 0; k < O; ++k) Acol[k] = live_col ? a.A[k * O + cc] : (T)0;
    T Bcol[BREG];
#pragma unroll
    for (int j = 0; j < BREG; ++j) Bcol[j] = (live_col && j < a.d) ? a.B[j * O + cc] : (T)0;
    if (tid < TPW * OS) xs[tid] = (T)0;
    __syncthreads();
    if (live_col) xs[tl * OS + col] = col < a.o ? a.obs0[col] : (T)0;
    if (O < OS && col == 0) xs[TPW * OS + tl * OS + O] = (T)0;   // (the pad entry of the second buffer: read, never written)
    __syncthreads();
    if (dbg) dbg[25] = wall_clock64();
    const T* __restrict__ act_g = a.actions + (size_t)(live ? n : a.n - 1) * a.h * a.d;
    const int lane = tid & 63;
    const unsigned long long row_mask = (LPT == 64 ? ~0ull : ((1ull << LPT) - 1ull)) << ((lane / LPT) * LPT);
    const int ch = a.ch;                     // steps per staged chunk (host: the whole horizon where it fits)
    T* my_acts = As + (size_t)tl * ch * ds;
    const bool b_regs = a.d <= BREG;
    // (the cost's scalars as locals: the argument block's term list must not be live across the loop -- with it in reach the
    //  step restored 600 spilled scalar registers through v_readlane, 1.1 us per step whatever the arithmetic.  Costs with a
    //  term list -- cs.ext -- keep the thread form: launch_rollout_k)
    const T flip_th = a.cs.flip_th, flip_pen = a.cs.flip_pen, ctrl_w = a.cs.ctrl_w, lin_w = a.cs.lin_w;
    const int lin_idx = a.cs.lin_idx, flip_idx = a.cs.flip_idx, cost_mode = a.cost_mode, hh = a.h, dd = a.d, oo = a.o;
    T* const obs_out = a.observations;
    T acc = (T)0;
    for (int t = 0; t < hh; ++t) {
        if (t % ch == 0) {
            // the row's lanes fetch their trajectory's next chunk (one coalesced span: a step's actions straight from HBM were
            // a cold miss every other step -- 1.5 us per step, 44 us per launch); same wave writes and reads: no workgroup barrier
            const int cnt = (hh - t < ch ? hh - t : ch) * dd;
            // (eight loads in flight per lane: one at a time, each a cold miss, was 0.6 us per step of the horizon)
            for (int e0 = col; e0 < cnt; e0 += 8 * LPT) {
                T v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = act_g[t * dd + (e0 + u * LPT < cnt ? e0 + u * LPT : 0)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * LPT;
                    if (e < cnt) my_acts[(e / dd) * ds + e % dd] = v[u];
                }
            }
            for (int e = col; e < ch * (ds - dd); e += LPT) my_acts[(e / (ds - dd)) * ds + dd + e % (ds - dd)] = (T)0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (dbg && t < 3) dbg[26 + t] = wall_clock64();
        const T* act = my_acts + (t % ch) * ds;
        const T* pre = xs + ((t & 1) * TPW + tl) * OS;
        T* post = xs + (((t + 1) & 1) * TPW + tl) * OS;
        // the whole state and the step's actions in a few wide broadcast reads, THEN the chain (the LDS pipe of a CU that
        // holds eight such waves was the bound with one read per entry: 0.4 us of every step)
        T xk[OS], av[BREG];
#pragma unroll
        for (int k = 0; k < OS; k += 2) {
            const T2 v = *reinterpret_cast<const T2*>(pre + k);
            xk[k] = v[0];
            xk[k + 1] = v[1];
        }
        if (b_regs) {   // the step's actions ride the same wait (the row is padded to BREG entries)
#pragma unroll
            for (int j = 0; j < BREG; j += 2) {
                const T2 v = *reinterpret_cast<const T2*>(act + j);
                av[j] = v[0];
                av[j + 1] = v[1];
            }
        }
        T nx = (T)0;
#pragma unroll
        for (int k = 0; k < O; ++k) nx = fmad(xk[k], Acol[k], nx);
        T ctrl = (T)0;
        if (b_regs) {
#pragma unroll
            for (int j = 0; j < BREG; ++j) {
                if (j < dd) {   // (wave-uniform; the padding entries take no part: an exact zero keeps its sign)
                    ctrl = fmad(av[j], av[j], ctrl);
                    nx = fmad(av[j], Bcol[j], nx);
                }
            }
        } else {
            for (int j = 0; j < dd; ++j) {
                const T aj = act[j];
                ctrl = fmad(aj, aj, ctrl);
                nx = fmad(aj, Bs[j * OS + cc], nx);
            }
        }
        const T pv = (KIND == ICEM_MODEL_TANH) ? act_tanh(nx) : nx;
        if (live_col) post[col] = pv;
        // (two more LDS reads, not a select chain over the registers: 34 compare masks do not fit the scalar registers)
        const T lin = (lin_idx >= 0 && lin_idx < O) ? pre[lin_idx] : (T)0;
        const T ang = (flip_idx >= 0 && flip_idx < O) ? pre[flip_idx] : (T)0;
        T c = (T)0;
        if (flip_idx >= 0) {
            c += (ang > flip_th) ? flip_pen : (T)0;
            c += (ang < -flip_th) ? flip_pen : (T)0;
        }
        c += ctrl_w * ctrl;
        if (lin_w != (T)0) c += lin_w * lin;
        if (t == 0 || cost_mode == ICEM_COST_FINAL)
            acc = c;
        else if (cost_mode == ICEM_COST_SUM)
            acc += c;
        else
            acc = (c < acc || c != c) ? c : acc;  // np.amin: a NaN step cost makes the trajectory's cost NaN
        if (obs_out != nullptr && live && live_col && col < oo)
            obs_out[((size_t)n * hh + t) * oo + col] = pre[col];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (live && col == 0) a.costs[n] = acc;
    if (dbg) dbg[29] = wall_clock64();
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
    // an index list padded by icem_topk_sorted (k > n: entries INT_MAX) repeats its best row instead of reading out of bounds
    auto row = [&](int r) -> size_t {
        const int i = idx[r];
        return (size_t)((i < 0 || i == INT_MAX) ? idx[0] : i);
    };
    for (int e = blockIdx.x * WG + threadIdx.x; e < hd; e += gridDim.x * WG) {
        if (elites_out != nullptr)
            for (int r = 0; r < K; ++r) elites_out[(size_t)r * hd + e] = actions[row(r) * hd + e];
        T nm, ns;
        refit_element<T>(K, alpha, mean[e], std[e], [&](int r) { return actions[row(r) * hd + e]; }, nm, ns);
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
    XchgWait xw;  // in-library exchange: wait for the ranks' records first (flags == nullptr: they are in place)
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
    if (a.xw.flags != nullptr) {  // in-library exchange: the peers' records of this iteration have landed
        if (threadIdx.x < 64) xchg_wait(a.xw, threadIdx.x);
        __syncthreads();
    }
    block_select_sorted<T>(
        cnt, a.K,
        [&](int e) { return e < a.n_rec ? nan_to_inf(a.records[(size_t)e * rs]) : a.elites_cost_cur[e - a.n_rec]; },
        [&](int e) { return e < a.n_rec ? rec_gidx(a.records + (size_t)e * rs) : a.n_global + (e - a.n_rec); }, sel_c,
        sel_i, sel_e, red_c, red_i, red_e);
    __syncthreads();
    auto src_row = [&](int r) -> const T* {
        int e = sel_e[r];
        if (e < 0) e = sel_e[0];  // fewer than K live candidates (cannot happen for K <= N / 2): repeat the best
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


// ---------------------------------------------------------------------------------------------
// K3 + K4 of a single-GPU step in ONE launch: threshold selection + gather + refit
// ---------------------------------------------------------------------------------------------
// topk_partial -> local_pack -> merge_refit are three launches of K rounds of a workgroup-wide minimum each (two barriers
// per round): 46 us per iteration in float64 for work that is a few hundred compares.  One workgroup, the selection by
// THRESHOLD (as the f32 merges do): every thread's best key, the K-th smallest of a wave's 64 bests (K rounds of a wave
// minimum over order-preserving 64-bit images of the costs) is an upper bound of the K-th smallest key overall, the tightest
// of the waves' bounds is T; the keys at or below T (K .. a few dozen) are collected and PLACED by counting -- a key's place
// is the number of smaller keys (cost, then global index: np.argsort's order on distinct indices) -- and the K rows are
// gathered and refitted exactly as merge_refit_kernel does (same refit_element, same epilogue).  Same elites, same bits.
template <typename T>
struct SelectArgs {
    int n_cand;   // pool rows with a cost: the sampled rows, then (iteration 0) the shifted elites
    int n_loc;    // ... of which sampled rows (global index = pool row); the others: n_global + (row - n_loc)
    int cap;      // candidate capacity (host: SELECT_CAP)
    const T* costs;
    const T* actions;
    long long* dbg;   // icem_debug_stamps: phase stamps [16..22] (tools/dbg/f64_stamps.py), else null
    MergeArgs<T> m;   // records / n_rec / xw unused: the rows come from the pool
};

constexpr int SELECT_NT = 256;
constexpr int SELECT_CAP = 3072;

__device__ __forceinline__ unsigned long long order_image(double c) {   // monotone: c1 < c2  <=>  image(c1) < image(c2); -0 == +0
    c = c == 0.0 ? 0.0 : c;
    const unsigned long long b = (unsigned long long)__double_as_longlong(c);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ unsigned long long order_image(float c) { return order_image((double)c); }   // (exact widening)

__device__ __forceinline__ unsigned long long wave_min_u64_shfl(unsigned long long x) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long o = __shfl_xor(x, s, 64);
        x = o < x ? o : x;
    }
    return x;
}

template <typename T>
__global__ __launch_bounds__(SELECT_NT) void select_refit_kernel(SelectArgs<T> s) {
    constexpr int NW = SELECT_NT / 64;
    __shared__ unsigned long long wave_T[NW];
    __shared__ unsigned long long wave_best[NW][64];
    __shared__ T cand_c[SELECT_CAP];
    __shared__ int cand_i[SELECT_CAP];
    __shared__ int cand_e[SELECT_CAP];
    __shared__ int n_c;
    __shared__ T sel_c[ICEM_MAX_ELITES];
    __shared__ int sel_i[ICEM_MAX_ELITES];
    __shared__ int sel_e[ICEM_MAX_ELITES];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* new_mean = reinterpret_cast<T*>(smem_raw);  // [hd]
    const MergeArgs<T>& a = s.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = s.n_cand + a.n_keep;
    auto cost_of = [&](int e) -> T { return e < s.n_cand ? nan_to_inf(s.costs[e]) : a.elites_cost_cur[e - s.n_cand]; };
    auto gidx_of = [&](int e) -> int { return e < s.n_loc ? e : a.n_global + (e < s.n_cand ? e - s.n_loc : e - s.n_cand); };
    if (s.dbg && tid == 0) s.dbg[16] = wall_clock64();
    if (tid == 0) n_c = 0;
    if (tid < ICEM_MAX_ELITES) {
        sel_c[tid] = inf_v<T>();
        sel_i[tid] = INT_MAX;
        sel_e[tid] = -1;
    }
    // every thread's best cost (image), then the K-th smallest of the wave's 64
    // (loads in batches of eight, all requested before the first is used: one thread's keys are 2 KB apart -- a cold miss
    //  each, and a loop of dependent misses was 16 us of this kernel)
    unsigned long long best = ~0ull;
    constexpr int KEEP = 16;             // a thread's first keys stay in registers for the second pass (all of them up to 4096 keys)
    T kept[KEEP];
#pragma unroll
    for (int b = 0; b < KEEP / 8; ++b) {
        const int e0 = tid + b * 8 * SELECT_NT;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * SELECT_NT;
            kept[b * 8 + u] = cost_of(e < total ? e : 0);
        }
    }
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
        const unsigned long long v = tid + q * SELECT_NT < total ? order_image(kept[q]) : ~0ull;
        best = v < best ? v : best;
    }
    for (int e0 = tid + KEEP * SELECT_NT; e0 < total; e0 += 8 * SELECT_NT) {
        T cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * SELECT_NT;
            cv[u] = cost_of(e < total ? e : tid);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned long long v = e0 + u * SELECT_NT < total ? order_image(cv[u]) : ~0ull;
            best = v < best ? v : best;
        }
    }
    if (s.dbg && tid == 0) s.dbg[17] = wall_clock64();
    // ... by counting: a lane's rank among the wave's 64 (value, then lane) is the number of smaller ones -- 64 broadcast
    // reads, no dependent cross-lane chain (K rounds of a shuffled 64-bit minimum cost 5 us)
    wave_best[wave][lane] = best;
    if (lane == 0) wave_T[wave] = ~0ull;   // fewer than K threads with a key: everything is a candidate
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < 64; ++j) {
        const unsigned long long o = wave_best[wave][j];
        rank += (o < best || (o == best && j < lane)) ? 1 : 0;
    }
    if (rank == a.K - 1 && best != ~0ull) wave_T[wave] = best;
    __syncthreads();
    if (s.dbg && tid == 0) s.dbg[18] = wall_clock64();
    unsigned long long Tt = wave_T[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) Tt = wave_T[w] < Tt ? wave_T[w] : Tt;
    // the keys at or below the threshold (costs that tie with it all survive)
#pragma unroll
    for (int q = 0; q < KEEP; ++q) {
        const int e = tid + q * SELECT_NT;
        if (e < total && order_image(kept[q]) <= Tt) {
            const int pos = atomicAdd(&n_c, 1);
            if (pos < s.cap) {
                cand_c[pos] = kept[q];
                cand_i[pos] = gidx_of(e);
                cand_e[pos] = e;
            }
        }
    }
    for (int e0 = tid + KEEP * SELECT_NT; e0 < total; e0 += 8 * SELECT_NT) {
        T cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * SELECT_NT;
            cv[u] = cost_of(e < total ? e : tid);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u * SELECT_NT;
            if (e < total && order_image(cv[u]) <= Tt) {
                const int pos = atomicAdd(&n_c, 1);
                if (pos < s.cap) {
                    cand_c[pos] = cv[u];
                    cand_i[pos] = gidx_of(e);
                    cand_e[pos] = e;
                }
            }
        }
    }
    __syncthreads();
    if (s.dbg && tid == 0) s.dbg[19] = wall_clock64();
    // More keys at or below the threshold than the candidate array holds: costs that TIE with it in bulk (a collapsed
    // distribution -- every trajectory the same --, a population of NaNs).  Which of them landed in the array is a race, and
    // np.argsort's order among equal costs is by index: take the three-launch path's selection instead (K rounds of a
    // workgroup-wide minimum over all keys: deterministic, slow, rare).
    if (n_c > s.cap) {
        __shared__ T red_c[SELECT_NT / 64];
        __shared__ int red_i[SELECT_NT / 64];
        __shared__ int red_e[SELECT_NT / 64];
        static_assert(SELECT_NT == WG, "block_select_sorted strides by WG");
        block_select_sorted<T>(total, a.K, cost_of, gidx_of, sel_c, sel_i, sel_e, red_c, red_i, red_e);
        __syncthreads();
    }
    const int nc = n_c <= s.cap ? n_c : 0;   // (overflow: the selection above stands, nothing is placed)
    for (int i = tid; i < nc; i += SELECT_NT) {
        const T c = cand_c[i];
        const int g = cand_i[i];
        int place = 0;
#pragma unroll 8
        for (int j = 0; j < nc; ++j) place += key_less(cand_c[j], cand_i[j], c, g) ? 1 : 0;
        if (place < a.K) {
            sel_c[place] = c;
            sel_i[place] = g;
            sel_e[place] = cand_e[i];
        }
    }
    __syncthreads();
    if (s.dbg && tid == 0) {
        s.dbg[20] = wall_clock64();
        s.dbg[23] = nc;
    }
    const int hd = a.h * a.d;
    auto src_row = [&](int r) -> const T* {
        int e = sel_e[r];
        if (e < 0) e = sel_e[0];  // fewer than K live candidates (cannot happen for K <= N / 2): repeat the best
        return e < s.n_cand ? s.actions + (size_t)e * hd : a.elites_cur + (size_t)(e - s.n_cand) * hd;
    };
    constexpr int KR = 16;   // elite rows gathered into registers at once (K <= KR: every load requested before the first use)
    for (int e = tid; e < hd; e += SELECT_NT) {
        T nm, ns;
        if (a.K <= KR) {
            T xr[KR];
#pragma unroll
            for (int r = 0; r < KR; ++r) xr[r] = src_row(r < a.K ? r : 0)[e];
            const T om = a.mean_in[e], os = a.std_in[e];
#pragma unroll
            for (int r = 0; r < KR; ++r)
                if (r < a.K) a.elites_next[(size_t)r * hd + e] = xr[r];
            refit_element_regs<T, KR>(a.K, a.alpha, om, os, xr, nm, ns);
        } else {
            for (int r = 0; r < a.K; ++r) a.elites_next[(size_t)r * hd + e] = src_row(r)[e];
            refit_element<T>(a.K, a.alpha, a.mean_in[e], a.std_in[e], [&](int r) { return src_row(r)[e]; }, nm, ns);
        }
        if (!a.last) {
            a.mean[e] = nm;
            a.std[e] = ns;
        } else {
            new_mean[e] = nm;
        }
    }
    if (tid < a.K) a.elites_cost_next[tid] = sel_c[tid];
    if (s.dbg && tid == 0) s.dbg[21] = wall_clock64();
    if (a.last) {
        __syncthreads();
        for (int e = tid; e < hd; e += SELECT_NT) {
            const int j = e % a.d;
            a.mean[e] = (e + a.d < hd) ? new_mean[e + a.d] : new_mean[e];
            a.std[e] = (a.high[j] - a.low[j]) / (T)2 * a.init_std;
        }
        if (tid < a.d) a.executed[tid] = src_row(0)[tid];
        if (tid == 0) a.best_cost[0] = sel_c[0];
    }
}


// =============================================================================================
// launchers
// =============================================================================================

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
int launch_sample(const icem_handle* h, const SampleArgs<T>& a_in, hipStream_t st) {
    if (a_in.n <= 0) return ICEM_OK;
    SampleArgs<T> a = a_in;
    // four lanes per row where the population leaves the chip mostly empty (the one-thread-per-row form's 256 rows per
    // workgroup are then a few long chains per CU); option gk_sample = 0: never (A/B, tests; read per call)
    // (decided from the handle's GLOBAL population, never from a call's or a shard's row count: one summation order per handle)
    const bool quad = opt_i(OPT_GK_SAMPLE) != 0 && a.d <= WG / 4 && (long long)h->cfg.num_traj * a.d <= 262144;
    if (quad) a.tpw = std::max(1, (WG / 4) / a.d);
    const int grid = (a.n + a.tpw - 1) / a.tpw;
    size_t lds = (size_t)a.tpw * a.h * a.d * sizeof(T);
    ProfScope prof(h, ICEM_K_SAMPLE, (long long)a.n * (a.h - a.t_begin), st);
    const bool r7 = h->cfg.rng_rounds == 7;
    if (quad) {
        lds = ((((size_t)a.tpw * a.h * a.d + 1) & ~(size_t)1) + (size_t)a.h * h->HMAX) * sizeof(T);
        if (h->HMAX == 32) {
            if (r7)
                hipLaunchKernelGGL((sample_clip_quad_kernel<T, 32, 7>), dim3(grid), dim3(WG), lds, st, a);
            else
                hipLaunchKernelGGL((sample_clip_quad_kernel<T, 32, 10>), dim3(grid), dim3(WG), lds, st, a);
        } else {
            if (r7)
                hipLaunchKernelGGL((sample_clip_quad_kernel<T, 64, 7>), dim3(grid), dim3(WG), lds, st, a);
            else
                hipLaunchKernelGGL((sample_clip_quad_kernel<T, 64, 10>), dim3(grid), dim3(WG), lds, st, a);
        }
        ICEM_HIP_TRY(hipGetLastError());
        return ICEM_OK;
    }
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
    const bool thread_form = opt_i(OPT_GK_ROLLOUT_THREAD) != 0;   // read per call: the path-equivalence test flips it between planners
    if (!thread_form && !a.cs.ext) {   // a trajectory's row of lanes (rollout_cost_rows_kernel); term lists: the thread form
        switch (h->O) {
#define ICEM_CASE(OV)                                                                                                          \
    case OV: {                                                                                                                 \
        constexpr int TPW = WG / (OV <= 16 ? 16 : 32);                                                                         \
        RolloutArgs<T> ar = a;                                                                                                 \
        constexpr int OSV = (OV + 1) & ~1;                                                                                     \
        const int dsv = a.d <= 8 ? 8 : ((a.d + 1) & ~1);                                                                       \
        ar.ch = std::max(1, std::min(a.h, (int)(32768 / ((size_t)TPW * dsv * sizeof(T)))));   /* at most 32 KB of actions */   \
        const size_t lds = ((size_t)2 * TPW * OSV + (size_t)a.d * OSV + (size_t)TPW * ar.ch * dsv) * sizeof(T);                \
        hipLaunchKernelGGL((rollout_cost_rows_kernel<T, OV, KIND>), dim3((a.n + TPW - 1) / TPW), dim3(WG), lds, st, ar);      \
        break;                                                                                                                 \
    }
            ICEM_CASE(8)
            ICEM_CASE(16)
            ICEM_CASE(17)
            ICEM_CASE(18)
            ICEM_CASE(24)
            ICEM_CASE(32)
#undef ICEM_CASE
            default:
                return fail(ICEM_E_UNSUPPORTED, "the generic rollout is compiled for padded observation widths 8, 16, 17, 18, 24, 32 only");
        }
        ICEM_HIP_TRY(hipGetLastError());
        return ICEM_OK;
    }
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
            return fail(ICEM_E_UNSUPPORTED, "the generic rollout is compiled for padded observation widths 8, 16, 17, 18, 24, 32 only");
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

void fill_cost_args_f32(const icem_handle* h, CostArgs<float>& cs) { fill_cost_args<float>(h, cs); }

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
    a.ch = 0;
    a.dbg = h->dbg;
    return h->model_kind == ICEM_MODEL_TANH ? launch_rollout_k<T, ICEM_MODEL_TANH>(h, a, st)
                                            : launch_rollout_k<T, ICEM_MODEL_LINEAR>(h, a, st);
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

// ---- dtype-erased entry points (host_common.h) ---------------------------------------------------------------------

int gk_sample(const icem_handle* h, int n, long long first_index, const void* mean, const void* std, const void* low,
              const void* high, const void* zr, const void* zi, uint64_t offset, int t_begin, int row0_mean, void* out,
              hipStream_t st) {
    return ICEM_DISPATCH(h,
                         launch_sample<float>(h, make_sample_args<float>(h, n, first_index, mean, std, low, high, zr, zi, offset, t_begin, row0_mean, out), st),
                         launch_sample<double>(h, make_sample_args<double>(h, n, first_index, mean, std, low, high, zr, zi, offset, t_begin, row0_mean, out), st));
}

int gk_sample_truncnorm(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                        const void* lower, const void* upper, const void* u, uint64_t offset, void* actions, hipStream_t st) {
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

int gk_sample_piecewise(const icem_handle* h, int n, long long call_offset, int change_freq, long long first_block,
                        const void* low, const void* high, const void* u, void* actions, hipStream_t st) {
    const icem_config& c = h->cfg;
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

int gk_cem_bounds(const icem_handle* h, int like_levine, const void* mean, void* std, const void* low, const void* high,
                  void* lower, void* upper, hipStream_t st) {
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

int gk_philox_normals(const icem_handle* h, int n, long long first_index, uint64_t offset, void* z_r, void* z_i,
                      hipStream_t st) {
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

int gk_rollout(const icem_handle* h, int n, const void* obs0, const void* actions, void* costs, void* observations,
               hipStream_t st) {
    return ICEM_DISPATCH(h, launch_rollout<float>(h, n, obs0, actions, costs, observations, st),
                         launch_rollout<double>(h, n, obs0, actions, costs, observations, st));
}

int gk_trajectory_cost(const icem_handle* h, int n, int o, const void* obs, const void* nxt, long long ts, long long ss,
                       const void* actions, void* costs, hipStream_t st) {
    return ICEM_DISPATCH(h, launch_trajectory_cost<float>(h, n, o, obs, nxt, ts, ss, actions, costs, st),
                         launch_trajectory_cost<double>(h, n, o, obs, nxt, ts, ss, actions, costs, st));
}

int gk_cost_reduce(const icem_handle* h, int n, const void* step_costs, void* costs, hipStream_t st) {
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

int gk_topk(const icem_handle* h, int n, int K, const void* costs, void* out_c, int* out_i, void* ws, hipStream_t st) {
    return ICEM_DISPATCH(h, launch_topk<float>(n, K, costs, out_c, out_i, ws, st),
                         launch_topk<double>(n, K, costs, out_c, out_i, ws, st));
}

template <typename T>
static int topk_partial_t(const icem_handle* h, int n_cand, int K, const void* costs, void* ws, int nblk, hipStream_t st) {
    T* pc;
    int* pi;
    split_partial_ws<T>(ws, nblk, K, &pc, &pi);
    ProfScope prof(h, ICEM_K_TOPK_PARTIAL, n_cand, st);
    hipLaunchKernelGGL((topk_partial_kernel<T>), dim3(nblk), dim3(WG), 0, st, n_cand, K, (const T*)costs, pc, pi);
    return ICEM_OK;
}
int gk_topk_partial(const icem_handle* h, int n_cand, int K, const void* costs, void* ws, int nblk, hipStream_t st) {
    ICEM_DISPATCH(h, topk_partial_t<float>(h, n_cand, K, costs, ws, nblk, st), topk_partial_t<double>(h, n_cand, K, costs, ws, nblk, st));
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

template <typename T>
static int local_pack_t(const icem_handle* h, int nblk, int K, int n_loc, int shard_lo, int n_global, const void* ws,
                        const void* actions, void* records, hipStream_t st) {
    T* pc;
    int* pi;
    split_partial_ws<T>(const_cast<void*>(ws), nblk, K, &pc, &pi);
    ProfScope prof(h, ICEM_K_LOCAL_PACK, nblk * K, st);
    hipLaunchKernelGGL((local_pack_kernel<T>), dim3(1), dim3(WG), 0, st, nblk * K, K, h->hd, n_loc, shard_lo, n_global,
                       (const T*)pc, (const int*)pi, (const T*)actions, (T*)records);
    return ICEM_OK;
}
int gk_local_pack(const icem_handle* h, int nblk, int K, int n_loc, int shard_lo, int n_global, const void* ws,
                  const void* actions, void* records, hipStream_t st) {
    ICEM_DISPATCH(h, local_pack_t<float>(h, nblk, K, n_loc, shard_lo, n_global, ws, actions, records, st),
                  local_pack_t<double>(h, nblk, K, n_loc, shard_lo, n_global, ws, actions, records, st));
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int gk_gather_refit(const icem_handle* h, const void* actions, const int32_t* idx, int k, void* mean, void* std,
                    void* elites_out, hipStream_t st) {
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

int gk_shift(const icem_handle* h, void* mean, void* std, const void* low, const void* high, hipStream_t st) {
    if (h->cfg.dtype == ICEM_F64)
        hipLaunchKernelGGL((shift_kernel<double>), dim3(1), dim3(WG), (size_t)h->hd * 8, st, h->cfg.horizon, h->cfg.act_dim,
                           (double)h->cfg.init_std, (double*)mean, (double*)std, (const double*)low, (const double*)high);
    else
        hipLaunchKernelGGL((shift_kernel<float>), dim3(1), dim3(WG), (size_t)h->hd * 4, st, h->cfg.horizon, h->cfg.act_dim,
                           (float)h->cfg.init_std, (float*)mean, (float*)std, (const float*)low, (const float*)high);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int gk_reset(const icem_handle* h, void* mean, void* std, const void* low, const void* high, hipStream_t st) {
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

int gk_shift_elites(const icem_handle* h, int n_extra, const void* elites, void* dst, hipStream_t st) {
    const icem_config& c = h->cfg;
    if (c.dtype == ICEM_F64)
        hipLaunchKernelGGL((shift_elites_kernel<double>), dim3(1), dim3(WG), 0, st, n_extra, c.horizon, c.act_dim,
                           (const double*)elites, (double*)dst);
    else
        hipLaunchKernelGGL((shift_elites_kernel<float>), dim3(1), dim3(WG), 0, st, n_extra, c.horizon, c.act_dim,
                           (const float*)elites, (float*)dst);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

template <typename T>
static void merge_refit_t(const icem_handle* h, const MergeArgsV& v, hipStream_t st) {
    MergeArgs<T> a;
    a.n_rec = v.n_rec;
    a.n_keep = v.n_keep;
    a.K = v.K;
    a.h = v.h;
    a.d = v.d;
    a.n_global = v.n_global;
    a.last = v.last;
    a.alpha = (T)v.alpha;
    a.init_std = (T)v.init_std;
    a.records = (const T*)v.records;
    a.elites_cur = (const T*)v.elites_cur;
    a.elites_cost_cur = (const T*)v.elites_cost_cur;
    a.elites_next = (T*)v.elites_next;
    a.elites_cost_next = (T*)v.elites_cost_next;
    a.mean_in = (const T*)v.mean_in;
    a.std_in = (const T*)v.std_in;
    a.mean = (T*)v.mean;
    a.std = (T*)v.std;
    a.low = (const T*)v.low;
    a.high = (const T*)v.high;
    a.executed = (T*)v.executed;
    a.best_cost = (T*)v.best_cost;
    a.xw = v.xw;
    ProfScope prof(h, ICEM_K_MERGE_REFIT, a.n_rec + a.n_keep, st);
    hipLaunchKernelGGL((merge_refit_kernel<T>), dim3(1), dim3(WG), (size_t)v.h * v.d * sizeof(T), st, a);
}
// world == 1: selection + gather + refit of an iteration in one launch (select_refit_kernel).  The worst case of the
// threshold -- K threads of each of the workgroup's waves hold ALL their keys at or below it -- must fit the candidate
// array; beyond that size the three-launch path runs.
bool gk_select_ok(const icem_handle* h, int n_cand, int n_keep, int K) {
    const bool off = opt_i(OPT_GK_SELECT) == 0;     // read per call (path-equivalence test)
    if (off || h->cfg.world != 1 || K < 1 || K > ICEM_MAX_ELITES) return false;
    const long long per_thread = ((long long)n_cand + n_keep + SELECT_NT - 1) / SELECT_NT;
    return (long long)(SELECT_NT / 64) * K * per_thread <= SELECT_CAP;
}

template <typename T>
static void select_refit_t(const icem_handle* h, int n_cand, int n_loc, const void* costs, const void* actions, const MergeArgsV& v,
                           hipStream_t st) {
    SelectArgs<T> s;
    s.n_cand = n_cand;
    s.n_loc = n_loc;
    s.cap = SELECT_CAP;
    s.costs = (const T*)costs;
    s.actions = (const T*)actions;
    s.dbg = h->dbg;
    MergeArgs<T>& a = s.m;
    a.n_rec = 0;
    a.n_keep = v.n_keep;
    a.K = v.K;
    a.h = v.h;
    a.d = v.d;
    a.n_global = v.n_global;
    a.last = v.last;
    a.alpha = (T)v.alpha;
    a.init_std = (T)v.init_std;
    a.records = nullptr;
    a.elites_cur = (const T*)v.elites_cur;
    a.elites_cost_cur = (const T*)v.elites_cost_cur;
    a.elites_next = (T*)v.elites_next;
    a.elites_cost_next = (T*)v.elites_cost_next;
    a.mean_in = (const T*)v.mean_in;
    a.std_in = (const T*)v.std_in;
    a.mean = (T*)v.mean;
    a.std = (T*)v.std;
    a.low = (const T*)v.low;
    a.high = (const T*)v.high;
    a.executed = (T*)v.executed;
    a.best_cost = (T*)v.best_cost;
    a.xw = XchgWait{};
    ProfScope prof(h, ICEM_K_MERGE_REFIT, n_cand + v.n_keep, st);
    hipLaunchKernelGGL((select_refit_kernel<T>), dim3(1), dim3(SELECT_NT), (size_t)v.h * v.d * sizeof(T), st, s);
}
int gk_select_refit(const icem_handle* h, int n_cand, int n_loc, const void* costs, const void* actions, const MergeArgsV& a,
                    hipStream_t st) {
    if (h->cfg.dtype == ICEM_F64)
        select_refit_t<double>(h, n_cand, n_loc, costs, actions, a, st);
    else
        select_refit_t<float>(h, n_cand, n_loc, costs, actions, a, st);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int gk_merge_refit(const icem_handle* h, const MergeArgsV& a, hipStream_t st) {
    if (h->cfg.dtype == ICEM_F64)
        merge_refit_t<double>(h, a, st);
    else
        merge_refit_t<float>(h, a, st);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

}  // namespace icem
