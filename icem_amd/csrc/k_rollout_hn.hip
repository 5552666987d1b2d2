// k_rollout_hn.hip -- K2 + K3 for the narrow shapes the reference ships besides HalfCheetah -- Door (d = 28, o = 39;
// icem/environments/mjenvs.py:57-78), Relocate (d = 30, o = 39; mjenvs.py:155-174), FetchPickAndPlace (d = 4, o = 28;
// icem/environments/robotics.py:150-164) -- with their costs as icem_cost_terms: rollout_hn_kernel, one wavefront per 16
// trajectories, the model step (abstract_models.py:17-26) on the 16-bit matrix cores, the cost evaluated across the four
// lanes of a trajectory.  Before round 5 these handles rolled out on the exact-f32 GEMM kernel (k_rollout_wide.hip), whose
// waves stream the model from L2 every step and walk the term list through dependent LDS reads: 2.7-2.9 us per step for a
// lone wave, 82 us per launch at N = 4096 (EXPERIMENTS R4.8) against 5.6 us of steps in the HalfCheetah tile kernels.
//
// TileHN is Tile16H (fused_dev.h) for NT = ceil(o / 16) <= 3 output tiles, ALL on the matrix cores: contraction block kb of
// 32 slots = the 16 columns of tile kb (slots 8g .. 8g+3 of lane group g: the lane's own accumulators of the previous step)
// + 16 action entries e = 16 kb + 4 q + g (slots 8g+4 ..).  One v_mfma_f32_16x16x32_f16 per (output tile, block, product):
// 3 NT^2 per step, the model's NT^2 x 2 planes resident in registers (72 at NT = 3: the kernel runs at two waves per SIMD).
// Same scales as Tile16H: one power of two per launch for the state (from |obs0| and the action bound), one each for A and B.
// The cost: every step the lanes park the UNSCALED pre-action observation of their trajectory in an LDS row (the term list
// reads arbitrary slices of it: norms, gates, thresholds), lane g evaluates terms g, g + 4 of icem_cost_terms, lane 0 the
// icem_cost_spec part and the health term, the control cost comes from the action entries every lane holds anyway; one
// permlane reduction sums the four shares.  The difference term (Ant / Hopper: next_obs - obs) closes a step's cost one step
// later, when the next observation is in the row.  cost_along_trajectory as k_rollout_wide.hip (np.amin: a NaN step cost makes
// the trajectory's cost NaN).
#include "fused_dev.h"
#include "wide_dev.h"

// shapes (H, D, o) with a compiled TileHN rollout: the reference's door / relocate / fpp settings at h = 30
#ifndef ICEM_HN_SHAPES
#define ICEM_HN_SHAPES(X) X(30, 28, 39) X(30, 30, 39) X(30, 4, 28)
#endif

namespace icem {

namespace {

// A workgroup barrier that orders LDS traffic ONLY: __syncthreads() carries a release / acquire fence that also waits for every
// global load in flight (s_waitcnt vmcnt(0)) -- the staging wave's prefetch of the next action chunk, requested a moment
// earlier: a cold HBM round trip at the barrier of every chunk, for every wave of the workgroup (half of a launch).  The step
// loops below exchange through LDS alone; the prefetch's registers are waited for where they are used.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int HN_MAX_WAVES = 4;   // one per SIMD: the tile's registers (model planes + two operand planes of the state + staging) reach 260-340 at d = 30

// N32 / N4 / NP: the term list's compile-time shape (hn_cost_program, abi.hip sorts the handle's terms into it and pads with
// null terms): slots [0, N32) slice terms of up to 32 entries, [N32, N32 + N4) slice terms of up to 4, then NP point terms
// (ICEM_TERM_STEP_GT / _SQ_OFFSET).  Evaluated by ALL four lanes of a trajectory together, branch-free: a slice's entries are
// spread over the lanes (entry m in lane m % 4), every lane issues its LDS reads of all terms back to back, one permlane
// reduction per slice.  (One lane per term walking its slice -- wide_step_cost_lanes -- is a chain of dependent LDS reads:
// Door's 30-entry velocity term alone was 3.6 us per step for a lone wave; EXPERIMENTS R5.4.)
template <int H, int D, int O, int KIND, int N32, int N4, int NP>
struct TileHN {
    static constexpr int NT = (O + 15) / 16;
    static constexpr int OP = 16 * NT;                 // padded observation width
    static constexpr int RS = OP + 4;                  // floats per trajectory row of the observation rows in LDS
    static constexpr int SLACK = 4, TAIL = 8;          // (StreamT's staging layout)
    static_assert(NT >= 1 && NT <= 3 && D <= 16 * NT && D >= 1, "up to 48 observation entries, 16 action entries per block");
    // action entry group (kb, q) holds entries 16 kb + 4 q + g: all four lanes valid, none, or the first D % 4 of them
    static constexpr int EV(int kb, int q) { return 16 * kb + 4 * q; }
    static constexpr bool all_valid(int kb, int q) { return EV(kb, q) + 3 < D; }
    static constexpr bool none_valid(int kb, int q) { return EV(kb, q) >= D; }

    unsigned aH[NT][NT][4], aL[NT][NT][4];   // [output tile][contraction block]: planes of slots 8g .. 8g+7
    f32x4 obs_init[NT];
    float T, invT, invM, sact;               // accumulator scale, its inverse, 1 / model scale, the actions' scale into the B operand
    float ctrl_w, part_w;                    // control weight; 1 / 0: this lane holds a valid entry of the partial group
    int part_off;                            // ... and where it reads that group's entry (a valid lane: in place; else action 0)
    WideCost wc;
    const CostArgs<float>* csg;              // the term list in device memory, sorted into the program's slots (read only if N32 + N4 + NP > 0)
    float* row;                              // this lane's trajectory row of the wave's observation rows
    float pen_g, lin_g, ksum;                // icem_cost_spec's flip penalty / linear weight in lane group 0, 0 elsewhere; 1 (sum) / 0 (final)
    int flip_i;
    int g, cost_mode;
    float sM, sB, act_mag;

    // A [o, lda] row-major (x' = x A + a B), B [d, ldb]
    // with_model = false: a wave that only evaluates costs (rollout_hn_pair_kernel) leaves the model's planes alone
    __device__ __forceinline__ void load(const FastRolloutArgs& a, const float* A, int lda, const float* B, int ldb, int o, int lane,
                                         bool with_model = true) {
        const int j = lane & 15;
        g = lane >> 4;
        sM = a.m_scale;
        sB = a.b_scale;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            if (!with_model) break;
            const int i = 16 * c + j;   // output column
#pragma unroll
            for (int kb = 0; kb < NT; ++kb) {
                float m[8];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = 16 * kb + 4 * g + s;
                    m[s] = (k < o && i < o) ? A[(size_t)k * lda + i] * sM : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = EV(kb, q) + g;
                    m[4 + q] = (e < D && i < o) ? B[(size_t)e * ldb + i] * sB : 0.f;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) split_pair_f16(m[2 * p], m[2 * p + 1], aH[c][kb][p], aL[c][kb][p]);
            }
        }
        ctrl_w = a.ctrl_w;
        part_w = g < (D & 3) ? 1.f : 0.f;
        part_off = g < (D & 3) ? 0 : -(4 * (D / 4) + g);   // entry 4 (D / 4) + g of the step -> entry 0
        act_mag = a.act_mag;
        cost_mode = a.cost_mode;
        ksum = a.cost_mode == 0 ? 1.f : 0.f;
    }
    __device__ __forceinline__ void set_cost(const WideCost& w, const CostArgs<float>* terms) {
        wc = w;
        csg = terms;
        pen_g = (w.flip_idx >= 0 && g == 0) ? w.flip_pen : 0.f;
        lin_g = (w.lin_w != 0.f && g == 0) ? w.lin_w : 0.f;   // a zero weight drops the term (icem_cost_spec)
        flip_i = w.flip_idx >= 0 ? w.flip_idx : 0;
    }

    // obs: OP floats in LDS (natural order, zeros behind entry o - 1); rows: this wave's [16][RS] observation rows
    __device__ __forceinline__ void load_obs(const float* obs, float* rows) {
        const int lane = (int)(threadIdx.x & 63);
        float mx = lane < OP ? __builtin_fabsf(obs[lane < OP ? lane : 0]) : 0.f;
        mx = mx != mx ? 0.f : mx;
        float mm = __uint_as_float(~wave_min_u32(~__float_as_uint(mx)));
        mm = mm > act_mag ? mm : act_mag;
        if (KIND == 1) mm = mm > 1.f ? mm : 1.f;
        int ex = (int)((__float_as_uint(mm) >> 23) & 0xFF) - 127;
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        auto uni = [](float x) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
        const float S = __uint_as_float((unsigned)(127 + 4 - ex) << 23);
        T = uni(S * sM);
        invT = uni(__uint_as_float((unsigned)(127 - 4 + ex) << 23) / sM);
        invM = uni(1.f / sM);
        sact = uni(T / sB);
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s) obs_init[c][s] = obs[16 * c + 4 * g + s] * T;
        row = rows + (lane & 15) * RS;
    }

    __device__ __forceinline__ const float* read_ptr(const float* buf, int lane, int stride) const {
        return buf + SLACK + (lane & 15) * stride + (lane >> 4);
    }

    struct State {
        f32x4 cur[NT];     // T x the lane's columns 16 c + 4 g .. + 3
        float acc_s, acc_b;
    };
    __device__ __forceinline__ void init(State& st) const {
#pragma unroll
        for (int c = 0; c < NT; ++c) st.cur[c] = obs_init[c];
        st.acc_s = 0.f;
        st.acc_b = INFINITY;
    }
    // the trajectory's UNSCALED observation -> its LDS row (every lane its own 4 NT columns)
    __device__ __forceinline__ void park(const State& st) const {
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = st.cur[c][s] * invT;
            *reinterpret_cast<f32x4*>(row + 16 * c + 4 * g) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // this step's action entries (entry 16 kb + 4 q + g at rd[16 kb + 4 q]) and the control cost's share of this lane
    __device__ __forceinline__ float actions_of(const float* rd, float (&xv)[NT][4]) const {
        float u = 0.f;
#pragma unroll
        for (int kb = 0; kb < NT; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (none_valid(kb, q)) {
                    xv[kb][q] = 0.f;
                } else if (all_valid(kb, q)) {
                    xv[kb][q] = rd[EV(kb, q)];
                    u = __builtin_fmaf(xv[kb][q], xv[kb][q], u);
                } else {   // the first D % 4 lane groups hold an entry; the others read a finite stand-in that weighs nothing
                    xv[kb][q] = rd[EV(kb, q) + part_off] * part_w;
                    u = __builtin_fmaf(xv[kb][q], xv[kb][q], u);
                }
            }
        return u;
    }
    // the step's cost from the pre-action observation in row `x` (this lane's trajectory) and the control cost's share u: the
    // four lanes' shares (control cost, lane group 0's icem_cost_spec terms) summed by one reduction, the term list on top (the
    // same value in all four lanes)
    __device__ __forceinline__ void cost_step(State& st, float u, const float* x) const {
        float c = u * ctrl_w;
        {
            const float ang = x[flip_i];
            c += (ang > wc.flip_th) ? pen_g : 0.f;
            c += (ang < -wc.flip_th) ? pen_g : 0.f;
            c = __builtin_fmaf(lin_g, x[wc.lin_idx], c);
        }
        c = reduce_groups(c) + terms_all(x);
        st.acc_s = __builtin_fmaf(st.acc_s, ksum, c);
        st.acc_b = (c < st.acc_b || c != c) ? c : st.acc_b;   // np.amin: a NaN step cost makes the trajectory's cost NaN
    }
    // the model step: B operand planes per contraction block -- own columns (x 1 / sM: from the accumulators' scale to the
    // operand's), actions -- then 3 NT^2 MFMAs
    __device__ __forceinline__ void model_step(State& st, const float (&xv)[NT][4]) const {
        unsigned bH[NT][4], bL[NT][4];
#pragma unroll
        for (int kb = 0; kb < NT; ++kb) {
            split_pair_f16_scaled(st.cur[kb][0], invM, st.cur[kb][1], invM, bH[kb][0], bL[kb][0]);
            split_pair_f16_scaled(st.cur[kb][2], invM, st.cur[kb][3], invM, bH[kb][1], bL[kb][1]);
            if (none_valid(kb, 0)) bH[kb][2] = bL[kb][2] = 0u;
            else split_pair_f16_scaled(xv[kb][0], sact, xv[kb][1], sact, bH[kb][2], bL[kb][2]);
            if (none_valid(kb, 2)) bH[kb][3] = bL[kb][3] = 0u;
            else split_pair_f16_scaled(xv[kb][2], sact, xv[kb][3], sact, bH[kb][3], bL[kb][3]);
        }
        // NT independent accumulator chains, smallest products first
        f32x4 nxt[NT];
#pragma unroll
        for (int c2 = 0; c2 < NT; ++c2) nxt[c2] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NT; ++kb)
#pragma unroll
            for (int c2 = 0; c2 < NT; ++c2) nxt[c2] = mfma_f16_32(aL[c2][kb], bH[kb], nxt[c2]);
#pragma unroll
        for (int kb = 0; kb < NT; ++kb)
#pragma unroll
            for (int c2 = 0; c2 < NT; ++c2) nxt[c2] = mfma_f16_32(aH[c2][kb], bL[kb], nxt[c2]);
#pragma unroll
        for (int kb = 0; kb < NT; ++kb)
#pragma unroll
            for (int c2 = 0; c2 < NT; ++c2) nxt[c2] = mfma_f16_32(aH[c2][kb], bH[kb], nxt[c2]);
#pragma unroll
        for (int c2 = 0; c2 < NT; ++c2) {
            if (KIND == 1) {
#pragma unroll
                for (int s = 0; s < 4; ++s) st.cur[c2][s] = fast_tanh(nxt[c2][s] * invT) * T;
            } else {
                st.cur[c2] = nxt[c2];
            }
        }
    }
    // ... into another buffer of rows (rollout_hn_pair_kernel's two: `off` floats from the first), no wave-level fence: the
    // reader is another wave, behind a workgroup barrier
    __device__ __forceinline__ void park_to(const State& st, int off) const {
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            f32x4 v;
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = st.cur[c][s] * invT;
            *reinterpret_cast<f32x4*>(row + off + 16 * c + 4 * g) = v;
        }
    }
    __device__ __forceinline__ void step(State& st, const float* rd) const {
        float xv[NT][4];
        const float u = actions_of(rd, xv);
        // (the cost block BEHIND the MFMAs -- in their shadow, the accumulators are not needed before the next step -- measured
        //  slower: 71 instead of 57.5 us per Door launch at N = 4096)
        park(st);
        cost_step(st, u, row);
        model_step(st, xv);
    }
    // The term parameters are read through the CONSTANT address space (scalar loads: wave-uniform values in scalar registers,
    // selects on them are scalar selects, nothing is exec-masked); x: the trajectory's observation row in LDS.
    typedef const __attribute__((address_space(4))) typename CostArgs<float>::Term* TermP;
    __device__ __forceinline__ TermP term(int j) const {
        return reinterpret_cast<TermP>(reinterpret_cast<const __attribute__((address_space(4))) char*>((unsigned long long)csg) +
                                       offsetof(CostArgs<float>, terms) + (size_t)j * sizeof(typename CostArgs<float>::Term));
    }
    static __device__ __forceinline__ float gate_of(TermP tm, const float* x) {
        const int gi = tm->gate_idx;
        const float gv = x[gi >= 0 ? gi : 0];
        return gi >= 0 ? (gv > tm->gate_th ? 1.f : 0.f) : 1.f;   // a product, as in the reference (NaN * 0 = NaN)
    }
    template <int PASSES>
    __device__ __forceinline__ float slice_term(TermP tm, const float* x) const {
        const int ia = tm->a, tb = tm->b, len = tm->len, kind = tm->kind;
        const int ib = tb >= 0 ? tb : ia;
        const float hb = tb >= 0 ? 1.f : 0.f;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int m = g + 4 * i;
            const bool ok = m < len;
            const int mo = ok ? m : 0;
            float dv = __builtin_fmaf(-hb, x[ib + mo], x[ia + mo]);
            dv = ok ? dv : 0.f;
            acc = __builtin_fmaf(dv, dv, acc);
        }
        acc = reduce_groups(acc);
        const float r = __builtin_amdgcn_sqrtf(acc), th = tm->th;
        float f = kind == ICEM_TERM_SUMSQ ? acc : (kind == ICEM_TERM_NORM ? r : (kind == ICEM_TERM_NORM_GT ? (r > th ? 1.f : 0.f) : (r < th ? 1.f : 0.f)));
        f *= gate_of(tm, x);
        return kind < 0 ? 0.f : tm->w * f;   // (kind -1: a padding slot)
    }
    __device__ __forceinline__ float point_term(TermP tm, const float* x) const {
        const int kind = tm->kind;
        const float v = x[tm->a], th = tm->th;
        const float dv = v - th;
        float f = kind == ICEM_TERM_STEP_GT ? (v > th ? 1.f : 0.f) : dv * dv;
        f *= gate_of(tm, x);
        return kind < 0 ? 0.f : tm->w * f;
    }
    // ---- the cost in pieces, for kernels that spread it over two waves (rollout_hn_split_kernel): the part in front of the term
    // list, one term's value, the accumulation -- composed in cost_step's order they give cost_step's bits
    static constexpr int NTERMS = N32 + N4 + NP;
    static constexpr int NB = N32 > 0 ? N32 : N4 / 2;   // the terms a second cost wave takes: the long slices, else half of the short ones
    __device__ __forceinline__ float base_cost(float u, const float* x) const {
        float c = u * ctrl_w;
        const float ang = x[flip_i];
        c += (ang > wc.flip_th) ? pen_g : 0.f;
        c += (ang < -wc.flip_th) ? pen_g : 0.f;
        c = __builtin_fmaf(lin_g, x[wc.lin_idx], c);
        return reduce_groups(c);
    }
    template <int J>
    __device__ __forceinline__ float term_val(const float* x) const {
        if constexpr (J < N32) return slice_term<8>(term(J), x);
        else if constexpr (J < N32 + N4) return slice_term<1>(term(J), x);
        else return point_term(term(J), x);
    }
    __device__ __forceinline__ void acc_step(State& st, float c) const {
        st.acc_s = __builtin_fmaf(st.acc_s, ksum, c);
        st.acc_b = (c < st.acc_b || c != c) ? c : st.acc_b;   // np.amin: a NaN step cost makes the trajectory's cost NaN
    }
    // the term list, the same value in all four lanes of the trajectory
    __device__ __forceinline__ float terms_all(const float* x) const {
        float c = 0.f;
#pragma unroll
        for (int j = 0; j < N32; ++j) c += slice_term<8>(term(j), x);
#pragma unroll
        for (int j = 0; j < N4; ++j) c += slice_term<1>(term(N32 + j), x);
#pragma unroll
        for (int j = 0; j < NP; ++j) c += point_term(term(N32 + N4 + j), x);
        return c;
    }
    __device__ __forceinline__ float cost(const State& st) const { return cost_mode == 1 ? st.acc_b : st.acc_s; }
};

struct HnArgs {
    FastRolloutArgs r;           // n_rows, n_cand, K, o, cost_mode, obs0, actions, costs, part_*, ctrl_w, act_mag, m_scale, b_scale
    const float* A;              // [o, lda] row-major f32
    const float* B;              // [d, ldb]
    int lda, ldb;
    WideCost wc;
    const CostArgs<float>* cs;   // device copy of the cost terms, nullptr: none
};

template <int H, int D, int O, int KIND, int WAVES, int N32, int N4, int NP>
__global__ __launch_bounds__(64 * WAVES) void rollout_hn_kernel(HnArgs a) {
    using Tile = TileHN<H, D, O, KIND, N32, N4, NP>;
    using Stream = StreamT<Tile, H, D>;
    __shared__ __attribute__((aligned(16))) float stage[WAVES][Stream::STG];
    __shared__ __attribute__((aligned(16))) float rows[WAVES][16 * Tile::RS];
    __shared__ unsigned long long wg_keys[2][WAVES][32];
    __shared__ __attribute__((aligned(16))) float obs_stage[Tile::OP];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const float obs_reg = a.r.obs0[((int)threadIdx.x < Tile::OP && (int)threadIdx.x < a.r.o) ? threadIdx.x : 0];
    Tile tile;
    tile.load(a.r, a.A, a.lda, a.B, a.ldb, a.r.o, lane);
    tile.set_cost(a.wc, a.cs);
    Stream stream;
    stream.init(tile, stage[wave], lane);
    const int tiles = (a.r.n_rows + 15) / 16;
    const int tile0 = wave * gridDim.x + blockIdx.x;
    typename Stream::Vec pre[Stream::NLD];
    if ((int)threadIdx.x < Tile::OP) obs_stage[threadIdx.x] = (int)threadIdx.x < a.r.o ? obs_reg : 0.f;
    __syncthreads();
    tile.load_obs(obs_stage, rows[wave]);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    for (int tile_id = tile0; tile_id < tiles; tile_id += WAVES * gridDim.x) {
        stream.first_loads(a.r.actions, a.r.n_rows, tile_id, pre);
        run_key = stream.run(tile, a.r, tile_id, lane, run_key, first, pre);
        first = false;
    }
    if (a.r.K > 0) wg_merge_emit<WAVES>(wg_keys, run_key, a.r.K, lane, wave, a.r);
}

// The same rollout with TWO waves per tile for populations that leave the chip mostly empty (at most two tiles per CU): a lone
// wave issues a step's 150-330 vector instructions and 12-27 MFMAs one after the other, 1.0-2.3 us per step.  Here wave 2p of
// a workgroup is tile p's MODEL wave (actions -> operand planes -> MFMAs -> tanh -> the next observation parked in the pair's
// LDS rows) and wave 2p + 1 its COST wave (control cost, icem_cost_spec part and the term list of the observation parked one
// barrier earlier), one workgroup barrier per step, the observation rows and the staged action chunks double-buffered between
// them.  Same operations in the same order on the same values as rollout_hn_kernel: the same bits (tested).
template <int H, int D, int O, int KIND, int PAIRS, int N32, int N4, int NP>
__global__ __launch_bounds__(128 * PAIRS) void rollout_hn_pair_kernel(HnArgs a) {
    using Tile = TileHN<H, D, O, KIND, N32, N4, NP>;
    using Stream = StreamT<Tile, H, D>;
    constexpr int TC = Stream::TC, NCH = Stream::NCH, C4 = Stream::C4, CBP = Stream::CBP, VW = Stream::VW, NLD = Stream::NLD;
    constexpr int ROWS = 16 * Tile::RS;
    __shared__ __attribute__((aligned(16))) float stage[PAIRS][2][Stream::STG];
    __shared__ __attribute__((aligned(16))) float rows[PAIRS][2][ROWS];
    __shared__ unsigned long long wg_keys[2][PAIRS][32];
    __shared__ __attribute__((aligned(16))) float obs_stage[Tile::OP];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int pair = wave >> 1;
    const bool model = (wave & 1) == 0;
    const float obs_reg = a.r.obs0[((int)threadIdx.x < Tile::OP && (int)threadIdx.x < a.r.o) ? threadIdx.x : 0];
    Tile tile;
    tile.load(a.r, a.A, a.lda, a.B, a.ldb, a.r.o, lane, model);
    tile.set_cost(a.wc, a.cs);
    Stream stream;
    stream.init(tile, stage[pair][0], lane);
    const int tiles = (a.r.n_rows + 15) / 16;
    if ((int)threadIdx.x < Tile::OP) obs_stage[threadIdx.x] = (int)threadIdx.x < a.r.o ? obs_reg : 0.f;
    __syncthreads();
    tile.load_obs(obs_stage, rows[pair][0]);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    // (every pair of the workgroup walks the same number of tiles: the barriers are the workgroup's; a pair past the end rolls a
    //  clamped tile out and drops it)
    const int stride = PAIRS * (int)gridDim.x;
    const int rounds = (tiles - (int)blockIdx.x + stride - 1) / stride;
    for (int rnd = 0; rnd < rounds; ++rnd) {
        const int tile_id = rnd * stride + pair * (int)gridDim.x + (int)blockIdx.x;
        const bool tile_on = tile_id < tiles;
        const int tid_c = tile_on ? tile_id : 0;
        const int row = tid_c * 16 + (lane & 15);
        const bool live = tile_on && row < a.r.n_rows;
        typename Stream::Vec pre[NLD];
        const typename Stream::Vec* src[NLD];
        if (model) {
#pragma unroll
            for (int m = 0; m < NLD; ++m) {
                const int r = tid_c * 16 + stream.ld_row[m];
                src[m] = reinterpret_cast<const typename Stream::Vec*>(a.r.actions + (size_t)(r < a.r.n_rows ? r : 0) * (H * D)) + stream.ld_c4[m];
                pre[m] = src[m][0];
            }
        }
        typename Tile::State st;
        tile.init(st);
        if (model) {   // chunk 0 -> staging buffer 0, chunk 1 requested; the start observation -> rows buffer 0
#pragma unroll
            for (int m = 0; m < NLD; ++m)
                if (stream.ld_on[m]) *reinterpret_cast<typename Stream::Vec*>(&stage[pair][0][Tile::SLACK + stream.ld_row[m] * CBP + VW * stream.ld_c4[m]]) = pre[m];
            if (1 < NCH) {
#pragma unroll
                for (int m = 0; m < NLD; ++m) pre[m] = src[m][C4];
            }
            tile.park_to(st, 0);
        }
        __syncthreads();
        for (int ch = 0; ch < NCH; ++ch) {
            const int cb = ch & 1;
#pragma unroll
            for (int ts = 0; ts < TC; ++ts) {
                const int t = ch * TC + ts;
                const float* rd = stream.rd0 + cb * Stream::STG + ts * D;
                float xv[Tile::NT][4];
                const float u = tile.actions_of(rd, xv);
                if (model) {
                    if (ts == TC - 1 && ch + 1 < NCH) {   // the next chunk -> the other staging buffer (read from the next step on)
#pragma unroll
                        for (int m = 0; m < NLD; ++m)
                            if (stream.ld_on[m])
                                *reinterpret_cast<typename Stream::Vec*>(&stage[pair][cb ^ 1][Tile::SLACK + stream.ld_row[m] * CBP + VW * stream.ld_c4[m]]) = pre[m];
                        if (ch + 2 < NCH) {
#pragma unroll
                            for (int m = 0; m < NLD; ++m) pre[m] = src[m][(ch + 2) * C4];
                        }
                    }
                    tile.model_step(st, xv);
                    if (t + 1 < H) tile.park_to(st, ((t + 1) & 1) * ROWS);
                } else {
                    tile.cost_step(st, u, tile.row + (t & 1) * ROWS);
                }
                lds_barrier();
            }
        }
        if (!model) {
            const float cost = tile.cost(st);
            if (live && lane < 16) a.r.costs[row] = cost;
            note_nonfinite(a.r, cost, live && lane < 16);
            if (a.r.K > 0) {
                const unsigned long long key = (lane < 16 && live && row < a.r.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
                run_key = topk_push16<false>(run_key, key, first, a.r.K, lane);   // (the per-lane predicate: see bitonic_step)
            }
            first = false;
        }
    }
    if (a.r.K > 0) wg_merge_emit<PAIRS>(wg_keys, run_key, a.r.K, lane, model ? PAIRS : pair, a.r);
}

// ... and with the MODEL split over the output tiles as well: NT model waves + one cost wave per tile, for populations of at most
// one tile per CU.  Model wave c owns output tile c: its planes of the model (a third of TileHN's operand registers), the
// lane's four columns of that tile, their operand planes and those of action block c; the waves exchange the operand planes
// through LDS (lane to SAME lane: 2 x 16 bytes per lane, block and step), one workgroup barrier per step; each runs its own
// accumulator's 3 NT MFMAs in TileHN's order, so every accumulator sees the products it sees there: the same bits.
template <int H, int D, int O, int KIND, int N32, int N4, int NP>
struct TileHNc {
    using Full = TileHN<H, D, O, KIND, N32, N4, NP>;
    static constexpr int NT = Full::NT, OP = Full::OP, RS = Full::RS;
    unsigned mH[NT][4], mL[NT][4];   // [contraction block]: planes of slots 8g .. 8g+7 of output tile c
    f32x4 obs_init;
    float T, invT, invM, sact, part_w;
    int part_off, c, g;
    float sM, sB, act_mag;
    float* row;

    __device__ __forceinline__ void load(const FastRolloutArgs& a, const float* A, int lda, const float* B, int ldb, int o, int lane, int tile_c) {
        const int j = lane & 15;
        g = lane >> 4;
        c = tile_c;
        sM = a.m_scale;
        sB = a.b_scale;
        const int i = 16 * c + j;   // output column
#pragma unroll
        for (int kb = 0; kb < NT; ++kb) {
            float m[8];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 16 * kb + 4 * g + s;
                m[s] = (k < o && i < o) ? A[(size_t)k * lda + i] * sM : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = Full::EV(kb, q) + g;
                m[4 + q] = (e < D && i < o) ? B[(size_t)e * ldb + i] * sB : 0.f;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) split_pair_f16(m[2 * p], m[2 * p + 1], mH[kb][p], mL[kb][p]);
        }
        part_w = g < (D & 3) ? 1.f : 0.f;
        part_off = g < (D & 3) ? 0 : -(4 * (D / 4) + g);
        act_mag = a.act_mag;
    }
    __device__ __forceinline__ void load_obs(const float* obs, float* rows) {
        const int lane = (int)(threadIdx.x & 63);
        float mx = lane < OP ? __builtin_fabsf(obs[lane < OP ? lane : 0]) : 0.f;
        mx = mx != mx ? 0.f : mx;
        float mm = __uint_as_float(~wave_min_u32(~__float_as_uint(mx)));
        mm = mm > act_mag ? mm : act_mag;
        if (KIND == 1) mm = mm > 1.f ? mm : 1.f;
        int ex = (int)((__float_as_uint(mm) >> 23) & 0xFF) - 127;
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        auto uni = [](float x) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
        const float S = __uint_as_float((unsigned)(127 + 4 - ex) << 23);
        T = uni(S * sM);
        invT = uni(__uint_as_float((unsigned)(127 - 4 + ex) << 23) / sM);
        invM = uni(1.f / sM);
        sact = uni(T / sB);
#pragma unroll
        for (int s = 0; s < 4; ++s) obs_init[s] = obs[16 * c + 4 * g + s] * T;
        row = rows + (lane & 15) * RS;
    }
    // this lane's operand planes of block c for the step whose actions are at rd: own columns, then the block's action entries
    // block c's action entries of the step at rd (read early: they do not depend on the state)
    __device__ __forceinline__ void actions_c(const float* rd, float (&xv)[4]) const {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ev = 16 * c + 4 * q;
            xv[q] = ev >= D ? 0.f : (ev + 3 < D ? rd[ev] : rd[ev + part_off] * part_w);
        }
    }
    __device__ __forceinline__ void planes(const f32x4& cur, const float* rd, unsigned (&pH)[4], unsigned (&pL)[4]) const {
        float xv[4];
        actions_c(rd, xv);
        planes_xv(cur, xv, pH, pL);
    }
    __device__ __forceinline__ void planes_xv(const f32x4& cur, const float (&xv)[4], unsigned (&pH)[4], unsigned (&pL)[4]) const {
        planes_state(cur, pH, pL);
        planes_actions(xv, pH, pL);
    }
    // the two halves of a lane's operand planes: slots 0-1 from the state (behind the MFMAs), slots 2-3 from the actions (any time)
    __device__ __forceinline__ void planes_state(const f32x4& cur, unsigned (&pH)[4], unsigned (&pL)[4]) const {
        split_pair_f16_scaled(cur[0], invM, cur[1], invM, pH[0], pL[0]);
        split_pair_f16_scaled(cur[2], invM, cur[3], invM, pH[1], pL[1]);
    }
    __device__ __forceinline__ void planes_actions(const float (&xv)[4], unsigned (&pH)[4], unsigned (&pL)[4]) const {
        if (16 * c >= D) pH[2] = pL[2] = 0u;
        else split_pair_f16_scaled(xv[0], sact, xv[1], sact, pH[2], pL[2]);
        if (16 * c + 8 >= D) pH[3] = pL[3] = 0u;
        else split_pair_f16_scaled(xv[2], sact, xv[3], sact, pH[3], pL[3]);
    }
    // the accumulator of output tile c: TileHN::model_step's order for it -- (lo x hi) over the blocks, (hi x lo), (hi x hi)
    __device__ __forceinline__ f32x4 advance(const unsigned (&bH)[NT][4], const unsigned (&bL)[NT][4]) const {
        f32x4 nxt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < NT; ++kb) nxt = mfma_f16_32(mL[kb], bH[kb], nxt);
#pragma unroll
        for (int kb = 0; kb < NT; ++kb) nxt = mfma_f16_32(mH[kb], bL[kb], nxt);
#pragma unroll
        for (int kb = 0; kb < NT; ++kb) nxt = mfma_f16_32(mH[kb], bH[kb], nxt);
        if (KIND == 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) nxt[s] = fast_tanh(nxt[s] * invT) * T;
        }
        return nxt;
    }
    __device__ __forceinline__ void park_to(const f32x4& cur, int off) const {
        f32x4 v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = cur[s] * invT;
        *reinterpret_cast<f32x4*>(row + off + 16 * c + 4 * g) = v;
    }
};

// compile-time loop over term numbers [J0, J1)
template <int J0, int J1, typename F>
__device__ __forceinline__ void hn_for_terms(F&& f) {
    if constexpr (J0 < J1) {
        f(std::integral_constant<int, J0>{});
        hn_for_terms<J0 + 1, J1>(f);
    }
}

// waves of a workgroup: NT model waves, cost wave A (control cost, icem_cost_spec part, the terms behind the first NB, the
// accumulation) and -- where the env has a term list -- cost wave B (the first NB terms: Door's 30-entry slice, half of
// Relocate's norms): with the model on NT waves the term list had become the launch's chain (1.4 us of a Door step against the
// model's 1.0).  B parks its terms' values in LDS; A closes step t one barrier later, adding the values IN THE LIST'S ORDER
// (B's, then its own): cost_step's sum, bit for bit.
template <int H, int D, int O, int KIND, int N32, int N4, int NP>
__global__ __launch_bounds__(64 * (TileHN<H, D, O, KIND, N32, N4, NP>::NT + 1 + (TileHN<H, D, O, KIND, N32, N4, NP>::NB > 0 ? 1 : 0)))
void rollout_hn_split_kernel(HnArgs a) {
    using Tile = TileHN<H, D, O, KIND, N32, N4, NP>;
    using TileC = TileHNc<H, D, O, KIND, N32, N4, NP>;
    using Stream = StreamT<Tile, H, D>;
    constexpr int NT = Tile::NT, TC = Stream::TC, NCH = Stream::NCH, C4 = Stream::C4, CBP = Stream::CBP, VW = Stream::VW, NLD = Stream::NLD;
    constexpr int NB = Tile::NB, NA = Tile::NTERMS - NB;
    static_assert(TC >= 2, "the next chunk is staged one barrier ahead of its first reader");
    constexpr int ROWS = 16 * Tile::RS;
    __shared__ __attribute__((aligned(16))) float stage[2][Stream::STG];
    __shared__ __attribute__((aligned(16))) float rows[2][ROWS];
    __shared__ __attribute__((aligned(16))) uint4 xch[2][NT][2][64];   // [step parity][block][hi | lo][lane]: the operand planes
    __shared__ float tb[2][NB > 0 ? NB : 1][16];                       // [step parity][term][trajectory]: cost wave B's values
    __shared__ unsigned long long wg_keys[2][1][32];
    __shared__ __attribute__((aligned(16))) float obs_stage[Tile::OP];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool model = wave < NT;
    const bool cost_a = wave == NT;
    const float obs_reg = a.r.obs0[((int)threadIdx.x < Tile::OP && (int)threadIdx.x < a.r.o) ? threadIdx.x : 0];
    Tile tile;      // the cost waves' (and the staging layout's): no model planes
    TileC mt;       // a model wave's
    tile.load(a.r, a.A, a.lda, a.B, a.ldb, a.r.o, lane, false);
    tile.set_cost(a.wc, a.cs);
    if (model) mt.load(a.r, a.A, a.lda, a.B, a.ldb, a.r.o, lane, wave);
    Stream stream;
    stream.init(tile, stage[0], lane);
    const int tiles = (a.r.n_rows + 15) / 16;
    if ((int)threadIdx.x < Tile::OP) obs_stage[threadIdx.x] = (int)threadIdx.x < a.r.o ? obs_reg : 0.f;
    __syncthreads();
    tile.load_obs(obs_stage, rows[0]);
    if (model) mt.load_obs(obs_stage, rows[0]);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    for (int tile_id = blockIdx.x; tile_id < tiles; tile_id += gridDim.x) {
        const int row = tile_id * 16 + (lane & 15);
        const bool live = row < a.r.n_rows;
        typename Stream::Vec pre[NLD];
        const typename Stream::Vec* src[NLD];
        const bool stager = wave == 0;
        if (stager) {
#pragma unroll
            for (int m = 0; m < NLD; ++m) {
                const int r = tile_id * 16 + stream.ld_row[m];
                src[m] = reinterpret_cast<const typename Stream::Vec*>(a.r.actions + (size_t)(r < a.r.n_rows ? r : 0) * (H * D)) + stream.ld_c4[m];
                pre[m] = src[m][0];
            }
#pragma unroll
            for (int m = 0; m < NLD; ++m)
                if (stream.ld_on[m]) *reinterpret_cast<typename Stream::Vec*>(&stage[0][Tile::SLACK + stream.ld_row[m] * CBP + VW * stream.ld_c4[m]]) = pre[m];
            if (1 < NCH) {
#pragma unroll
                for (int m = 0; m < NLD; ++m) pre[m] = src[m][C4];
            }
        }
        typename Tile::State st;   // (cost wave A's accumulators)
        tile.init(st);
        f32x4 cur = mt.obs_init;
        float base_prev = 0.f;                 // cost wave A: the step that waits for B's values
        float va_prev[NA > 0 ? NA : 1];
        __syncthreads();   // chunk 0 is staged
        if (model) {       // the start observation -> rows buffer 0; step 0's operand planes -> exchange buffer 0
            mt.park_to(cur, 0);
            unsigned pH[4], pL[4];
            mt.planes(cur, stream.rd0, pH, pL);
            xch[0][wave][0][lane] = uint4{pH[0], pH[1], pH[2], pH[3]};
            xch[0][wave][1][lane] = uint4{pL[0], pL[1], pL[2], pL[3]};
        }
        __syncthreads();
        // cost wave A closes step `tp` (its own share kept in base_prev / va_prev, B's in tb[tp & 1]): the list's order
        auto close_step = [&](int tp) {
            float c_terms = 0.f;
            hn_for_terms<0, NB>([&](auto J) { c_terms += tb[tp & 1][decltype(J)::value][lane & 15]; });
            hn_for_terms<0, NA>([&](auto J) { c_terms += va_prev[decltype(J)::value]; });
            tile.acc_step(st, base_prev + c_terms);
        };
        for (int ch = 0; ch < NCH; ++ch) {
            const int cb = ch & 1;
#pragma unroll
            for (int ts = 0; ts < TC; ++ts) {
                const int t = ch * TC + ts;
                if (model) {
                    if (stager && ts == 0 && ch + 1 < NCH) {   // the next chunk -> the other staging buffer, one barrier ahead of its readers
#pragma unroll
                        for (int m = 0; m < NLD; ++m)
                            if (stream.ld_on[m])
                                *reinterpret_cast<typename Stream::Vec*>(&stage[cb ^ 1][Tile::SLACK + stream.ld_row[m] * CBP + VW * stream.ld_c4[m]]) = pre[m];
                        if (ch + 2 < NCH) {
#pragma unroll
                            for (int m = 0; m < NLD; ++m) pre[m] = src[m][(ch + 2) * C4];
                        }
                    }
                    // the NEXT step's actions ride the same LDS round trip as this step's operand planes (a dependent LDS read
                    // behind the MFMAs cost a lone wave 100 ns of every step)
                    const int t1 = t + 1 < H ? t + 1 : t;
                    const float* rd1 = stream.rd0 + ((t1 / TC) & 1) * Stream::STG + (t1 % TC) * D;
                    float xv1[4];
                    mt.actions_c(rd1, xv1);
                    unsigned bH[NT][4], bL[NT][4];
#pragma unroll
                    for (int kb = 0; kb < NT; ++kb) {
                        const uint4 h4 = xch[t & 1][kb][0][lane], l4 = xch[t & 1][kb][1][lane];
                        bH[kb][0] = h4.x; bH[kb][1] = h4.y; bH[kb][2] = h4.z; bH[kb][3] = h4.w;
                        bL[kb][0] = l4.x; bL[kb][1] = l4.y; bL[kb][2] = l4.z; bL[kb][3] = l4.w;
                    }
                    unsigned pH[4], pL[4];
                    mt.planes_actions(xv1, pH, pL);   // (in front of the MFMAs: nothing of it waits for the state)
                    cur = mt.advance(bH, bL);
                    if (t + 1 < H) {
                        mt.park_to(cur, ((t + 1) & 1) * ROWS);
                        mt.planes_state(cur, pH, pL);
                        xch[t1 & 1][wave][0][lane] = uint4{pH[0], pH[1], pH[2], pH[3]};
                        xch[t1 & 1][wave][1][lane] = uint4{pL[0], pL[1], pL[2], pL[3]};
                    }
                } else if (cost_a) {
                    const float* rd = stream.rd0 + cb * Stream::STG + ts * D;
                    const float* x = tile.row + (t & 1) * ROWS;
                    float xv[NT][4];
                    const float u = tile.actions_of(rd, xv);
                    const float base = tile.base_cost(u, x);
                    float va[NA > 0 ? NA : 1];
                    hn_for_terms<0, NA>([&](auto J) { va[decltype(J)::value] = tile.template term_val<NB + decltype(J)::value>(x); });
                    if constexpr (NB > 0) {
                        if (t > 0) close_step(t - 1);            // B's values of step t - 1 are behind the barrier that opened this step
                        base_prev = base;
                        hn_for_terms<0, NA>([&](auto J) { va_prev[decltype(J)::value] = va[decltype(J)::value]; });
                    } else {                                      // (no second cost wave: the step closes on the spot)
                        float c_terms = 0.f;
                        hn_for_terms<0, NA>([&](auto J) { c_terms += va[decltype(J)::value]; });
                        tile.acc_step(st, base + c_terms);
                    }
                } else {   // cost wave B: the first NB terms of this step's observation
                    const float* x = tile.row + (t & 1) * ROWS;
                    hn_for_terms<0, NB>([&](auto J) {
                        const float v = tile.template term_val<decltype(J)::value>(x);
                        if (lane < 16) tb[t & 1][decltype(J)::value][lane] = v;
                    });
                }
                lds_barrier();
            }
        }
        if constexpr (NB > 0) {
            if (cost_a) close_step(H - 1);
        }
        if (cost_a) {
            const float cost = tile.cost(st);
            if (live && lane < 16) a.r.costs[row] = cost;
            note_nonfinite(a.r, cost, live && lane < 16);
            if (a.r.K > 0) {
                const unsigned long long key = (lane < 16 && live && row < a.r.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
                run_key = topk_push16<false>(run_key, key, first, a.r.K, lane);   // (the per-lane predicate: see bitonic_step)
            }
            first = false;
        }
    }
    // (tail workgroups -- trailing shifted-elite rows that would have opened a second round of tiles -- roll out and store
    //  costs but emit no list: the merge takes those rows through the cost array; a.r.list_wgs = the list-writing workgroups)
    const int n_wg = a.r.list_wgs > 0 ? a.r.list_wgs : (int)gridDim.x;
    if (a.r.K > 0 && (int)blockIdx.x < n_wg) wg_merge_emit<1>(wg_keys, run_key, a.r.K, lane, cost_a ? 0 : 1, a.r, blockIdx.x, n_wg);
}

constexpr int HN_SPLIT_MAX_TILES = FAST_MAX_LISTS + 16;   // one tile per CU (+ a second round for a few shifted-elite rows)
constexpr int HN_PAIR_MAX_TILES = 512;   // two tiles per CU: beyond, four lone waves per CU fill the SIMDs as well

void hn_shape(int n_rows, int* grid, int* waves) {
    r16_shape(n_rows, grid, waves);
    if (*waves > HN_MAX_WAVES) *waves = HN_MAX_WAVES;
}

}  // namespace

bool hn_rollout_supported(int h, int d, int o, int K) {
    if (K > 32) return false;
#define X(HH, DD, OO) \
    if (h == HH && d == DD && o == OO) return true;
    ICEM_HN_SHAPES(X)
#undef X
    return false;
}

// the two-wave-per-tile form: populations of at most HN_PAIR_MAX_TILES tiles (option hn_pair = 0: never -- A/B, tests; read per call)
static bool hn_pair_shape(int n_rows, int* grid, int* pairs) {
    const int tiles = std::max(1, (n_rows + 15) / 16);
    if (!opt_i(OPT_HN_PAIR) || tiles > HN_PAIR_MAX_TILES) return false;
    *pairs = tiles > FAST_MAX_LISTS ? 2 : 1;
    *grid = std::min(FAST_MAX_LISTS, (tiles + *pairs - 1) / *pairs);
    return true;
}

// the split form (NT model waves + a cost wave per tile): at most HN_SPLIT_MAX_TILES tiles (option hn_split = 0: never)
static bool hn_split_shape(int n_rows, int* grid) {
    const int tiles = std::max(1, (n_rows + 15) / 16);
    if (!opt_i(OPT_HN_SPLIT) || !opt_i(OPT_HN_PAIR) || tiles > HN_SPLIT_MAX_TILES) return false;
    *grid = std::min(FAST_MAX_LISTS, tiles);
    return true;
}

// trailing shifted-elite rows of a launch whose sampled rows fill whole tiles: with the split form they get workgroups of
// their own BEHIND the list-writing ones (no list: scored through the cost array) instead of a second round on workgroup 0
int hn_tail_rows(int n_rows, int n_tail) {
    int g;
    if (n_tail <= 0 || n_tail > 64 || n_rows - n_tail <= 0 || (n_rows - n_tail) % 16 != 0) return 0;
    if (!hn_split_shape(n_rows - n_tail, &g) || (n_rows - n_tail) / 16 > FAST_MAX_LISTS) return 0;
    return n_tail;
}

int hn_rollout_lists(int n_rows) {
    int g, w;
    if (hn_split_shape(n_rows, &g)) return g;
    if (hn_pair_shape(n_rows, &g, &w)) return g;
    hn_shape(n_rows, &g, &w);
    return g;
}

// the compiled term programs (N32, N4, NP): none, FetchPickAndPlace's two 3-entry norms, Relocate's four + its lift bonus,
// Door's 30-entry velocity term + palm-handle norm + hinge offset and three opening bonuses; a handle's list takes the
// first program that holds it (hn_cost_program)
bool hn_cost_program(int n32, int n4, int np, int* prog) {
    static const int P[4][3] = {{0, 0, 0}, {0, 2, 0}, {0, 4, 1}, {1, 1, 4}};
    for (int p = 0; p < 4; ++p)
        if (n32 <= P[p][0] && n4 <= P[p][1] && np <= P[p][2]) {
            if (prog) { prog[0] = P[p][0]; prog[1] = P[p][1]; prog[2] = P[p][2]; }
            return true;
        }
    return false;
}

void launch_rollout_hn(const FastRolloutArgs& r, int h, int d, int o, int kind, const float* A, int lda, const float* B, int ldb,
                       int lin_idx, int flip_idx, const CostArgs<float>* cs, const int* prog, hipStream_t st) {
    HnArgs a{r, A, B, lda, ldb, WideCost{lin_idx, flip_idx, r.ctrl_w, r.lin_w, r.flip_pen, r.flip_th}, cs};
    int grid, waves;
    if (hn_split_shape(r.n_rows, &grid)) {
        if (r.list_wgs > 0) grid = (r.n_rows + 15) / 16;   // list workgroups + the tail's (hn_tail_rows)
#define SP(HH, DD, OO, KK, A32, A4, AP)                                                                                             \
    if (prog[0] == A32 && prog[1] == A4 && prog[2] == AP) {                                                                         \
        constexpr int NTV = (OO + 15) / 16;                                                                                         \
        constexpr int NWV = NTV + 1 + ((A32 > 0 ? A32 : A4 / 2) > 0 ? 1 : 0);   /* model waves + cost wave A (+ B) */                \
        hipLaunchKernelGGL((rollout_hn_split_kernel<HH, DD, OO, KK, A32, A4, AP>), dim3(grid), dim3(64 * NWV), 0, st, a);           \
        return;                                                                                                                     \
    }
#define SK(HH, DD, OO, KK) SP(HH, DD, OO, KK, 0, 0, 0) SP(HH, DD, OO, KK, 0, 2, 0) SP(HH, DD, OO, KK, 0, 4, 1) SP(HH, DD, OO, KK, 1, 1, 4)
#define SR(HH, DD, OO)                   \
    if (h == HH && d == DD && o == OO) { \
        if (kind == 1) {                 \
            SK(HH, DD, OO, 1)            \
        } else {                         \
            SK(HH, DD, OO, 0)            \
        }                                \
        return;                          \
    }
        ICEM_HN_SHAPES(SR)
#undef SR
#undef SK
#undef SP
        return;
    }
    if (hn_pair_shape(r.n_rows, &grid, &waves)) {
        const int pairs = waves;
#define PP(HH, DD, OO, KK, PPV, A32, A4, AP)                                                                                       \
    if (prog[0] == A32 && prog[1] == A4 && prog[2] == AP) {                                                                        \
        hipLaunchKernelGGL((rollout_hn_pair_kernel<HH, DD, OO, KK, PPV, A32, A4, AP>), dim3(grid), dim3(128 * PPV), 0, st, a);     \
        return;                                                                                                                    \
    }
#define PK(HH, DD, OO, KK, PPV) PP(HH, DD, OO, KK, PPV, 0, 0, 0) PP(HH, DD, OO, KK, PPV, 0, 2, 0) PP(HH, DD, OO, KK, PPV, 0, 4, 1) PP(HH, DD, OO, KK, PPV, 1, 1, 4)
#define PW(HH, DD, OO, PPV)                 \
    if (pairs == PPV) {                     \
        if (kind == 1) {                    \
            PK(HH, DD, OO, 1, PPV)          \
        } else {                            \
            PK(HH, DD, OO, 0, PPV)          \
        }                                   \
        return;                             \
    }
#define PR(HH, DD, OO)                   \
    if (h == HH && d == DD && o == OO) { \
        PW(HH, DD, OO, 1)                \
        PW(HH, DD, OO, 2)                \
    }
        ICEM_HN_SHAPES(PR)
#undef PR
#undef PW
#undef PK
#undef PP
        return;
    }
    hn_shape(r.n_rows, &grid, &waves);
#define XP(HH, DD, OO, KK, WW, A32, A4, AP)                                                                                   \
    if (prog[0] == A32 && prog[1] == A4 && prog[2] == AP) {                                                                   \
        hipLaunchKernelGGL((rollout_hn_kernel<HH, DD, OO, KK, WW, A32, A4, AP>), dim3(grid), dim3(64 * WW), 0, st, a);        \
        return;                                                                                                               \
    }
#define XK(HH, DD, OO, KK, WW) XP(HH, DD, OO, KK, WW, 0, 0, 0) XP(HH, DD, OO, KK, WW, 0, 2, 0) XP(HH, DD, OO, KK, WW, 0, 4, 1) XP(HH, DD, OO, KK, WW, 1, 1, 4)
#define XW(HH, DD, OO, WW)                  \
    if (waves == WW) {                      \
        if (kind == 1) {                    \
            XK(HH, DD, OO, 1, WW)           \
        } else {                            \
            XK(HH, DD, OO, 0, WW)           \
        }                                   \
        return;                             \
    }
#define XR(HH, DD, OO)                   \
    if (h == HH && d == DD && o == OO) { \
        XW(HH, DD, OO, 1)                \
        XW(HH, DD, OO, 2)                \
        XW(HH, DD, OO, 4)                \
    }
    ICEM_HN_SHAPES(XR)
#undef XR
#undef XW
#undef XK
#undef XP
}

}  // namespace icem
