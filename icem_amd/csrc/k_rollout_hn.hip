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
    __device__ __forceinline__ void load(const FastRolloutArgs& a, const float* A, int lda, const float* B, int ldb, int o, int lane) {
        const int j = lane & 15;
        g = lane >> 4;
        sM = a.m_scale;
        sB = a.b_scale;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
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
    __device__ __forceinline__ void step(State& st, const float* rd) const {
        // this step's action entries (entry 16 kb + 4 q + g at rd[16 kb + 4 q]) and the control cost's share
        float xv[NT][4];
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
        // (the same block BEHIND the MFMAs -- in their shadow, the accumulators are not needed before the next step -- measured
        //  slower: 71 instead of 57.5 us per Door launch at N = 4096)
        // the pre-action observation in the row; the cost: the four lanes' shares (control cost, lane group 0's icem_cost_spec
        // terms) summed by one reduction, the term list on top (the same value in all four lanes)
        park(st);
        float c = u * ctrl_w;
        {
            const float ang = row[flip_i];
            c += (ang > wc.flip_th) ? pen_g : 0.f;
            c += (ang < -wc.flip_th) ? pen_g : 0.f;
            c = __builtin_fmaf(lin_g, row[wc.lin_idx], c);
        }
        c = reduce_groups(c) + terms_all(row);
        st.acc_s = __builtin_fmaf(st.acc_s, ksum, c);
        st.acc_b = (c < st.acc_b || c != c) ? c : st.acc_b;   // np.amin: a NaN step cost makes the trajectory's cost NaN
        // B operand planes per contraction block: own columns (x 1 / sM: from the accumulators' scale to the operand's), actions
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

int hn_rollout_lists(int n_rows) {
    int g, w;
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
