// k_rollout_wide.hip -- K2 + K3 for WIDE observations (32 < o <= 384, e.g. HumanoidStandup's real o = 378, d = 17:
// icem/environments/mujoco.py:241-277): rollout_wide_kernel.
//
// At this width the model step is a real GEMM: per step [n, o + d] x [o + d, o], 2 (o + d) o = 299 kflop per
// trajectory-step at o = 378 against 68 bytes of actions -- compute bound by three orders of magnitude (SURVEY
// 7.3-11), so the kernel is built around keeping the f32 matrix pipe fed, not around HBM:
//   * one wavefront owns 16 trajectories; their contraction vectors [obs | action | 0-pad] live in the wave's own LDS
//     rows X[16][XS] (f32; 25.6 KB at o = 378), the new observation of all NT = ceil(o / 16) column tiles in 4 * NT
//     accumulator registers (96 at o = 378);
//   * per 4-wide contraction block kb: ONE ds_read of the B operand (X[j][4 kb + g]: trajectory j = lane % 16, slot
//     g = lane / 16), then NT v_mfma_f32_16x16x4_f32 (exact f32, same fmaf chain as scalar code) on independent
//     accumulators, their A operands (the 16 x 4 blocks of the model) coming as 16-byte loads from a host-packed copy
//     of [A ; B] in exactly that order: Mp[kb][ct / 4][lane][ct % 4] = M[4 kb + lane / 16][16 ct + lane % 16] -- 6
//     coalesced dwordx4 loads per 24 MFMAs, shared by every wave of the chip through L2 (the 608 KB model stays
//     L2-resident);
//   * after the last block lane (j, g) holds columns 16 ct + 4 g .. + 3 of trajectory j in accumulator ct: activation,
//     then one 16-byte LDS store per tile back into X (the wave's LDS traffic executes in order: no barrier);
//   * the step cost (icem_cost_spec: control cost + linear + flip terms of the PRE-action observation) is read from X
//     by lanes 0..15; costs and the wave's running sorted top-K as in k_rollout.hip; one candidate list per workgroup.
// The arithmetic is exact f32, so the tolerance against the float64 oracle is the 1e-5 of every other f32 kernel.
// (A bf16 x 3-plane split on v_mfma_f32_16x16x32_bf16 would need 94 B/clk of model operands per CU at this tiling --
// more than the L2 port delivers: DESIGN.md section 4.)
#include "fused_dev.h"
#include "wide_dev.h"

namespace icem {

namespace {

constexpr int WIDE_WAVES = 4;

// EXT: icem_cost_terms on.  Its own instantiation: the model loop below is scheduled to the register -- with the terms'
// code merely PRESENT (never executed) the allocator moved an in-flight model operand, which is a wait for all requests
// in flight, and every launch was 9 % slower (1 358 -> 1 466 us at o = 378, N = 16 384), whatever was tried to keep the
// terms out of the loop's live set.
template <int NT, int KIND, int WAVES, bool EXT>
__global__ __launch_bounds__(64 * WAVES) void rollout_wide_kernel(WideRolloutArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs_all[];  // [WAVES][16][XS]
    __shared__ unsigned long long wg_keys[2][WAVES][32];
    constexpr int NQ = NT / 4;  // 16-byte model loads per contraction block
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 15, g = lane >> 4;
    const int XS = a.xs, KB = a.kb, o = a.o, d = a.d, H = a.h;
    float* X = xs_all + (size_t)wave * 16 * XS;
    const float4* __restrict__ Mp = reinterpret_cast<const float4*>(a.Mp);
    const WideCost wc{a.lin_idx, a.flip_idx, a.ctrl_w, a.lin_w, a.flip_pen, a.flip_th};
    __shared__ CostArgs<float> cs_s;
    __shared__ float park[WAVES][32];
    constexpr bool ext = EXT;
    if (EXT) wide_stage_terms(cs_s, a.cs, threadIdx.x, 64 * WAVES);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    const int tiles = (a.n_rows + 15) / 16;
    for (int tile = wave * gridDim.x + blockIdx.x; tile < tiles; tile += WAVES * gridDim.x) {
        const int row0 = tile * 16;
        // contraction vectors: the start observation in every row, zeros behind
        for (int e = lane; e < 16 * XS; e += 64) {
            const int c = e % XS;
            X[e] = c < o ? a.obs0[c] : 0.f;
        }
        float acc_c = 0.f;
        // Narrow observations (NT <= 8: a step is a few hundred cycles): the NEXT step's actions are requested while this step
        // runs and held in registers -- a load -> LDS store per step would expose a global round trip that outweighs the step.
        // (At NT = 24 a step is 40 us and the kernel has no register to spare: EXPERIMENTS.md R3.12.)
        constexpr int AE = NT <= 8 ? 4 : 0;   // elements per lane held ahead: d <= 16
        const bool ahead = AE > 0 && 16 * d <= 64 * AE;
        float an[AE > 0 ? AE : 1];
        int aoff[AE > 0 ? AE : 1], xoff[AE > 0 ? AE : 1];
        if (ahead) {
#pragma unroll
            for (int i = 0; i < AE; ++i) {
                const int e = lane + 64 * i;
                const int r = e / d, c = e - r * d;
                const bool in = e < 16 * d;
                xoff[i] = in ? r * XS + o + c : -1;
                aoff[i] = (in && row0 + r < a.n_rows) ? (r * H) * d + c : -1;
            }
        }
        const float* abase = a.actions + (size_t)row0 * H * d;
        auto request_actions = [&](int t) {
#pragma unroll
            for (int i = 0; i < AE; ++i) an[i] = aoff[i] >= 0 ? abase[aoff[i] + t * d] : 0.f;
        };
        if (ahead) request_actions(0);
        for (int t = 0; t < H; ++t) {
            // this step's actions -> X[:, o .. o + d)
            if (ahead) {
#pragma unroll
                for (int i = 0; i < AE; ++i)
                    if (xoff[i] >= 0) X[xoff[i]] = an[i];
                if (t + 1 < H) request_actions(t + 1);
            } else {
                for (int e = lane; e < 16 * d; e += 64) {
                    const int r = e / d, c = e - r * d;
                    const int row = row0 + r;
                    X[r * XS + o + c] = row < a.n_rows ? a.actions[((size_t)row * H + t) * d + c] : 0.f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // step cost of trajectory `lane` (lanes 0..15) from the pre-action observation; with cost terms that need the
            // whole row (finite check / state box) lane (j, g) sweeps entries g, g + 4, .. of row j first
            bool bad = false;
            if (ext && cs_s.health_idx >= 0) {
                const float* xj = X + j * XS;
                bool b = false;
                for (int k = g; k < o; k += 4) b |= wide_bad_entry(cs_s, xj[k], k);
                const unsigned long long m = __ballot(b);
                bad = ((m >> (lane & 15)) & 0x0001000100010001ull) != 0ull;
            }
            // (nothing of the cost may stay in registers across the model loop below: two more live values and the
            // register allocator moves an in-flight model operand, i.e. waits for ALL requests -- 9 % of a launch.  With a
            // difference term the step's partial cost and obs[diff_idx] are parked in LDS instead.)
            const bool diff = ext && cs_s.diff_idx >= 0;
            {
                float c_step = 0.f, dold = 0.f;
                if (ext && NT <= 8) {   // narrow widths: the step is short, the term list walked by one lane is not
                    c_step = reduce_groups(wide_step_cost_lanes(wc, cs_s, X + j * XS, o, d, bad, g, dold));
                } else if (lane < 16) {
                    c_step = wide_step_cost(wc, ext, cs_s, X + lane * XS, o, d, bad, dold);
                }
                if (diff) {
                    if (lane < 16) { park[wave][lane] = c_step; park[wave][16 + lane] = dold; }
                } else {
                    acc_c = wide_accumulate(acc_c, c_step, t, a.cost_mode);
                }
            }
            f32x4 acc[NT];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* xb = X + j * XS + g;
            // model blocks are requested ahead of the MFMAs that consume them: named register sets, the loop unrolled,
            // and scheduling barriers so that the requests stay where they are written (left alone the compiler sinks
            // them next to their first use and every block pays an L2 round trip: 2.19 instead of 1.70 ms per launch)
            float4 mA[NQ], mB[NQ], mC[NQ];
            float bA, bB, bC;
            auto request = [&](float4 (&m)[NQ], float& b, int kb) {
                kb = kb < KB ? kb : KB - 1;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) m[q] = Mp[((size_t)kb * NQ + q) * 64 + lane];
                b = xb[4 * kb];
                __builtin_amdgcn_sched_barrier(0);
            };
            auto block = [&](const float4 (&m)[NQ], float b) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[4 * q + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[q].x, b, acc[4 * q + 0], 0, 0, 0);
                    acc[4 * q + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[q].y, b, acc[4 * q + 1], 0, 0, 0);
                    acc[4 * q + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[q].z, b, acc[4 * q + 2], 0, 0, 0);
                    acc[4 * q + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(m[q].w, b, acc[4 * q + 3], 0, 0, 0);
                }
            };
            // TWO blocks in flight ahead of the one being multiplied: with four waves per CU each pulling 6 KB per block
            // through the CU's vector-memory port a request comes back after well over the 24 x 32 = 768 cycles a
            // block's MFMAs take -- one block ahead left part of every round trip exposed (1.97 ms per launch, this:
            // 1.67).  Three register sets, the loop unrolled by three and branch-free (KB is a multiple of 3: wide_kb
            // pads the model with zero blocks, which add exact zeros); the accumulators stay in AGPRs throughout.
            request(mA, bA, 0);
            request(mB, bB, 1);
#pragma unroll 1
            for (int kb = 0; kb < KB; kb += 3) {
                request(mC, bC, kb + 2);
                block(mA, bA);
                request(mA, bA, kb + 3);
                block(mB, bB);
                request(mB, bB, kb + 4);
                block(mC, bC);
            }
            // new observation: lane (j, g) holds columns 16 ct + 4 g .. + 3 of trajectory j (columns >= o: the model's
            // zero padding, they stay 0 for the linear model and tanh(0) = 0)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                f32x4 v = acc[ct];
                if (KIND == 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fast_tanh(v[k]);
                }
                const int col = 16 * ct + 4 * g;
                if (col < o) *reinterpret_cast<f32x4*>(X + j * XS + col) = v;
            }
            // (a last column group that straddles o also zeroes the first action slots: they are reloaded next step)
            if (diff) {   // the term that reads the observation just written
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                float c_step = 0.f;
                if (lane < 16) c_step = park[wave][lane] + wide_diff_cost(cs_s, X[lane * XS + cs_s.diff_idx], park[wave][16 + lane]);
                acc_c = wide_accumulate(acc_c, c_step, t, a.cost_mode);
            }
        }
        const float cost = acc_c;
        const int row = row0 + (lane & 15);
        const bool live = row < a.n_rows;
        if (live && lane < 16) a.costs[row] = cost;
        if (a.K > 0) {
            const unsigned long long key = (lane < 16 && live && row < a.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
            run_key = topk_push16(run_key, key, first, a.K, lane);
            first = false;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (a.K > 0) {
        FastRolloutArgs fr{};  // wg_merge_emit only looks at the candidate outputs
        fr.part_k = a.part_k;
        fr.part_c = a.part_c;
        fr.part_i = a.part_i;
        wg_merge_emit<WAVES>(wg_keys, run_key, a.K, lane, wave, fr);
    }
}

// A few single rows at the same widths: one workgroup per trajectory, thread c owns observation column c and runs the
// contraction as ONE fmaf chain over k = 0 .. o + d - 1 from the row-major model -- the chain an f32 MFMA executes over
// its slots, blocks in rollout_wide_kernel's order, so the costs are the bits the tile kernel would produce for the row
// (the zero padding blocks add exact zeros).  For the handful of shifted elites (icem.py:131-137) that would otherwise
// open a 16-row tile of their own: at N = 16 384 that tile is number 1 025 on 1 024 wavefront slots and doubles the
// launch (2.47 instead of 1.26 ms); these rows take ~0.1 ms.
struct WideRowsArgs {
    int row0, n_tail, o, d, h, cost_mode;
    WideCost wc;
    const CostArgs<float>* cs;
    const float* A;  // [o, o] row-major
    const float* B;  // [d, o]
    const float* obs0;
    const float* actions;
    float* costs;
};

template <int KIND>
__global__ __launch_bounds__(384) void rollout_rows_wide_kernel(WideRowsArgs a) {
    __shared__ float x[2][448];  // [obs | action], double buffered over the steps
    const int c = threadIdx.x, o = a.o, d = a.d, H = a.h;
    const int row = a.row0 + blockIdx.x;
    x[0][c] = c < o ? a.obs0[c] : 0.f;
    __shared__ CostArgs<float> cs_s;
    const bool ext = a.cs != nullptr;
    wide_stage_terms(cs_s, a.cs, threadIdx.x, 384);
    const bool sweep = ext && cs_s.health_idx >= 0, diff = ext && cs_s.diff_idx >= 0;
    float acc_c = 0.f, c_step = 0.f, dold = 0.f;
    for (int t = 0; t < H; ++t) {
        float* xc = x[t & 1];
        if (c < d) xc[o + c] = a.actions[((size_t)row * H + t) * d + c];
        __syncthreads();
        if (c == 0 && t > 0) {   // the previous step's cost is complete now that its next observation is visible
            if (diff) c_step += wide_diff_cost(cs_s, xc[cs_s.diff_idx], dold);
            acc_c = wide_accumulate(acc_c, c_step, t - 1, a.cost_mode);
        }
        bool bad = false;
        if (sweep) bad = __syncthreads_or(c < o && wide_bad_entry(cs_s, xc[c], c)) != 0;
        if (c == 0) c_step = wide_step_cost(a.wc, ext, cs_s, xc, o, d, bad, dold);   // rollout_wide_kernel's expression
        if (c < o) {
            float acc = 0.f;
            const float* Ac = a.A + c;
#pragma unroll 8
            for (int k = 0; k < o; ++k) acc = __builtin_fmaf(Ac[(size_t)k * o], xc[k], acc);
            const float* Bc = a.B + c;
#pragma unroll 4
            for (int k = 0; k < d; ++k) acc = __builtin_fmaf(Bc[(size_t)k * o], xc[o + k], acc);
            x[(t & 1) ^ 1][c] = KIND == 1 ? fast_tanh(acc) : acc;
        }
    }
    __syncthreads();
    if (c == 0) {
        if (diff) c_step += wide_diff_cost(cs_s, x[H & 1][cs_s.diff_idx], dold);
        a.costs[row] = wide_accumulate(acc_c, c_step, H - 1, a.cost_mode);
    }
}

}  // namespace

bool wide_rollout_supported(int o, int d, int K) { return o > 32 && o <= 384 && d >= 1 && d <= 64 && K <= 32; }
bool gemm_rollout_supported(int o, int d, int K) { return o >= 1 && o <= 384 && d >= 1 && d <= 64 && K <= 32; }

int wide_rollout_lists(int n_rows) { return std::min(std::max(1, (n_rows + 15) / 16), FAST_MAX_LISTS); }

// contraction rows of the packed model: [obs (o) | act (d)] padded to whole 4-blocks; X row stride in floats
// (a multiple of 3: rollout_wide_kernel's block loop is unrolled by three; the padding blocks are zero)
int wide_kb(int o, int d) { return ((o + d + 3) / 4 + 2) / 3 * 3; }
int wide_xs(int o, int d) { return 4 * wide_kb(o, d) + 4; }
int wide_nt(int o) { const int nt = (o + 15) / 16; return nt <= 4 ? 4 : nt <= 8 ? 8 : nt <= 16 ? 16 : 24; }

// Mp[kb][q][lane][v] = M[4 kb + lane / 16][16 (4 q + v) + lane % 16], M = [A ; B] ([o + d, o], zero padded)
void pack_wide_model(int o, int d, const double* A, const double* B, std::vector<float>& Mp) {
    const int KB = wide_kb(o, d), NT = wide_nt(o), NQ = NT / 4;
    Mp.assign((size_t)KB * NQ * 64 * 4, 0.f);
    auto M = [&](int r, int c) -> double {
        if (c >= o) return 0.0;
        if (r < o) return A[(size_t)r * o + c];
        if (r < o + d) return B[(size_t)(r - o) * o + c];
        return 0.0;
    };
    for (int kb = 0; kb < KB; ++kb)
        for (int q = 0; q < NQ; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int v = 0; v < 4; ++v)
                    Mp[(((size_t)kb * NQ + q) * 64 + lane) * 4 + v] = (float)M(4 * kb + lane / 16, 16 * (4 * q + v) + lane % 16);
}

void launch_rollout_wide(const WideRolloutArgs& a, int kind, hipStream_t st) {
    const int grid = wide_rollout_lists(a.n_rows);
    const size_t lds = (size_t)WIDE_WAVES * 16 * a.xs * sizeof(float);
    const int NT = wide_nt(a.o);
#define XW1(NTV, KINDV, EXTV)                                                                                         \
    {                                                                                                                 \
        auto kfn = rollout_wide_kernel<NTV, KINDV, WIDE_WAVES, EXTV>;                                                 \
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * WIDE_WAVES), lds, st, a);                                       \
    }
#define XW(NTV)                                                                                                       \
    if (NT == NTV) {                                                                                                  \
        if (kind == 1) {                                                                                              \
            if (a.cs) XW1(NTV, 1, true) else XW1(NTV, 1, false)                                                       \
        } else {                                                                                                      \
            if (a.cs) XW1(NTV, 0, true) else XW1(NTV, 0, false)                                                       \
        }                                                                                                             \
        return;                                                                                                       \
    }
    XW(4) XW(8) XW(16) XW(24)
#undef XW
#undef XW1
}

void launch_rollout_rows_wide(const WideRolloutArgs& w, int row0, int n_tail, const float* A, const float* B, int kind,
                              hipStream_t st) {
    if (n_tail <= 0) return;
    WideRowsArgs a{row0, n_tail, w.o, w.d, w.h, w.cost_mode, WideCost{w.lin_idx, w.flip_idx, w.ctrl_w, w.lin_w, w.flip_pen, w.flip_th},
                   w.cs, A, B, w.obs0, w.actions, w.costs};
    if (kind == 1)
        hipLaunchKernelGGL((rollout_rows_wide_kernel<1>), dim3(n_tail), dim3(384), 0, st, a);
    else
        hipLaunchKernelGGL((rollout_rows_wide_kernel<0>), dim3(n_tail), dim3(384), 0, st, a);
}

}  // namespace icem
