// See icem_rssm.h.  One workgroup (8 wavefronts) per 16 trajectories (two such tiles from N = 8192 on).  Every layer is D = W-block . X^T with
// v_mfma_f32_16x16x32_bf16: the A operand is a 16 x 32 block of the weight (output row i = lane % 16, k = 8 * (lane / 16)
// + 0..7), the B operand the activations (trajectory j = lane % 16, same k), and the result leaves lane (j, g) holding
// outputs 4g .. 4g+3 of trajectory j -- exactly the slice that lane writes back (bias, activation, bf16) to the LDS
// activation row it will later be read from as a B operand.  The waves split a layer's output blocks; the
// weights stream from L2 (0.9 MB of padded bf16 per step, the same for all workgroups), the recurrent state stays in LDS
// in f32.
#include "rssm_dev.h"

namespace icem {
namespace {
using namespace rssm_dev;


// TT tiles of 16 trajectories per workgroup: every weight block a wave requests feeds TT MFMAs (large populations are
// bound by re-streaming the weights from L2 per tile; small ones want as many workgroups as possible: TT = 1)
template <int TT>
__global__ __launch_bounds__(NTHR) void rssm_rollout_kernel(int n, int horizon, int cost_mode, const unsigned short* __restrict__ Pg,
                                                           const float* __restrict__ obs0, const float* __restrict__ actions,
                                                           float* __restrict__ costs) {
    constexpr int NR = 16 * TT;   // trajectories (LDS rows) of the workgroup
    __shared__ __attribute__((aligned(16))) unsigned short zA[NR * ZS];       // [z_t | a_t]
    __shared__ __attribute__((aligned(16))) unsigned short hb[2][NR * RS];    // h_t in bf16 (operand), ping-pong
    __shared__ __attribute__((aligned(16))) unsigned short xb[NR * RS];       // x, then p
    __shared__ __attribute__((aligned(16))) unsigned short r1[NR * RS];
    __shared__ __attribute__((aligned(16))) unsigned short r2[NR * RS];
    __shared__ __attribute__((aligned(16))) float h32[NR * HS];               // h_t in f32 (the recurrence)
    __shared__ __attribute__((aligned(16))) float bs[NBIAS];                  // all biases
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: block addresses stay scalar
    const int j = lane & 15, g = lane >> 4;
    const int base = blockIdx.x * NR;
    // ---- initial state: every trajectory starts from obs0 = [h (200) | z (30)] ----
    for (int e = tid; e < NR * RS; e += NTHR) {
        const int k = e % RS;
        const float v = k < DET ? obs0[k] : 0.f;
        hb[0][e] = to_bf16(v);
        hb[1][e] = 0;
        xb[e] = r1[e] = r2[e] = 0;
    }
    for (int e = tid; e < NR * HS; e += NTHR) h32[e] = (e % HS) < DET ? obs0[e % HS] : 0.f;
    {
        constexpr size_t offs[8] = {B1, BGI, BGH, B4, B5, B6, B7, B8};
#pragma unroll
        for (int l = 0; l < 8; ++l)
            for (int e = tid; e < bias_len(offs[l]); e += NTHR)
                bs[bias_slot(offs[l]) + e] = reinterpret_cast<const float*>(Pg + offs[l])[e];
    }
    for (int e = tid; e < NR * ZS; e += NTHR) {
        const int k = e % ZS, jj = e / ZS;
        float v = 0.f;
        if (k < STOCH) v = obs0[DET + k];
        else if (k >= 32 && k < 32 + ACT) v = actions[(size_t)(base + jj < n ? base + jj : n - 1) * horizon * ACT + (k - 32)];
        zA[e] = to_bf16(v);   // padding trajectories repeat the last one, never stored
    }
    __syncthreads();
    const int xr = j * RS + 8 * g, zr = j * ZS + 8 * g;   // operand reads: 8 bf16 of a 32-wide k-block
    const int xo = j * RS + 4 * g, zo = j * ZS + 4 * g, ho = j * HS + 4 * g;   // result writes: 4 outputs of a 16-wide block
    float acc_cost[TT];
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) acc_cost[tt] = 0.f;
    int cur = 0;
    // The first layers' weight blocks of a step are requested at the end of the previous one (in front of its last
    // barrier), W7's during phase 1 and W5's during phase 3: three of the four barriers of a step no longer have an L2
    // round trip behind them.
    // (Only with one tile per workgroup: with two, the extra live operands spill and the prefetch loses.)
    constexpr bool AHEAD = TT == 1;
    v4i Ah[NOB][DETK], Az[NOB][STK], A1[NOB][K1K];
    if (AHEAD) {
        gptr Plane = (gptr)Pg + lane * 8;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NOB; ++i) {
            gptr W = Plane + W6 + (size_t)own_block(w, i) * K6K * BLK;
            request<DETK>(W, Ah[i]);
            request<STK>(W + (size_t)DETK * BLK, Az[i]);
        }
        req_own<K1K>(Plane, W1, w, A1);
    }
    // reward = W8 r2 + b8 of the state step t starts from, by the wave that owns one block less than wave 0
    auto score = [&](gptr Plane, int t) {
        v4i A8[HIDK];
        request<HIDK>(Plane + W8, A8);
        const v4f b8 = bias4(bs, B8, 4 * g);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const v4f a = mma<HIDK>(A8, r2 + tt * 16 * RS + xr, b8);
            const float c = -a[0];   // output 0 of trajectory j lives in lane (j, g = 0), register 0
            if (t == 0 || cost_mode == 2) acc_cost[tt] = c;
            else if (cost_mode == 0) acc_cost[tt] += c;
            else acc_cost[tt] = (c < acc_cost[tt] || c != c) ? c : acc_cost[tt];
        }
    };
    // horizon - 1 full steps; the state the LAST step starts from is only scored (the transition behind it never is)
    for (int t = 0; t + 1 < horizon; ++t) {
        // the parameters do not depend on t, and the optimizer would gladly keep whatever fits of them in registers
        // across steps (it filled all 512 and spilled): re-derive the pointer behind an opaque barrier every step
        gptr P = (gptr)Pg;
        asm volatile("" : "+s"(P));
        gptr Plane = P + lane * 8;   // this lane's 8 bf16 inside every A-operand block
        // ---- phase 1 (reads h_t, z_t, a_t): r1 = relu(W6 [h | z] + b6) and x = relu(W1 [z | a] + b1) ----
        if (!AHEAD) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NOB; ++i) {
                gptr W = Plane + W6 + (size_t)own_block(w, i) * K6K * BLK;
                request<DETK>(W, Ah[i]);
                request<STK>(W + (size_t)DETK * BLK, Az[i]);
            }
            req_own<K1K>(Plane, W1, w, A1);
        }
        v4i A7[NOB][HIDK];
        if (AHEAD) req_own<HIDK>(Plane, W7, w, A7);   // next phase's first layer: in flight across the barrier
#pragma unroll
        for (int i = 0; i < NOB; ++i) {
            const int ob = own_block(w, i);
            const v4f b = bias4(bs, B6, ob * 16 + 4 * g);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                v4f a = mma<DETK>(Ah[i], hb[cur] + tt * 16 * RS + xr, b);
                a = mma<STK>(Az[i], zA + tt * 16 * ZS + zr, a);
                if (w + WAVES * i < 13) *reinterpret_cast<v4s*>(r1 + tt * 16 * RS + xo + ob * 16) = relu_pack(a);
            }
        }
        fin_dense<K1K, TT>(bs, B1, A1, zA + zr, 16 * ZS, xb + xo, w, g);
        __syncthreads();
        // ---- phase 2: r2 = relu(W7 r1 + b7);  GRU h' = (1 - u) n + u h, one output block at a time ----
        if (!AHEAD) req_own<HIDK>(Plane, W7, w, A7);
        fin_dense<HIDK, TT>(bs, B7, A7, r1 + xr, 16 * RS, r2 + xo, w, g);
#pragma unroll 1
        for (int i = 0; i < NOB; ++i) {
            const int ob = own_block(w, i), bi = ob * 16 + 4 * g;
            // all six gate matrices of this output block: 78 8-byte loads per lane in flight
            v4i Ar[HIDK], Au[HIDK], An[HIDK], Br[DETK], Bu[DETK], Bn[DETK];
            __builtin_amdgcn_sched_barrier(0);
            request<HIDK>(Plane + WGI + (size_t)(ob) * HIDK * BLK, Ar);
            request<HIDK>(Plane + WGI + (size_t)(DETB + ob) * HIDK * BLK, Au);
            request<HIDK>(Plane + WGI + (size_t)(2 * DETB + ob) * HIDK * BLK, An);
            request<DETK>(Plane + WGH + (size_t)(ob) * DETK * BLK, Br);
            request<DETK>(Plane + WGH + (size_t)(DETB + ob) * DETK * BLK, Bu);
            request<DETK>(Plane + WGH + (size_t)(2 * DETB + ob) * DETK * BLK, Bn);
            __builtin_amdgcn_sched_barrier(0);
            const v4f bir = bias4(bs, BGI, bi), biu = bias4(bs, BGI, 16 * DETB + bi), bin = bias4(bs, BGI, 32 * DETB + bi);
            const v4f bhr = bias4(bs, BGH, bi), bhu = bias4(bs, BGH, 16 * DETB + bi), bhn = bias4(bs, BGH, 32 * DETB + bi);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const unsigned short* X = xb + tt * 16 * RS + xr;
                const unsigned short* H = hb[cur] + tt * 16 * RS + xr;
                const v4f ir = mma<HIDK>(Ar, X, bir), iu = mma<HIDK>(Au, X, biu), in = mma<HIDK>(An, X, bin);
                const v4f hr = mma<DETK>(Br, H, bhr), hu = mma<DETK>(Bu, H, bhu), hn = mma<DETK>(Bn, H, bhn);
                if (w + WAVES * i < 13) {
                    float* hp = h32 + tt * 16 * HS + ho + ob * 16;
                    float nh[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        nh[r] = gru_out(ir[r], iu[r], in[r], hr[r], hu[r], hn[r], hp[r]);
                        hp[r] = nh[r];
                    }
                    *reinterpret_cast<v4s*>(hb[cur ^ 1] + tt * 16 * RS + xo + ob * 16) = pack4(nh[0], nh[1], nh[2], nh[3]);
                }
            }
        }
        __syncthreads();
        // ---- phase 3: reward = W8 r2 + b8 (one wave);  p = relu(W4 h' + b4) ----
        v4i A5[HIDK];
        if (AHEAD && w < STB) request<HIDK>(Plane + W5 + (size_t)w * HIDK * BLK, A5);   // next phase's layer, across the barrier
        __builtin_amdgcn_sched_barrier(0);
        if (w == WAVES - 1) score(Plane, t);
        {
            v4i A4[NOB][DETK];
            req_own<DETK>(Plane, W4, w, A4);
            fin_dense<DETK, TT>(bs, B4, A4, hb[cur ^ 1] + xr, 16 * RS, xb + xo, w, g);
        }
        __syncthreads();
        // ---- phase 4: z' = W5 p + b5 (waves 0, 1) and the next action (wave 2) -> [z | a] ----
        if (AHEAD) {   // the next step's first layers (always: a request that is always made leaves nothing to keep alive)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NOB; ++i) {
                gptr W = Plane + W6 + (size_t)own_block(w, i) * K6K * BLK;
                request<DETK>(W, Ah[i]);
                request<STK>(W + (size_t)DETK * BLK, Az[i]);
            }
            req_own<K1K>(Plane, W1, w, A1);
        }
        if (w < STB) {
            if (!AHEAD) request<HIDK>(Plane + W5 + (size_t)w * HIDK * BLK, A5);
            const v4f b5 = bias4(bs, B5, w * 16 + 4 * g);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const v4f a = mma<HIDK>(A5, xb + tt * 16 * RS + xr, b5);
                *reinterpret_cast<v4s*>(zA + tt * 16 * ZS + zo + w * 16) = pack4(a[0], a[1], a[2], a[3]);
            }
        } else if (w == STB) {
            if (g < 2) {   // lanes (j, 0): a[0..3]; lanes (j, 1): a[4], a[5], 0, 0
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const int rr = base + tt * 16 + j;
                    const float* an = actions + ((size_t)(rr < n ? rr : n - 1) * horizon + (t + 1)) * ACT;
                    const float a0 = an[4 * g], a1 = an[4 * g + 1];
                    const float a2 = g == 0 ? an[2] : 0.f, a3 = g == 0 ? an[3] : 0.f;
                    *reinterpret_cast<v4s*>(zA + (tt * 16 + j) * ZS + 32 + 4 * g) = pack4(a0, a1, a2, a3);
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    {   // the last state: r1 = relu(W6 [h | z] + b6), r2 = relu(W7 r1 + b7), reward
        gptr Plane = (gptr)Pg + lane * 8;
        if (!AHEAD) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NOB; ++i) {
                gptr W = Plane + W6 + (size_t)own_block(w, i) * K6K * BLK;
                request<DETK>(W, Ah[i]);
                request<STK>(W + (size_t)DETK * BLK, Az[i]);
            }
        }
        v4i A7[NOB][HIDK];
        req_own<HIDK>(Plane, W7, w, A7);
#pragma unroll
        for (int i = 0; i < NOB; ++i) {
            const int ob = own_block(w, i);
            const v4f b = bias4(bs, B6, ob * 16 + 4 * g);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                v4f a = mma<DETK>(Ah[i], hb[cur] + tt * 16 * RS + xr, b);
                a = mma<STK>(Az[i], zA + tt * 16 * ZS + zr, a);
                if (w + WAVES * i < 13) *reinterpret_cast<v4s*>(r1 + tt * 16 * RS + xo + ob * 16) = relu_pack(a);
            }
        }
        __syncthreads();
        fin_dense<HIDK, TT>(bs, B7, A7, r1 + xr, 16 * RS, r2 + xo, w, g);
        __syncthreads();
        if (w == WAVES - 1) score(Plane, horizon - 1);
    }
    if (w == WAVES - 1 && g == 0) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
            if (base + tt * 16 + j < n) costs[base + tt * 16 + j] = acc_cost[tt];
    }
}
}  // namespace

hipError_t launch_rssm_rollout(int n, int horizon, int cost_mode, const unsigned short* params, const float* obs0,
                               const float* actions, float* costs, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (rssm_split_ok(n, horizon)) return launch_rssm_split(n, horizon, cost_mode, params, obs0, actions, costs, st);
    if (n >= 8192)   // more tiles than CUs several times over: two tiles per workgroup share every weight load
        hipLaunchKernelGGL(rssm_rollout_kernel<2>, dim3((n + 31) / 32), dim3(NTHR), 0, st, n, horizon, cost_mode, params, obs0,
                           actions, costs);
    else
        hipLaunchKernelGGL(rssm_rollout_kernel<1>, dim3((n + 15) / 16), dim3(NTHR), 0, st, n, horizon, cost_mode, params, obs0,
                           actions, costs);
    return hipGetLastError();
}
}  // namespace icem
