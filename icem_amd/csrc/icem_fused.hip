// icem_fused.hip -- the f32 throughput kernels for gfx950 (see icem_fused.h).
//
// sample_folded_kernel<H, ROUNDS>   (K1)
//   one thread per (trajectory, action-dim) row: Philox4x32 -> Box-Muller -> the h white draws of
//   the row in registers; inverse real DFT folded on its cos/sin symmetry (t and h-t share the even
//   sum and negate the odd one: ~h*h/2 FMAs instead of h*h) with the table rows as wave-uniform
//   scalar operands; affine (mean/std staged in LDS) + clip; samples parked in an LDS tile laid out
//   like the [n, h, d] output so the slab leaves as coalesced stores.
// rollout_mfma_kernel<H, D, O, KIND> (K2 + K3)
//   one wavefront per 64 trajectories, lane = trajectory.  The model step [o | a] . [A ; B] runs on
//   the matrix pipe as v_mfma_f32_4x4x1 (16 blocks of 4 trajectories, exact f32, the same k-ordered
//   fmaf chain as scalar code): A-operand = 4 model output columns (resident in VGPRs), B-operand =
//   the lane's own x_k, result register i of column tile ct = output 4*ct + i of THIS lane's
//   trajectory -- no transposes, no LDS.  The cost runs on the VALU under the MFMAs; actions stream
//   from HBM/L2 as 16-byte loads one group of steps ahead.  The wave then bitonic-sorts its 64
//   (cost, index) keys and keeps a running sorted top-K: K candidates per wave.
// merge_single_kernel: 1024 threads, one per candidate list; K tournament rounds over the list
//   heads give the global sorted top-K; then gather + refit + epilogue (icem.py:163-211).
#include "icem_fused.h"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "philox.h"
#include "refit.h"

namespace icem {

namespace {

constexpr int HMAX = 32;
constexpr int SWG = 256;  // sampling workgroup

__host__ __device__ constexpr int cmin(int a, int b) { return a < b ? a : b; }

// ---- packed (cost, index) keys: unsigned order == (cost, index) lexicographic order ------------
__device__ __forceinline__ unsigned long long make_key(float c, int idx) {
    c = (c != c) ? INFINITY : c + 0.0f;  // NaN -> +inf; -0 -> +0
    unsigned u = __float_as_uint(c);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned)idx;
}
__device__ __forceinline__ float key_cost(unsigned long long k) {
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)(unsigned)k; }
constexpr unsigned long long KEY_SENTINEL = 0xFF8000007FFFFFFFull;  // (+inf, INT_MAX)

// value of lane (lane ^ J): DPP inside a row of 16 lanes (quad permutes for 1 / 2, row rotations for
// 4 / 8), v_permlane16/32_swap across rows (16 / 32): no LDS-crossbar shuffles
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long x) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)x, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(x >> 32), CTRL, 0xF, 0xF, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
template <int J>
__device__ __forceinline__ unsigned long long xor_partner(unsigned long long k, int lane) {
    if constexpr (J == 1) {
        return dpp_u64<0xB1>(k);  // quad_perm [1,0,3,2]
    } else if constexpr (J == 2) {
        return dpp_u64<0x4E>(k);  // quad_perm [2,3,0,1]
    } else if constexpr (J == 4) {
        const unsigned long long up = dpp_u64<0x12C>(k);    // row_ror:12 -> from lane + 4 (mod 16)
        const unsigned long long down = dpp_u64<0x124>(k);  // row_ror:4  -> from lane - 4 (mod 16)
        return (lane & 4) ? down : up;
    } else if constexpr (J == 8) {
        return dpp_u64<0x128>(k);  // row_ror:8
    } else if constexpr (J == 16) {
        // v_permlane16_swap(a = x, b = x): a <- [x.row0, x.row0, x.row2, x.row2], b <- [x.row1, x.row1, x.row3, x.row3]
        const unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
        auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const bool odd = (lane & 16) != 0;
        return ((unsigned long long)(odd ? h[0] : h[1]) << 32) | (odd ? l[0] : l[1]);
    } else {
        static_assert(J == 32, "xor distance");
        // v_permlane32_swap(a = x, b = x): a <- [x.rows01, x.rows01], b <- [x.rows23, x.rows23]
        const unsigned lo = (unsigned)k, hi = (unsigned)(k >> 32);
        auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        const bool upper = (lane & 32) != 0;
        return ((unsigned long long)(upper ? h[0] : h[1]) << 32) | (upper ? l[0] : l[1]);
    }
}

template <int SIZE, int J>
__device__ __forceinline__ unsigned long long bitonic_step(unsigned long long k, int lane) {
    const unsigned long long o = xor_partner<J>(k, lane);
    const bool up = (lane & SIZE) == 0;  // this block sorts ascending
    const bool lower = (lane & J) == 0;  // this lane keeps the smaller of the pair
    const bool take_min = (up == lower);
    const bool o_less = o < k;
    return (take_min == o_less) ? o : k;
}

template <int SIZE, int J>
__device__ __forceinline__ unsigned long long bitonic_merge(unsigned long long k, int lane) {
    k = bitonic_step<SIZE, J>(k, lane);
    if constexpr (J > 1) k = bitonic_merge<SIZE, J / 2>(k, lane);
    return k;
}

// ascending bitonic sort of one key per lane across the 64-lane wave (21 compare-exchange steps)
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long k, int lane) {
    k = bitonic_merge<2, 1>(k, lane);
    k = bitonic_merge<4, 2>(k, lane);
    k = bitonic_merge<8, 4>(k, lane);
    k = bitonic_merge<16, 8>(k, lane);
    k = bitonic_merge<32, 16>(k, lane);
    k = bitonic_merge<64, 32>(k, lane);
    return k;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------------
// K1
// -------------------------------------------------------------------------------------------------
// The h colored samples of one (trajectory, action-dim) row: per-row RNG stream -> Box-Muller -> h white
// draws in registers -> inverse real DFT folded on its symmetry; emit(t, y) receives sample y of step t.
template <int H, int ROUNDS, typename Emit>
__device__ __forceinline__ void sample_row(const float* __restrict__ W, unsigned gi, unsigned j, unsigned off_lo,
                                           unsigned off_hi, unsigned seed_lo, unsigned seed_hi, Emit&& emit,
                                           bool white = false) {
    constexpr int F = H / 2 + 1;
    static_assert(H <= 32 && H >= 2, "white draws of a row live in 32 registers");
    float g[HMAX];
    Xoshiro128pp rng = row_stream<ROUNDS>(gi, j, off_lo, off_hi, seed_lo, seed_hi);
#pragma unroll
    for (int m = 0; m < H; m += 2) {
        const uint32_t xa = rng.next();
        const uint32_t xb = rng.next();
        box_muller(xa, xb, g[m], g[m + 1]);
    }
    if (white) {  // wave-uniform: noise_beta <= 0, the draws are the samples (icem.py:77)
#pragma unroll
        for (int t = 0; t < H; ++t) emit(t, g[t]);
        return;
    }
    {  // t = 0: every sine is zero
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int m = 0; m < F; m += 2) {
            e0 = __builtin_fmaf(g[m], W[m], e0);
            if (m + 1 < F) e1 = __builtin_fmaf(g[m + 1], W[m + 1], e1);
        }
        emit(0, e0 + e1);
    }
#pragma unroll 1
    for (int tp = 1; tp <= H / 2; ++tp) {
        const float* __restrict__ w = W + tp * HMAX;
        float e0 = 0.f, e1 = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int m = 0; m < F; m += 2) {
            e0 = __builtin_fmaf(g[m], w[m], e0);
            if (m + 1 < F) e1 = __builtin_fmaf(g[m + 1], w[m + 1], e1);
        }
#pragma unroll
        for (int m = F; m < H; m += 2) {
            o0 = __builtin_fmaf(g[m], w[m], o0);
            if (m + 1 < H) o1 = __builtin_fmaf(g[m + 1], w[m + 1], o1);
        }
        const float e = e0 + e1, od = o0 + o1;
        emit(tp, e + od);
        if (H - tp != tp) emit(H - tp, e - od);
    }
}

template <int H, int ROUNDS>
__global__ __launch_bounds__(SWG) void sample_folded_kernel(FastSampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int d = a.d;
    const int hd = H * d;
    const int tpw = SWG / d;
    float* ms = smem;            // mean | std
    float* tile = smem + 2 * hd;  // [tpw, hd]
    const int tid = threadIdx.x;
    for (int e = tid; e < hd; e += SWG) {
        ms[e] = a.mean[e];
        ms[hd + e] = a.std[e];
    }
    const int n_base = blockIdx.x * tpw;
    const int n_here = cmin(tpw, a.n - n_base);
    __syncthreads();
    if (a.n_shift > 0 && blockIdx.x == gridDim.x - 1) {
        // the extra workgroup: shifted elites.  Row (e, j) keeps elites[e, 1:, j] and draws its last action
        // from the full (n_shift, d, h) noise batch of stream off2 (only t = h-1 is used, icem.py:102)
        if (tid < a.n_shift * d) {
            const int e = tid / d;
            const int j = tid - e * d;
            const float lo = a.low[j], hi = a.high[j];
            float last = 0.f;
            sample_row<H, ROUNDS>(a.W, (unsigned)e, (unsigned)j, a.off2_lo, a.off2_hi, a.seed_lo, a.seed_hi,
                                  [&](int t, float y) {
                                      if (t == H - 1) {
                                          float v = __builtin_fmaf(y, ms[hd + t * d + j], ms[t * d + j]);
                                          v = v < lo ? lo : v;
                                          last = v > hi ? hi : v;
                                      }
                                  }, a.white != 0);
            float* dst = a.out + (size_t)(a.n + e) * hd + j;
            const float* src = a.elites_src + (size_t)e * hd + j;
            for (int t = 0; t < H - 1; ++t) dst[t * d] = src[(t + 1) * d];
            dst[(H - 1) * d] = last;
        }
        return;
    }
    if (tid < n_here * d) {
        const int nl = tid / d;
        const int j = tid - nl * d;
        const float lo = a.low[j], hi = a.high[j];
        float* trow = tile + nl * hd + j;
        const float* mrow = ms + j;
        sample_row<H, ROUNDS>(a.W, (unsigned)(a.first_index + n_base + nl), (unsigned)j, a.off_lo, a.off_hi, a.seed_lo,
                              a.seed_hi, [&](int t, float y) {
                                  const float v = __builtin_fmaf(y, mrow[hd + t * d], mrow[t * d]);
                                  trow[t * d] = __builtin_amdgcn_fmed3f(v, lo, hi);  // clip in one v_med3_f32
                              }, a.white != 0);
    }
    __syncthreads();
    if (a.row0_mean && a.first_index + n_base == 0) {  // icem.py:87-88
        for (int e = tid; e < hd; e += SWG) tile[e] = ms[e];
        __syncthreads();
    }
    float* gdst = a.out + (size_t)n_base * hd;
    const int total = n_here * hd;
    if ((hd & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gdst);
        for (int e = tid; e < total / 4; e += SWG) g4[e] = t4[e];
    } else {
        for (int e = tid; e < total; e += SWG) gdst[e] = tile[e];
    }
}

// -------------------------------------------------------------------------------------------------
// K2 + K3: rollout + cost + per-workgroup sorted top-K
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_fn(float x, std::integral_constant<int, 0>) { return x; }
// tanh in ~14 instructions (libm's tanhf is ~40, and a tanh model spends most of its step there): 1 - 2 / (e^2x + 1)
// on the hardware exp2 / rcp, which loses relative accuracy near 0 to cancellation, so |x| < 0.1 takes the odd
// Taylor polynomial to x^7 instead (error < 3e-9 there).  Absolute error <= 2e-7 everywhere, saturates to +-1.
__device__ __forceinline__ float fast_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);  // e^(2x)
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    const float x2 = x * x;
    const float small = x * __builtin_fmaf(x2, __builtin_fmaf(x2, __builtin_fmaf(x2, -17.f / 315.f, 2.f / 15.f), -1.f / 3.f), 1.f);
    return __builtin_fabsf(x) < 0.1f ? small : big;
}
__device__ __forceinline__ float act_fn(float x, std::integral_constant<int, 1>) { return fast_tanh(x); }

// 16 trajectories per wavefront on v_mfma_f32_16x16x4_f32:
// D[16 x 16] += A[16 x 4] . B[4 x 16] with A = a 16 x 4 block of M^T (output column i = lane % 16, contraction
// slot g = lane / 16) and B = X^T (trajectory j = lane % 16, slot g).  The result leaves lane (j, g) holding the
// new observation columns 4g .. 4g+3 of trajectory j in its 4 accumulator registers -- and MFMA number s of the
// next step wants, in lane (j, g), one observation column per contraction slot.  Ordering the contraction so that
// slot g of MFMA s IS column 4g + s makes accumulator register s of one step the B operand of MFMA s of the next:
// no transposes, no LDS, no cross-lane traffic for the first 16 columns, and the model operand of a lane is ONE
// register per MFMA (6 for o=17, d=6, against 92 for the 4x4x1 tiling).  Observation columns >= 16 and the
// actions ride in extra contraction slots: extra e sits in slot e % 4 of MFMA 4 + e/4.  Output columns >= 16 (at
// most 4) are per-lane partial dot products summed over the 4 lanes of a trajectory with v_permlane32/16_swap;
// the same reduction sums the step cost.  A SIMD holds 4 such waves, whose memory / LDS / hazard stalls hide
// under each other's arithmetic.  (f32 MFMA and f32 VALU share one pipe on gfx950 -- tools/ubench/
// mfma_valu_two_waves.hip -- so their cycles add; the kernel is bound by that sum.)
__device__ __forceinline__ float reduce_groups(float x) {
    // sum over lanes l, l^16, l^32, l^48 (the 4 contraction slots of one trajectory), result in all of them
    unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    unsigned v = __float_as_uint(s);
    auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Two sums over the 4 lanes of a trajectory for the price of one: a <- sum of a (valid in lanes 0..31), b <- sum of
// b (valid in lanes 32..63).  v_permlane32_swap exchanges a's upper half with b's lower half, so one add sums both
// values over lane pairs (l, l^32); v_permlane16_swap + add finishes the pairs (l, l^16).  Same association as
// reduce_groups: (x0 + x2) + (x1 + x3).
__device__ __forceinline__ void reduce_groups_pair(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float s = __uint_as_float(r[0]) + __uint_as_float(r[1]);  // [a0+a2, a1+a3, b0+b2, b1+b3] by row of 16 lanes
    const unsigned v = __float_as_uint(s);
    auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    a = b = __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// steps per staged action chunk: divides H, whole vectors (float4, or float2 when h*d is not a multiple of 4) per
// row, odd vector count (16 rows then hit 16 distinct bank groups) when possible
__host__ __device__ constexpr int r16_chunk_steps(int h, int d, int vw) {
    int best = 0;
    for (int tc = 1; tc <= h; ++tc)
        if (h % tc == 0 && (tc * d) % vw == 0 && tc * d <= 110) best = tc;
    return best;
}
template <int VW> struct VecOf;
template <> struct VecOf<4> { using type = float4; };
template <> struct VecOf<2> { using type = float2; };

// Everything a wavefront needs to roll 16 trajectories out; step() is shared by the stand-alone rollout kernel
// and the sample+rollout kernel, so both produce the same bits for the same actions.
template <int H, int D, int O, int KIND>
struct Tile16 {
    // O <= 20: ONE 16-column output tile on the matrix pipe + up to 4 extra columns as permlane-reduced dot products;
    // 20 < O <= 28: TWO output tiles (columns 0..15 and 16..31, zero padded), no extra columns.
    static constexpr int NT = O > 20 ? 2 : 1;
    static constexpr int OP = NT == 2 ? 32 : O;        // observation rows of Mp (the action rows follow)
    static constexpr int REM = NT == 1 && O > 16 ? O - 16 : 0;  // output columns beyond the matrix-pipe tiles
    static constexpr int NKO = 4 * NT;                 // MFMAs fed by accumulator registers (tile s/4, register s%4)
    static constexpr int NX = REM + D;                 // extra contraction entries: columns >= 16 (NT = 1), then the actions
    static constexpr int NKX = (NX + 3) / 4;           // MFMAs that carry them
    static constexpr int NK = NKO + NKX;
    static constexpr int CT4 = NT == 2 ? 32 : ((O + 3) / 4) * 4;  // row stride of Mp
    static constexpr int SLACK = 4;  // floats in front of an action buffer: entries that are not actions read there
    static constexpr int TAIL = 8;   // and behind it: padding entries of the last row
    static_assert(O >= 16 && O <= 28, "observation width 16..28 (the staged observation keeps entry 31 as its zero)");

    float mA[NT][NK];                   // model operands: one register per MFMA
    float wR[REM > 0 ? REM : 1][NK];    // weights of this lane's contraction entries into output column 16 + r
    float cw[NKX];                      // ctrl_w where the extra entry is an action, else 0
    bool is_act[NKX];
    f32x4 obs_init[NT];
    float rem_init[REM > 0 ? REM : 1];
    int perm_base[NT][4], perm_rem[REM > 0 ? REM : 1];  // which observation entries this lane starts from
    float pen, lin_w, ksum, flip_th;
    bool ang_is_col1, use_min;
    int g;

    // All loads are unconditional (Mp carries a zero row behind the model, perm is padded to 32 entries that point
    // at a zero slot): a select behind a load would make the wave wait for it right here, in front of everything
    // the kernel does before it needs the model.
    static constexpr int ZROW = OP + D;  // the zero row of Mp
    __device__ __forceinline__ void load(const FastRolloutArgs& a, int lane) {
        const int j = lane & 15;
        g = lane >> 4;
#pragma unroll
        for (int s = 0; s < NKO; ++s) {
            const int k = 16 * (s / 4) + 4 * g + (s % 4);  // the observation column accumulator (s/4)[s%4] of slot g holds
#pragma unroll
            for (int to = 0; to < NT; ++to) mA[to][s] = a.Mp[k * CT4 + 16 * to + j];
#pragma unroll
            for (int r = 0; r < REM; ++r) wR[r][s] = a.Mp[k * CT4 + 16 + r];
        }
#pragma unroll
        for (int q = 0; q < NKX; ++q) {
            const int e = 4 * q + g;
            const bool valid = e < NX;
            const int k = !valid ? ZROW : (e < REM ? 16 + e : OP + (e - REM));
#pragma unroll
            for (int to = 0; to < NT; ++to) mA[to][NKO + q] = a.Mp[k * CT4 + 16 * to + j];
#pragma unroll
            for (int r = 0; r < REM; ++r) wR[r][NKO + q] = a.Mp[k * CT4 + 16 + r];
            is_act[q] = valid && e >= REM;
            cw[q] = is_act[q] ? a.ctrl_w : 0.f;
        }
        // the start observation is gathered through `perm`: two dependent global round trips if done here.  Only the
        // indices are fetched now; load_obs() picks the values from an LDS copy the kernel stages meanwhile.
#pragma unroll
        for (int to = 0; to < NT; ++to)
#pragma unroll
            for (int v = 0; v < 4; ++v) perm_base[to][v] = a.perm[16 * to + 4 * g + v];
#pragma unroll
        for (int r = 0; r < REM; ++r) perm_rem[r] = a.perm[16 + r];
        // cost terms that read observation columns 0 / 1 live in slot 0 only
        pen = (a.flip_col >= 0 && g == 0) ? a.flip_pen : 0.f;
        lin_w = g == 0 ? a.lin_w : 0.f;
        ang_is_col1 = a.flip_col == 1;
        ksum = a.cost_mode == 0 ? 1.f : 0.f;  // sum: acc = acc + c; final: acc = c
        use_min = a.cost_mode == 1;
        flip_th = a.flip_th;
    }

    // obs: 32 floats in LDS, the o start-observation entries in natural order, zeros behind (perm's padding -> 31)
    // (the empty asm keeps the compiler from hoisting the address arithmetic -- and with it the wait for the perm
    // loads -- up into the prologue)
    __device__ __forceinline__ void load_obs(const float* obs) {
#pragma unroll
        for (int to = 0; to < NT; ++to)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                int pb = perm_base[to][v];
                asm volatile("" : "+v"(pb));
                obs_init[to][v] = obs[pb];
            }
#pragma unroll
        for (int r = 0; r < REM; ++r) {
            int pb = perm_rem[r];
            asm volatile("" : "+v"(pb));
            rem_init[r] = obs[pb];
        }
    }

    // this lane's read pointer into an LDS action buffer whose row of trajectory j starts at buf + SLACK + j * stride
    __device__ __forceinline__ const float* read_ptr(const float* buf, int lane, int stride) const {
        return buf + SLACK + (lane & 15) * stride + ((lane >> 4) - REM);
    }

    // Rollout state of the lane's trajectory share.  Kernels drive it with their own (unrolled) time loop:
    // init, H x step(rd) with entry q of the step's actions at rd[4 * q], then cost().
    struct State {
        f32x4 cur[NT];
        float xr[REM > 0 ? REM : 1];
        float acc_s, acc_b;
    };
    __device__ __forceinline__ void init(State& st) const {
#pragma unroll
        for (int to = 0; to < NT; ++to) st.cur[to] = obs_init[to];
#pragma unroll
        for (int r = 0; r < REM; ++r) st.xr[r] = rem_init[r];
        st.acc_s = 0.f;
        st.acc_b = INFINITY;
    }
    __device__ __forceinline__ void step(State& st, const float* rd) const {
        float xv[NKX];
#pragma unroll
        for (int q = 0; q < NKX; ++q) {
            const float ld = rd[4 * q];
            float v = is_act[q] ? ld : 0.f;
#pragma unroll
            for (int r = 0; r < REM; ++r)
                if (r / 4 == q) v = (g == r % 4) ? st.xr[r] : v;
            xv[q] = v;
        }
        // step cost: this lane's share, then the sum over the trajectory's 4 lanes
        const float ang = ang_is_col1 ? st.cur[0][1] : st.cur[0][0];
        float c = 0.f;
        c += (ang > flip_th) ? pen : 0.f;
        c += (ang < -flip_th) ? pen : 0.f;
#pragma unroll
        for (int q = 0; q < NKX; ++q) c = __builtin_fmaf(xv[q] * xv[q], cw[q], c);
        c = __builtin_fmaf(lin_w, st.cur[0][0], c);
        float pr[REM > 0 ? REM : 1];
#pragma unroll
        for (int r = 0; r < REM; ++r) {
            float p = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) p = __builtin_fmaf(st.cur[0][s], wR[r][s], p);
#pragma unroll
            for (int q = 0; q < NKX; ++q) p = __builtin_fmaf(xv[q], wR[r][NKO + q], p);
            pr[r] = p;
        }
        // the first extra column (consumed by slot 0 = lanes 0..15) shares its reduction with the cost, which is
        // then valid in lanes 32..63 only (see cost())
        if (REM >= 1)
            reduce_groups_pair(pr[0], c);
        else
            c = reduce_groups(c);
#pragma unroll
        for (int r = 1; r < REM; ++r) pr[r] = reduce_groups(pr[r]);
        st.acc_s = __builtin_fmaf(st.acc_s, ksum, c);
        st.acc_b = c < st.acc_b ? c : st.acc_b;
        // model step on the matrix pipe: NT independent accumulator chains
        f32x4 nxt[NT];
#pragma unroll
        for (int to = 0; to < NT; ++to) nxt[to] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKO; ++s)
#pragma unroll
            for (int to = 0; to < NT; ++to)
                nxt[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(mA[to][s], st.cur[s / 4][s % 4], nxt[to], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NKX; ++q)
#pragma unroll
            for (int to = 0; to < NT; ++to)
                nxt[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(mA[to][NKO + q], xv[q], nxt[to], 0, 0, 0);
#pragma unroll
        for (int to = 0; to < NT; ++to)
#pragma unroll
            for (int v = 0; v < 4; ++v) st.cur[to][v] = act_fn(nxt[to][v], std::integral_constant<int, KIND>{});
#pragma unroll
        for (int r = 0; r < REM; ++r) st.xr[r] = act_fn(pr[r], std::integral_constant<int, KIND>{});
    }
    __device__ __forceinline__ float cost(const State& st) const {
        const float v = use_min ? st.acc_b : st.acc_s;
        if (REM == 0) return v;
        // accumulated in lanes 32..63 (reduce_groups_pair): hand lane l + 32's value to lane l
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[1]);
    }
};

// a tile's 16 keys (lanes 0..15, the rest sentinels) join the wave's running sorted top-K: lanes 16..16+K-1 carry
// the running list, one sort
__device__ __forceinline__ unsigned long long topk_push16(unsigned long long run_key, unsigned long long key, bool first,
                                                          int K, int lane) {
    if (!first) {
        const unsigned long long prev = __shfl(run_key, lane - 16, 64);
        if (lane >= 16 && lane < 16 + K) key = prev;
    }
    return wave_sort64(key, lane);
}

// one sorted list per workgroup: tree merge of the first WAVES waves' lists through LDS (`fan` lists per sort),
// wave 0 writes list blockIdx.x.  Every thread of the workgroup must call it.
template <int WAVES>
__device__ __forceinline__ void wg_merge_emit(unsigned long long (*wg_keys)[WAVES][32], unsigned long long run_key, int K,
                                              int lane, int wave, const FastRolloutArgs& a) {
    if (WAVES > 1) {
        const int fan = 4 * K <= 64 ? 4 : 2;
        int lists = WAVES, par = 0;
        if (wave < WAVES && lane < K) wg_keys[0][wave][lane] = run_key;
        __syncthreads();
        while (lists > 1) {
            const int f = lists < fan ? lists : fan;
            const int out = lists / f;
            if (wave < out) {
                unsigned long long k2 = KEY_SENTINEL;
                if (lane < f * K) k2 = wg_keys[par][wave * f + lane / K][lane % K];
                run_key = wave_sort64(k2, lane);
                if (lane < K) wg_keys[par ^ 1][wave][lane] = run_key;
            }
            __syncthreads();
            par ^= 1;
            lists = out;
        }
    }
    if (wave == 0 && lane < K) {
        if (a.part_k) {
            a.part_k[(size_t)lane * gridDim.x + blockIdx.x] = run_key;
        } else {
            a.part_c[(size_t)blockIdx.x * K + lane] = key_cost(run_key);
            a.part_i[(size_t)blockIdx.x * K + lane] = key_idx(run_key);
        }
    }
}

template <int H, int D, int O, int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void rollout16_kernel(FastRolloutArgs a) {
    using Tile = Tile16<H, D, O, KIND>;
    constexpr int HD = H * D;
    constexpr int VW = HD % 4 == 0 ? 4 : 2;    // floats per load: rows are 16-byte aligned only if h*d % 4 == 0
    static_assert(HD % 2 == 0, "8-byte aligned action rows");
    using Vec = typename VecOf<VW>::type;
    constexpr int TC = r16_chunk_steps(H, D, VW);  // steps per action chunk
    static_assert(TC > 0, "no aligned action chunk for this (H, D)");
    constexpr int CB = TC * D;                 // floats per row and chunk
    constexpr int C4 = CB / VW;                // vectors per row and chunk
    constexpr int CBP = (C4 % 2) ? CB : CB + VW;  // LDS row stride: odd number of vectors
    constexpr int NCH = H / TC;
    constexpr int F4 = 16 * C4;                // vectors per chunk of a 16-trajectory tile
    constexpr int NLD = (F4 + 63) / 64;        // cooperative load instructions per chunk
    constexpr int STG = Tile::SLACK + 16 * CBP + Tile::TAIL;
    // the tile's actions are one contiguous 16 x H x D block of HBM: the wave fetches it with full-width coalesced
    // loads, chunk by chunk, into its own LDS buffer; each lane then reads the one or two entries it feeds to the MFMAs
    __shared__ __attribute__((aligned(16))) float stage[WAVES][STG];
    __shared__ unsigned long long wg_keys[2][WAVES][32];
    __shared__ float obs_stage[32];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // model operands and start observation in flight together: one wait at the barrier
    const float obs_reg = a.obs0[(threadIdx.x < 32 && (int)threadIdx.x < a.o) ? threadIdx.x : 0];
    Tile tile;
    tile.load(a, lane);
    if (threadIdx.x < 32) obs_stage[threadIdx.x] = (int)threadIdx.x < a.o ? obs_reg : 0.f;
    __syncthreads();
    tile.load_obs(obs_stage);
    const float* rd0 = tile.read_ptr(stage[wave], lane, CBP);
    // cooperative loads: float4 number f = m * 64 + lane of a chunk is row f / C4, float4 f % C4 of that row
    int ld_row[NLD], ld_c4[NLD];
    bool ld_on[NLD];
#pragma unroll
    for (int m = 0; m < NLD; ++m) {
        const int f = m * 64 + lane;
        ld_on[m] = f < F4;
        ld_row[m] = ld_on[m] ? f / C4 : 0;
        ld_c4[m] = ld_on[m] ? f % C4 : 0;
    }

    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    const int tiles = (a.n_rows + 15) / 16;
    // tile t of the launch belongs to wave t / gridDim.x of workgroup t % gridDim.x: a short launch thins every CU
    for (int tile_id = wave * gridDim.x + blockIdx.x; tile_id < tiles; tile_id += WAVES * gridDim.x) {
        const int row = tile_id * 16 + (lane & 15);
        const bool live = row < a.n_rows;
        const Vec* src[NLD];
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int r = tile_id * 16 + ld_row[m];
            src[m] = reinterpret_cast<const Vec*>(a.actions + (size_t)(r < a.n_rows ? r : 0) * HD) + ld_c4[m];
        }
        Vec pre[NLD];
#pragma unroll
        for (int m = 0; m < NLD; ++m) pre[m] = src[m][0];
        typename Tile::State st;
        tile.init(st);
#pragma unroll
        for (int t = 0; t < H; ++t) {
            if (t % TC == 0) {
                // next chunk: registers -> this wave's LDS buffer (only this wave touches it and a wave's LDS
                // operations execute in order: no barrier), then start fetching the one after
#pragma unroll
                for (int m = 0; m < NLD; ++m)
                    if (ld_on[m])
                        *reinterpret_cast<Vec*>(&stage[wave][Tile::SLACK + ld_row[m] * CBP + VW * ld_c4[m]]) = pre[m];
                if (t / TC + 1 < NCH) {
#pragma unroll
                    for (int m = 0; m < NLD; ++m) pre[m] = src[m][(t / TC + 1) * C4];
                }
            }
            tile.step(st, rd0 + (t % TC) * D);
        }
        const float cost = tile.cost(st);
        if (live && lane < 16) a.costs[row] = cost;
        if (a.K > 0) {
            const unsigned long long key = (lane < 16 && live && row < a.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
            run_key = topk_push16(run_key, key, first, a.K, lane);
            first = false;
        }
    }
    if (a.K > 0) wg_merge_emit<WAVES>(wg_keys, run_key, a.K, lane, wave, a);
}

// -------------------------------------------------------------------------------------------------
// merge: global sorted top-K from <= 256 sorted candidate lists (+ kept elites)
// -------------------------------------------------------------------------------------------------
constexpr int MERGE_WG = 256;  // wave 0 selects (no barriers inside); all 4 waves gather + refit
constexpr int LPL = 4;         // candidate lists per lane of wave 0 (<= 256 lists)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long min_dpp(unsigned long long x) {
    // lanes whose DPP source is invalid / masked keep their own value (old = x, bound_ctrl off)
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)x, (int)(unsigned)x, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(x >> 32), (int)(unsigned)(x >> 32), CTRL, ROW_MASK, 0xF, false);
    const unsigned long long o = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    return o < x ? o : x;
}

// min over the 64 lanes, returned wave-uniform.  DPP only: butterflies inside each row of 16 lanes
// (quad xor 1, quad xor 2, half-row mirror, row mirror), then row_bcast:15 into rows 1/3 and
// row_bcast:31 into rows 2/3; lane 63 ends up with the total.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long x) {
    x = min_dpp<0xB1, 0xF>(x);   // quad_perm [1,0,3,2]
    x = min_dpp<0x4E, 0xF>(x);   // quad_perm [2,3,0,1]
    x = min_dpp<0x141, 0xF>(x);  // row_half_mirror
    x = min_dpp<0x140, 0xF>(x);  // row_mirror
    x = min_dpp<0x142, 0xA>(x);  // row_bcast:15 -> rows 1 and 3
    x = min_dpp<0x143, 0xC>(x);  // row_bcast:31 -> rows 2 and 3
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)x, 63);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(x >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8, zero fill), then
// row_bcast:15 into rows 1 / 3 and row_bcast:31 into rows 2 / 3
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned add_dpp(unsigned x) {
    return x + (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
}
__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned x) {
    x = add_dpp<0x111, 0xF>(x);
    x = add_dpp<0x112, 0xF>(x);
    x = add_dpp<0x114, 0xF>(x);
    x = add_dpp<0x118, 0xF>(x);
    x = add_dpp<0x142, 0xA>(x);
    x = add_dpp<0x143, 0xC>(x);
    return x;
}

// Global sorted top-K of the candidate lists (+ kept elites): ONE wavefront; sel[0..K) receives the keys.
// Lane t owns lists t, t+64, t+128, t+192 (each sorted) in registers; key r of list w sits at
// part_k[r * n_lists + w], so every load is one contiguous 512 bytes.
template <int KREG>
__device__ __forceinline__ void merge_select(const MergeSingleArgs& a, int lane, unsigned long long* cand,
                                             unsigned long long* sel) {
    unsigned long long k[LPL][KREG];
#pragma unroll
    for (int l = 0; l < LPL; ++l) {
        const int list = lane + l * 64;
#pragma unroll
        for (int i = 0; i < KREG; ++i)
            k[l][i] = a.part_k[(size_t)(i < a.K ? i : 0) * a.n_lists + (list < a.n_lists ? list : 0)];
    }
#pragma unroll
    for (int l = 0; l < LPL; ++l) {
        const bool has_list = lane + l * 64 < a.n_lists;
#pragma unroll
        for (int i = 0; i < KREG; ++i) k[l][i] = (has_list && i < a.K) ? k[l][i] : KEY_SENTINEL;
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[1] = wall_clock64();
    if (lane < a.n_keep) {  // kept elite `lane` (icem.py:143-145) joins this lane's first list, order preserved
        unsigned long long v = make_key(a.elites_cost_cur[lane], a.n_global + lane);
#pragma unroll
        for (int i = 0; i < KREG; ++i) {
            const bool sw = v < k[0][i];
            const unsigned long long t = sw ? k[0][i] : v;
            k[0][i] = sw ? v : k[0][i];
            v = t;
        }
    }
    // selection by threshold: the K-th smallest of the 64 lane minima bounds the K-th smallest key overall,
    // so the global top-K is among the keys <= T; those (usually K..2K of them) are compacted into one key
    // per lane and sorted.  Two 64-key sorts instead of K dependent tournament rounds.
    if (a.dbg && threadIdx.x == 0) a.dbg[2] = wall_clock64();
    unsigned long long mine = k[0][0];
#pragma unroll
    for (int l = 1; l < LPL; ++l) mine = k[l][0] < mine ? k[l][0] : mine;
    const unsigned long long srt = wave_sort64(mine, lane);
    const unsigned long long T = __shfl(srt, a.K - 1, 64);
    // compaction: every lane counts its keys <= T, an exclusive DPP scan over the lanes gives it a slot range
    unsigned mine_n = 0;
#pragma unroll
    for (int l = 0; l < LPL; ++l)
#pragma unroll
        for (int i = 0; i < KREG; ++i) mine_n += (k[l][i] <= T && k[l][i] != KEY_SENTINEL) ? 1u : 0u;
    const unsigned incl = wave_incl_scan_u32(mine_n);
    const unsigned n_cand = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    unsigned pos = incl - mine_n;
#pragma unroll
    for (int l = 0; l < LPL; ++l) {
#pragma unroll
        for (int i = 0; i < KREG; ++i) {
            if (k[l][i] <= T && k[l][i] != KEY_SENTINEL) {
                if (pos < 64) cand[pos] = k[l][i];
                ++pos;
            }
        }
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[3] = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (n_cand <= 64) {
        unsigned long long key = lane < (int)n_cand ? *((volatile unsigned long long*)&cand[lane]) : KEY_SENTINEL;
        key = wave_sort64(key, lane);
        if (lane < a.K) sel[lane] = key;
    } else {
        // more than 64 keys tie at or below T: K tournament rounds over the list heads
        for (int r = 0; r < a.K; ++r) {
            unsigned long long head = k[0][0];
#pragma unroll
            for (int l = 1; l < LPL; ++l) head = k[l][0] < head ? k[l][0] : head;
            const unsigned long long best = wave_min_u64(head);
            if (best != KEY_SENTINEL) {  // keys embed the trajectory index: exactly one (lane, list) matches
#pragma unroll
                for (int l = 0; l < LPL; ++l) {
                    if (k[l][0] == best) {
#pragma unroll
                        for (int i = 0; i + 1 < KREG; ++i) k[l][i] = k[l][i + 1];
                        k[l][KREG - 1] = KEY_SENTINEL;
                    }
                }
            }
            if (lane == 0) sel[r] = best;
        }
    }
}

// The same selection with O(1) registers, for the merge prologues: it runs next to sampling waves, so its own
// latency hides but its register count binds the whole kernel.  Heads only for the threshold; one streaming pass
// over the (L2-resident, sorted) lists for the compaction -- keys <= T are a prefix of each list, so a list is left
// as soon as no lane has a survivor at the current depth; if more than 64 keys survive, K tournament rounds over
// per-list cursors.  Same result as merge_select (the K smallest keys, ascending).
__device__ __forceinline__ void merge_select_stream(const MergeSingleArgs& a, int lane, unsigned long long* cand,
                                                    unsigned long long* sel) {
    const int K = a.K, nl = a.n_lists;
    auto key_at = [&](int l, int i) -> unsigned long long {
        const int list = lane + l * 64;
        const unsigned long long v = a.part_k[(size_t)(i < K ? i : 0) * nl + (list < nl ? list : 0)];
        return (list < nl && i < K) ? v : KEY_SENTINEL;
    };
    // kept elite `lane` (icem.py:143-145): a one-key list of its own
    const unsigned long long kept = lane < a.n_keep ? make_key(a.elites_cost_cur[lane], a.n_global + lane) : KEY_SENTINEL;
    unsigned long long mine = kept;
#pragma unroll
    for (int l = 0; l < LPL; ++l) {
        const unsigned long long v = key_at(l, 0);
        mine = v < mine ? v : mine;
    }
    const unsigned long long srt = wave_sort64(mine, lane);
    const unsigned long long T = __shfl(srt, K - 1, 64);
    unsigned n_cand = 0;  // wave-uniform
    auto offer = [&](unsigned long long key) {
        const bool p = key <= T && key != KEY_SENTINEL;
        const unsigned long long m = __ballot(p);
        if (m != 0) {
            const unsigned pos = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (p && pos < 64) cand[pos] = key;
            n_cand += (unsigned)__popcll(m);
        }
        return m != 0;
    };
    offer(kept);
#pragma unroll 1
    for (int l = 0; l < LPL; ++l) {
        if (l * 64 >= nl) break;
#pragma unroll 1
        for (int i = 0; i < K; ++i)
            if (!offer(key_at(l, i))) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (n_cand <= 64) {
        unsigned long long key = lane < (int)n_cand ? *((volatile unsigned long long*)&cand[lane]) : KEY_SENTINEL;
        key = wave_sort64(key, lane);
        if (lane < K) sel[lane] = key;
    } else {
        int cur[LPL];
#pragma unroll
        for (int l = 0; l < LPL; ++l) cur[l] = 0;
        bool kept_live = true;
        for (int r = 0; r < K; ++r) {
            unsigned long long head = kept_live ? kept : KEY_SENTINEL;
#pragma unroll
            for (int l = 0; l < LPL; ++l) {
                const unsigned long long v = key_at(l, cur[l]);
                head = v < head ? v : head;
            }
            const unsigned long long best = wave_min_u64(head);
            if (best != KEY_SENTINEL) {  // keys embed the row index: exactly one (lane, list) holds it
                if (kept_live && kept == best) kept_live = false;
#pragma unroll
                for (int l = 0; l < LPL; ++l)
                    if (key_at(l, cur[l]) == best) ++cur[l];
            }
            if (lane == 0) sel[r] = best;
        }
    }
}

// Sharded runs: the candidates are n_rec <= 128 all-gathered records {cost, gidx, actions} (+ kept elites).  Two
// records and one kept elite per lane; same threshold selection; slot[r] receives the record number of selected key
// r (n_rec + e for kept elite e).  Ties cannot occur: keys embed the global trajectory index.
__device__ __forceinline__ void merge_select_records(const MergeSingleArgs& a, int lane, unsigned long long* cand,
                                                     unsigned long long* sel, int* slot) {
    const int K = a.K, rs = a.h * a.d + 2;
    auto rec_key = [&](int e) -> unsigned long long {
        if (e >= a.n_rec) return KEY_SENTINEL;
        const float* rec = a.records + (size_t)e * rs;
        return make_key(rec[0], reinterpret_cast<const int*>(rec + 1)[0]);
    };
    unsigned long long k[3] = {rec_key(lane), rec_key(lane + 64),
                               lane < a.n_keep ? make_key(a.elites_cost_cur[lane], a.n_global + lane) : KEY_SENTINEL};
    const unsigned long long k0 = k[0], k1 = k[1], k2 = k[2];
    unsigned long long mine = k0 < k1 ? k0 : k1;
    mine = k2 < mine ? k2 : mine;
    const unsigned long long srt = wave_sort64(mine, lane);
    const unsigned long long T = __shfl(srt, K - 1, 64);
    unsigned n_cand = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool p = k[i] <= T && k[i] != KEY_SENTINEL;
        const unsigned long long m = __ballot(p);
        const unsigned pos = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (p && pos < 64) cand[pos] = k[i];
        n_cand += (unsigned)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (n_cand <= 64) {
        unsigned long long key = lane < (int)n_cand ? *((volatile unsigned long long*)&cand[lane]) : KEY_SENTINEL;
        key = wave_sort64(key, lane);
        if (lane < K) sel[lane] = key;
    } else {
        for (int r = 0; r < K; ++r) {
            unsigned long long head = k[0] < k[1] ? k[0] : k[1];
            head = k[2] < head ? k[2] : head;
            const unsigned long long best = wave_min_u64(head);
            if (best != KEY_SENTINEL) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (k[i] == best) k[i] = KEY_SENTINEL;
            }
            if (lane == 0) sel[r] = best;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int r = 0; r < K; ++r) {
        const unsigned long long s = *((volatile unsigned long long*)&sel[r]);
        if (s != KEY_SENTINEL) {
            if (k0 == s) slot[r] = lane;
            if (k1 == s) slot[r] = lane + 64;
            if (k2 == s) slot[r] = a.n_rec + lane;
        } else if (lane == 0) {
            slot[r] = 0;  // fewer than K live candidates: reference behaviour undefined, stay in bounds
        }
    }
}

// pointers to the K selected rows (icem.py:201): pool rows, or kept elites behind index n_global
template <int KREG, bool REC>
__device__ __forceinline__ void merge_rows(const MergeSingleArgs& a, const unsigned long long* sel, const int* slot,
                                           const float* (&rows)[KREG]) {
    const int hd = a.h * a.d;
#pragma unroll
    for (int r = 0; r < KREG; ++r) {
        const int rr = r < a.K ? r : 0;
        if (REC) {  // sharded run: candidates are records
            const int e = slot[rr];
            rows[r] = e < a.n_rec ? a.records + (size_t)e * (hd + 2) + 2 : a.elites_cur + (size_t)(e - a.n_rec) * hd;
        } else {
            const int g = key_idx(sel[rr]);
            rows[r] = g < a.n_pool ? a.actions + (size_t)g * hd : a.elites_cur + (size_t)(g - a.n_global) * hd;
        }
    }
}

template <int KREG, bool REC>
__global__ __launch_bounds__(MERGE_WG) void merge_single_kernel(MergeSingleArgs a) {
    __shared__ unsigned long long sel[64];
    __shared__ unsigned long long cand[64];
    __shared__ int slot[64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* new_mean = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hd = a.h * a.d;
    if (a.dbg && threadIdx.x == 0) a.dbg[0] = wall_clock64();
    // old mean/std of this thread's elements: issued now, consumed after the selection
    constexpr int EPL = 4;
    const bool pre = hd <= MERGE_WG * EPL;
    float om[EPL], os[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int e = tid + i * MERGE_WG;
        om[i] = (pre && e < hd) ? a.mean[e] : 0.f;
        os[i] = (pre && e < hd) ? a.std[e] : 0.f;
    }
    if (tid < 64) {
        if constexpr (REC)
            merge_select_records(a, lane, cand, sel, slot);
        else
            merge_select<KREG>(a, lane, cand, sel);
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[4] = wall_clock64();
    __syncthreads();
    // ---- all 4 waves: gather + refit (icem.py:201-211); row pointers first, then all K loads in flight ----
    const float* rows[KREG];
    merge_rows<KREG, REC>(a, sel, slot, rows);
    auto finish_one = [&](int e, float old_mean, float old_std) {
        float xs[KREG];
#pragma unroll
        for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
#pragma unroll
        for (int r = 0; r < KREG; ++r)
            if (r < a.K) a.elites_next[(size_t)r * hd + e] = xs[r];
        float nm, ns;
        refit_element_regs<float, KREG>(a.K, a.alpha, old_mean, old_std, xs, nm, ns);
        if (!a.last) {
            a.mean_out[e] = nm;
            a.std_out[e] = ns;
        } else {
            new_mean[e] = nm;
        }
    };
    if (pre) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int e = tid + i * MERGE_WG;
            if (e < hd) finish_one(e, om[i], os[i]);
        }
    } else {
        for (int e = tid; e < hd; e += MERGE_WG) finish_one(e, a.mean[e], a.std[e]);
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[5] = wall_clock64();
    if (tid < a.K) a.elites_cost_next[tid] = key_cost(sel[tid]);
    if (a.last) {
        __syncthreads();
        for (int e = tid; e < hd; e += MERGE_WG) {
            const int j = e % a.d;
            a.mean_out[e] = (e + a.d < hd) ? new_mean[e + a.d] : new_mean[e];
            a.std_out[e] = (a.high[j] - a.low[j]) / 2.f * a.init_std;
        }
        if (tid < a.d) a.executed[tid] = rows[0][tid];
        if (tid == 0) a.best_cost[0] = key_cost(sel[0]);
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[6] = wall_clock64();
}

// Sharded runs: this rank's K best candidates (same selection) packed as records {cost, gidx, actions[h*d]} for
// the all-gather.  Local pool row li is global trajectory shard_lo + li, or n_global + (li - n_loc) for the
// replicated shifted elites behind the shard (icem_amd/distributed.py).
template <int KREG>
__global__ __launch_bounds__(MERGE_WG) void pack_records_kernel(MergeSingleArgs a, int n_loc, int shard_lo, float* records) {
    __shared__ unsigned long long sel[64];
    __shared__ unsigned long long cand[64];
    const int tid = threadIdx.x;
    const int hd = a.h * a.d;
    const int rs = hd + 2;
    if (tid < 64) merge_select<KREG>(a, tid, cand, sel);
    __syncthreads();
    // headers by the first K threads; rows: element e of all K rows per thread, every load in flight before a store
    if (tid < a.K) {
        const unsigned long long key = sel[tid];
        float* rec = records + (size_t)tid * rs;
        if (key == KEY_SENTINEL) {  // fewer than K candidates on this rank
            rec[0] = INFINITY;
            reinterpret_cast<int*>(rec + 1)[0] = INT_MAX;
        } else {
            const int li = key_idx(key);
            rec[0] = key_cost(key);
            reinterpret_cast<int*>(rec + 1)[0] = li < n_loc ? shard_lo + li : a.n_global + (li - n_loc);
        }
    }
    const float* rows[KREG];
    bool dead[KREG];
#pragma unroll
    for (int r = 0; r < KREG; ++r) {
        const unsigned long long key = sel[r < a.K ? r : 0];
        dead[r] = key == KEY_SENTINEL;
        rows[r] = a.actions + (size_t)(dead[r] ? 0 : key_idx(key)) * hd;
    }
    for (int e = tid; e < hd; e += MERGE_WG) {
        float xs[KREG];
#pragma unroll
        for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
#pragma unroll
        for (int r = 0; r < KREG; ++r)
            if (r < a.K) records[(size_t)r * rs + 2 + e] = dead[r] ? 0.f : xs[r];
    }
}

// -------------------------------------------------------------------------------------------------
// K1 with the PREVIOUS iteration's K3 + K4 in its prologue (populations too large for the single-launch kernel)
// -------------------------------------------------------------------------------------------------
// sample_folded_kernel plus one wavefront per workgroup that runs the low-register selection
// (merge_select_stream) on the previous iteration's candidate lists while the 4 sampling waves draw their noise
// into the LDS tile; then all 5 waves gather the K elite rows and refit, the affine map + clip is applied to the
// tile and the tile leaves as before.  Every workgroup redoes the same merge (L2 serves the 20 KB of keys and 7 KB
// of elite rows), workgroup 0 publishes it.  Saves the merge launch (8.6 + 2.6 us) for ~3 us more sampler time.
template <int H, int ROUNDS, int KREG, bool REC>
__global__ __launch_bounds__(SWG + 64) void sample_folded_merge_kernel(FastSampleMergeArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned long long sel[64];
    __shared__ unsigned long long cand[64];
    __shared__ int slot[64];
    constexpr int NTT = SWG + 64;
    const FastSampleArgs& a = args.s;
    const MergeSingleArgs& m = args.m;
    const int d = a.d;
    const int hd = H * d;
    const int tpw = SWG / d;
    float* ms = smem;            // mean | std (computed here)
    float* tile = smem + 2 * hd;  // [tpw, hd]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int n_base = blockIdx.x * tpw;
    const int n_here = cmin(tpw, a.n - n_base);
    const bool has_row = tid < n_here * d;
    const int nl = tid / d;
    const int j = tid - nl * d;
    float* trow = tile + nl * hd + j;
    if (tid >= SWG) {
        if constexpr (REC)
            merge_select_records(m, lane, cand, sel, slot);
        else
            merge_select_stream(m, lane, cand, sel);
    } else if (has_row) {
        sample_row<H, ROUNDS>(a.W, (unsigned)(a.first_index + n_base + nl), (unsigned)j, a.off_lo, a.off_hi, a.seed_lo,
                              a.seed_hi, [&](int t, float y) { trow[t * d] = y; }, a.white != 0);
    }
    __syncthreads();
    {
        const float* rows[KREG];
        merge_rows<KREG, REC>(m, sel, slot, rows);
        for (int e = tid; e < hd; e += NTT) {
            float xs[KREG];
#pragma unroll
            for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
            float nm, ns;
            refit_element_regs<float, KREG>(m.K, m.alpha, m.mean[e], m.std[e], xs, nm, ns);
            ms[e] = nm;
            ms[hd + e] = ns;
            if (blockIdx.x == 0) {
                m.mean_out[e] = nm;
                m.std_out[e] = ns;
#pragma unroll
                for (int r = 0; r < KREG; ++r)
                    if (r < m.K) m.elites_next[(size_t)r * hd + e] = xs[r];
            }
        }
        if (blockIdx.x == 0 && tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
    }
    __syncthreads();
    if (has_row) {
        const float lo = a.low[j], hi = a.high[j];
        const float* mrow = ms + j;
        for (int t = 0; t < H; ++t) {
            const float v = __builtin_fmaf(trow[t * d], mrow[hd + t * d], mrow[t * d]);
            trow[t * d] = __builtin_amdgcn_fmed3f(v, lo, hi);
        }
    }
    __syncthreads();
    if (a.row0_mean && a.first_index + n_base == 0) {  // icem.py:87-88
        for (int e = tid; e < hd; e += NTT) tile[e] = ms[e];
        __syncthreads();
    }
    float* gdst = a.out + (size_t)n_base * hd;
    const int total = n_here * hd;
    if ((hd & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gdst);
        for (int e = tid; e < total / 4; e += NTT) g4[e] = t4[e];
    } else {
        for (int e = tid; e < total; e += NTT) gdst[e] = tile[e];
    }
}

// -------------------------------------------------------------------------------------------------
// K1 + K2 + K3 in one launch for small populations (+ the PREVIOUS iteration's K3 + K4 in its prologue)
// -------------------------------------------------------------------------------------------------
// With a few thousand trajectories the chip is mostly empty and an iteration is a chain of latencies: launch,
// prologue, one thread's RNG -> DFT chain, HBM round trip of the actions, launch, prologue, 30 dependent model
// steps, launch, merge.  Here a workgroup samples 16 * RW trajectories into an LDS tile (one thread per
// (trajectory, dim) row, same code as sample_folded_kernel), writes the tile to HBM for the elite gather, and its
// first RW waves roll the trajectories out straight from the tile (same code as rollout16_kernel): one launch, no
// HBM round trip.
// KREG > 0 ("merge prologue"): the distribution this iteration samples from is not in memory yet -- every
// workgroup computes it itself from the previous iteration's candidate lists.  An extra wavefront runs the
// selection (merge_select) WHILE the sampling waves draw their noise (which does not depend on mean / std: the
// raw colored samples are parked in the tile); then all threads gather the K elite rows and refit (same
// arithmetic as merge_single_kernel, so every workgroup gets the same bits), the affine map + clip is applied to
// the tile, and the iteration proceeds as above.  Workgroup 0 also writes the new distribution and elite set for
// the host / the next launch.  That removes the merge launch and hides its latency behind the sampling.  The
// previous pool, lists and distribution are read while this launch writes new ones: all three are ping-pong
// buffers (icem_plan_step).
// The three kinds of rows of a single-launch slab.  RAW: the
// distribution is still being computed -- park the raw colored samples (merge prologue, sampled rows only).
template <int H, int D, int ROUNDS, bool RAW>
__device__ __forceinline__ void sample_into_tile(const FastSampleArgs& sa, int n_rows, int r_mine, int jd, float* trow,
                                                 const float* mrow) {
    constexpr int HD = H * D;
    if (r_mine < sa.n) {
        if (RAW) {
            sample_row<H, ROUNDS>(sa.W, (unsigned)(sa.first_index + r_mine), (unsigned)jd, sa.off_lo, sa.off_hi,
                                  sa.seed_lo, sa.seed_hi, [&](int t, float y) { trow[t * D] = y; }, sa.white != 0);
        } else {
            const float lo = sa.low[jd], hi = sa.high[jd];
            sample_row<H, ROUNDS>(sa.W, (unsigned)(sa.first_index + r_mine), (unsigned)jd, sa.off_lo, sa.off_hi,
                                  sa.seed_lo, sa.seed_hi, [&](int t, float y) {
                                      const float v = __builtin_fmaf(y, mrow[HD + t * D], mrow[t * D]);
                                      trow[t * D] = __builtin_amdgcn_fmed3f(v, lo, hi);
                                  }, sa.white != 0);
        }
    } else if (r_mine < n_rows && !RAW) {
        // shifted elite e: elites[e, 1:, j] and a last action drawn from the full (n_shift, d, h) noise batch
        // of stream off2 (only t = h-1 is used, icem.py:102); iteration 0 only, which has no merge prologue
        const int e = r_mine - sa.n;
        const float lo = sa.low[jd], hi = sa.high[jd];
        float last = 0.f;
        sample_row<H, ROUNDS>(sa.W, (unsigned)e, (unsigned)jd, sa.off2_lo, sa.off2_hi, sa.seed_lo, sa.seed_hi,
                              [&](int t, float y) {
                                  if (t == H - 1) {
                                      const float v = __builtin_fmaf(y, mrow[HD + t * D], mrow[t * D]);
                                      last = __builtin_amdgcn_fmed3f(v, lo, hi);
                                  }
                              }, sa.white != 0);
        const float* src = sa.elites_src + (size_t)e * HD + jd;
        for (int t = 0; t < H - 1; ++t) trow[t * D] = src[(t + 1) * D];
        trow[(H - 1) * D] = last;
    } else {
        for (int t = 0; t < H; ++t) trow[t * D] = 0.f;  // past the end: rolled out, dropped
    }
}

// One wave rolls its 16 trajectories of the slab out of the LDS tile, stores the costs and folds them into its
// running candidate list.
template <typename Tile, int H, int D>
__device__ __forceinline__ unsigned long long rollout_slab(Tile& tile, const FastRolloutArgs& ra, const float* rd0, int row,
                                                           int n_rows, unsigned long long run_key, bool first, int lane) {
    const bool live = row < n_rows;
    typename Tile::State st;
    tile.init(st);
#pragma unroll
    for (int t = 0; t < H; ++t) tile.step(st, rd0 + t * D);
    const float cost = tile.cost(st);
    if (live && lane < 16) ra.costs[row] = cost;
    if (ra.K > 0) {
        const unsigned long long key = (lane < 16 && live && row < ra.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
        run_key = topk_push16(run_key, key, first, ra.K, lane);
    }
    return run_key;
}

template <int H, int D, int O, int KIND, int ROUNDS, int RW, int KREG, bool REC>
__global__ __launch_bounds__(((16 * RW * D + 63) / 64) * 64 + (KREG > 0 ? 64 : 0)) void sample_rollout_kernel(FastIterArgs a) {
    using Tile = Tile16<H, D, O, KIND>;
    constexpr bool PM = KREG > 0;
    constexpr int HD = H * D;
    constexpr int TPB = 16 * RW;                     // trajectories per workgroup
    constexpr int ROWS = TPB * D;                    // (trajectory, dim) rows, one thread each
    constexpr int NT = ((ROWS + 63) / 64) * 64;      // sampling threads
    constexpr int NTT = NT + (PM ? 64 : 0);          // + the selection wavefront
    static_assert(NTT <= 1024 && HD % 2 == 0, "workgroup shape");
    constexpr int VW = HD % 4 == 0 ? 4 : 2;  // floats per vector of the tile -> HBM copy (rows are 4 * HD bytes)
    using Vec = typename VecOf<VW>::type;
    __shared__ __attribute__((aligned(16))) float ms[2 * HD];  // mean | std
    __shared__ __attribute__((aligned(16))) float tilebuf[Tile::SLACK + TPB * HD + Tile::TAIL];
    __shared__ unsigned long long wg_keys[2][RW][32];
    __shared__ float obs_stage[32];
    __shared__ unsigned long long sel[PM ? 64 : 1];
    __shared__ unsigned long long cand[PM ? 64 : 1];
    __shared__ int slot[PM ? 64 : 1];
    float* tile_rows = tilebuf + Tile::SLACK;
    const FastSampleArgs& sa = a.s;
    const FastRolloutArgs& ra = a.r;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_rows = ra.n_rows;       // sa.n sampled rows, then sa.n_shift shifted elites
    const int base = blockIdx.x * TPB;  // one slab of TPB trajectories per workgroup (launch_sample_rollout)
    if (base >= n_rows) return;
    if (ra.dbg && tid == 0 && blockIdx.x == 0) ra.dbg[8] = wall_clock64();
    // Order matters at this size: kernel arguments arrive through serialized scalar loads, so everything the RNG
    // chain does not need (start observation, model operands, bounds) is fetched AFTER the sampling got going.
    const int nl = tid / D, jd = tid - nl * D;
    const bool has_row = tid < ROWS;
    float* trow = tile_rows + nl * HD + jd;
    const float* mrow = ms + jd;
    Tile tile;
    float obs_reg = 0.f;
    if (!PM) {  // iteration 0 of an MPC step: the distribution is in memory; the model operands ride the same wait
        obs_reg = ra.obs0[(tid < 32 && tid < ra.o) ? tid : 0];
        if (wave < RW) tile.load(ra, lane);
        for (int e = tid; e < HD; e += NTT) {
            ms[e] = sa.mean[e];
            ms[HD + e] = sa.std[e];
        }
        __syncthreads();
    }
    if (ra.dbg && tid == 0 && blockIdx.x == 0) ra.dbg[9] = wall_clock64();
    const int r_mine = base + nl;
    if (has_row) sample_into_tile<H, D, ROUNDS, PM>(sa, n_rows, r_mine, jd, trow, mrow);
    if constexpr (PM) {
        if (tid >= NT) {
            if constexpr (REC)
                merge_select_records(a.m, lane, cand, sel, slot);
            else if constexpr (RW >= 8)  // 13 waves share the register file: the low-register selection
                merge_select_stream(a.m, lane, cand, sel);
            else
                merge_select<KREG>(a.m, lane, cand, sel);
        }
    }
    if (PM) {  // now the rest of the inputs: in flight across the barriers below
        obs_reg = ra.obs0[(tid < 32 && tid < ra.o) ? tid : 0];
        if (wave < RW) tile.load(ra, lane);
    }
    const float* rd0 = tile.read_ptr(tilebuf + (wave < RW ? wave : 0) * 16 * HD, lane, HD);
    if constexpr (PM) {
        const MergeSingleArgs& m = a.m;
        __syncthreads();
        // all threads: gather the elite rows + refit (icem.py:201-211) -> this workgroup's mean / std
        const float* rows[KREG > 0 ? KREG : 1];
        merge_rows<KREG, REC>(m, sel, slot, rows);
        for (int e = tid; e < HD; e += NTT) {
            float xs[KREG > 0 ? KREG : 1];
#pragma unroll
            for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
            float nm, ns;
            refit_element_regs<float, KREG>(m.K, m.alpha, m.mean[e], m.std[e], xs, nm, ns);
            ms[e] = nm;
            ms[HD + e] = ns;
            if (blockIdx.x == 0) {
                m.mean_out[e] = nm;
                m.std_out[e] = ns;
#pragma unroll
                for (int r = 0; r < KREG; ++r)
                    if (r < m.K) m.elites_next[(size_t)r * HD + e] = xs[r];
            }
        }
        if (blockIdx.x == 0 && tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
        __syncthreads();
        if (has_row && r_mine < sa.n) {  // y * std + mean, clipped (icem.py:79)
            const float lo = sa.low[jd], hi = sa.high[jd];
            for (int t = 0; t < H; ++t) {
                const float v = __builtin_fmaf(trow[t * D], mrow[HD + t * D], mrow[t * D]);
                trow[t * D] = __builtin_amdgcn_fmed3f(v, lo, hi);
            }
        }
    }
    if (tid < 32) {  // start observation -> LDS (consumed after the barrier)
        float ov = obs_reg;
        asm volatile("" : "+v"(ov));  // wait for the load here, not where it was issued
        obs_stage[tid] = tid < ra.o ? ov : 0.f;
    }
    if (ra.dbg && tid == 0 && blockIdx.x == 0) ra.dbg[10] = wall_clock64();
    __syncthreads();
    if (sa.row0_mean && sa.first_index + base == 0) {  // icem.py:87-88
        for (int e = tid; e < HD; e += NTT) tile_rows[e] = ms[e];
        __syncthreads();
    }
    {   // the tile is a contiguous block of the action tensor
        const int total4 = (n_rows - base < TPB ? n_rows - base : TPB) * (HD / VW);
        const Vec* t4 = reinterpret_cast<const Vec*>(tile_rows);
        Vec* g4 = reinterpret_cast<Vec*>(sa.out + (size_t)base * HD);
        for (int e = tid; e < total4; e += NTT) g4[e] = t4[e];
    }
    if (ra.dbg && tid == 0 && blockIdx.x == 0) ra.dbg[11] = wall_clock64();
    unsigned long long run_key = KEY_SENTINEL;
    if (wave < RW) {
        tile.load_obs(obs_stage);
        run_key = rollout_slab<Tile, H, D>(tile, ra, rd0, base + wave * 16 + (lane & 15), n_rows, run_key, true, lane);
        if (ra.dbg && tid == 0 && blockIdx.x == 0) ra.dbg[12] = wall_clock64();
    }
    if (ra.dbg && tid == 0 && blockIdx.x == 0) ra.dbg[13] = wall_clock64();
    if (ra.K > 0) wg_merge_emit<RW>(wg_keys, run_key, ra.K, lane, wave, ra);
    if (ra.dbg && tid == 0 && blockIdx.x == 0) ra.dbg[14] = wall_clock64();
}

}  // namespace

// shapes (H, D, O) with a compiled matrix-pipe rollout; anything else runs on the generic kernels
#define ICEM_FAST_SHAPES(X) X(30, 6, 17) X(30, 6, 18) X(12, 6, 17) X(13, 4, 17) X(30, 17, 24)

// rollout waves a single-launch workgroup can hold for this shape: 16 * RW * D sampling threads (+ 64) within 1024
// threads, the [16 * RW, H, D] tile within ~120 KB of LDS
constexpr int single_launch_max_rw(int h, int d) {
    int best = 0;
    for (int rw = 1; rw <= 8; rw *= 2)
        if (((16 * rw * d + 63) / 64) * 64 + 64 <= 1024 && 16 * rw * h * d * 4 <= 120 * 1024) best = rw;
    return best;
}

bool fast_rollout_supported(int h, int d, int O, int K) {
    if (K > 32) return false;
#define X(HH, DD, OO) \
    if (h == HH && d == DD && O == OO) return true;
    ICEM_FAST_SHAPES(X)
#undef X
    return false;
}

// rollout launch shape: one 16-trajectory tile per wave while they fit, at most FAST_MAX_LISTS workgroups (= lists)
static void r16_shape(int n_rows, int* grid, int* waves) {
    const int tiles = std::max(1, (n_rows + 15) / 16);
    const int g = std::min(tiles, FAST_MAX_LISTS);
    int w = 1;
    while (w < 16 && w * g < tiles) w *= 2;
    *grid = g;
    *waves = w;
}

int rollout_lists(int h, int d, int O, int n_rows) {
    int g, w;
    r16_shape(n_rows, &g, &w);
    return g;
}

void launch_rollout16(const FastRolloutArgs& a, int h, int d, int O, int kind, hipStream_t st) {
    int grid, waves;
    r16_shape(a.n_rows, &grid, &waves);
#define XW(HH, DD, OO, WW)                                                                                  \
    if (waves == WW) {                                                                                      \
        if (kind == 1)                                                                                      \
            hipLaunchKernelGGL((rollout16_kernel<HH, DD, OO, 1, WW>), dim3(grid), dim3(64 * WW), 0, st, a); \
        else                                                                                                \
            hipLaunchKernelGGL((rollout16_kernel<HH, DD, OO, 0, WW>), dim3(grid), dim3(64 * WW), 0, st, a); \
        return;                                                                                             \
    }
#define XR(HH, DD, OO)                   \
    if (h == HH && d == DD && O == OO) { \
        XW(HH, DD, OO, 1)                \
        XW(HH, DD, OO, 2)                \
        XW(HH, DD, OO, 4)                \
        XW(HH, DD, OO, 8)                \
        XW(HH, DD, OO, 16)               \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XW
}

// single-launch iteration: compiled for the default generator (10 Philox rounds) and 1, 2, 4 or 8 rollout waves per
// workgroup, one slab of 16 * rw trajectories each, at most FAST_MAX_LISTS workgroups (= candidate lists).
// sample_rollout_lists: workgroups of the launch, 0 when the shape or size is outside that (use the two-kernel
// path).  (Several slabs per workgroup through the same LDS tile were tried for larger populations: with one
// 92 KB tile per CU the sampling and rollout phases of a workgroup run back to back at 2-3 waves per SIMD, and
// N=65 536 took 297 instead of 220 us per MPC step -- the two full-occupancy kernels win there.)
static bool sample_rollout_shape(int h, int d, int O, int rounds, int n_rows, int* grid_out, int* rw_out) {
    static const int max_rw = [] { const char* e = getenv("ICEM_FUSE_MAX_RW"); return e ? atoi(e) : 8; }();
    int grid, rw;
    r16_shape(n_rows, &grid, &rw);
    if (rounds != 10 || rw > max_rw || n_rows <= 0 || !fast_rollout_supported(h, d, O, 1) || !fast_sample_supported(h, d))
        return false;
    if (rw > single_launch_max_rw(h, d)) return false;  // one slab of 16 * rw trajectories per workgroup
    *grid_out = std::min(grid, (n_rows + 16 * rw - 1) / (16 * rw));
    *rw_out = rw;
    return true;
}

int sample_rollout_lists(int h, int d, int O, int rounds, int n_rows) {
    int grid, rw;
    return sample_rollout_shape(h, d, O, rounds, n_rows, &grid, &rw) ? grid : 0;
}

// merge prologue: the selection wavefront joins the sampling waves (8 rollout waves: 13 waves share the register
// file, the selection runs in its low-register form)
bool sample_rollout_merge_ok(int h, int d, int O, int rounds, int n_rows, int K) {
    static const int on = [] { const char* e = getenv("ICEM_MERGE_PROLOGUE"); return e ? atoi(e) : 1; }();
    int grid, rw;
    return on && K + 1 <= 12 && sample_rollout_shape(h, d, O, rounds, n_rows, &grid, &rw);
}

void launch_sample_rollout(const FastIterArgs& a, int h, int d, int O, int kind, bool merge_prologue, hipStream_t st) {
    int grid, rw;
    if (!sample_rollout_shape(h, d, O, 10, a.r.n_rows, &grid, &rw)) return;
#define XK(HH, DD, OO, WW, KR, RC)                                                                                      \
    {                                                                                                                   \
        constexpr int NT = ((16 * WW * DD + 63) / 64) * 64 + (KR > 0 ? 64 : 0);                                         \
        if (kind == 1)                                                                                                  \
            hipLaunchKernelGGL((sample_rollout_kernel<HH, DD, OO, 1, 10, WW, KR, RC>), dim3(grid), dim3(NT), 0, st, a); \
        else                                                                                                            \
            hipLaunchKernelGGL((sample_rollout_kernel<HH, DD, OO, 0, 10, WW, KR, RC>), dim3(grid), dim3(NT), 0, st, a); \
        return;                                                                                                         \
    }
#define XW(HH, DD, OO, WW)                                                   \
    if constexpr (WW <= single_launch_max_rw(HH, DD)) {                      \
        if (rw == WW) {                                                      \
            if (merge_prologue && a.m.records) XK(HH, DD, OO, WW, 12, true)  \
            if (merge_prologue) XK(HH, DD, OO, WW, 12, false)                \
            XK(HH, DD, OO, WW, 0, false)                                     \
        }                                                                    \
    }
#define XR(HH, DD, OO)                   \
    if (h == HH && d == DD && O == OO) { \
        XW(HH, DD, OO, 1)                \
        XW(HH, DD, OO, 2)                \
        XW(HH, DD, OO, 4)                \
        XW(HH, DD, OO, 8)                \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XW
#undef XK
}

#define ICEM_FAST_HORIZONS(X) X(30) X(12) X(13) X(10)

bool fast_sample_supported(int h, int d) {
    if (d > SWG) return false;
#define X(HH) \
    if (h == HH) return true;
    ICEM_FAST_HORIZONS(X)
#undef X
    return false;
}

// sampler with the previous iteration's merge in its prologue (default generator only, K <= 11, no shifted elites)
bool sample_folded_merge_ok(int h, int d, int rounds, int K) {
    static const int on = [] { const char* e = getenv("ICEM_MERGE_PROLOGUE"); return e ? atoi(e) : 1; }();
    return on && rounds == 10 && K + 1 <= 12 && fast_sample_supported(h, d);
}

void launch_sample_folded_merge(const FastSampleMergeArgs& a, hipStream_t st) {
    const int tpw = SWG / a.s.d;
    const int grid = (a.s.n + tpw - 1) / tpw;
    const size_t lds = ((size_t)2 * a.s.h * a.s.d + (size_t)tpw * a.s.h * a.s.d) * sizeof(float);
#define X(HH)                                                                                                  \
    if (a.s.h == HH) {                                                                                         \
        if (a.m.records)                                                                                       \
            hipLaunchKernelGGL((sample_folded_merge_kernel<HH, 10, 12, true>), dim3(grid), dim3(SWG + 64), lds, st, a);  \
        else                                                                                                   \
            hipLaunchKernelGGL((sample_folded_merge_kernel<HH, 10, 12, false>), dim3(grid), dim3(SWG + 64), lds, st, a); \
        return;                                                                                                \
    }
    ICEM_FAST_HORIZONS(X)
#undef X
}

void launch_sample_folded(const FastSampleArgs& a, int rounds, hipStream_t st) {
    const int tpw = SWG / a.d;
    const int grid = (a.n + tpw - 1) / tpw + (a.n_shift > 0 ? 1 : 0);
    const size_t lds = ((size_t)2 * a.h * a.d + (size_t)tpw * a.h * a.d) * sizeof(float);
#define X(HH)                                                                                        \
    if (a.h == HH) {                                                                                 \
        if (rounds == 7)                                                                             \
            hipLaunchKernelGGL((sample_folded_kernel<HH, 7>), dim3(grid), dim3(SWG), lds, st, a);    \
        else                                                                                         \
            hipLaunchKernelGGL((sample_folded_kernel<HH, 10>), dim3(grid), dim3(SWG), lds, st, a);   \
        return;                                                                                      \
    }
    ICEM_FAST_HORIZONS(X)
#undef X
}

void launch_pack_records(const MergeSingleArgs& a, int n_loc, int shard_lo, float* records, hipStream_t st) {
    if (a.K + 1 <= 12)
        hipLaunchKernelGGL((pack_records_kernel<12>), dim3(1), dim3(MERGE_WG), 0, st, a, n_loc, shard_lo, records);
    else
        hipLaunchKernelGGL((pack_records_kernel<34>), dim3(1), dim3(MERGE_WG), 0, st, a, n_loc, shard_lo, records);
}

void launch_merge_single(const MergeSingleArgs& a, hipStream_t st) {
    const size_t lds = (size_t)a.h * a.d * sizeof(float);
    if (a.records) {
        if (a.K + 1 <= 12)
            hipLaunchKernelGGL((merge_single_kernel<12, true>), dim3(1), dim3(MERGE_WG), lds, st, a);
        else
            hipLaunchKernelGGL((merge_single_kernel<34, true>), dim3(1), dim3(MERGE_WG), lds, st, a);
    } else if (a.K + 1 <= 12) {
        hipLaunchKernelGGL((merge_single_kernel<12, false>), dim3(1), dim3(MERGE_WG), lds, st, a);
    } else {
        hipLaunchKernelGGL((merge_single_kernel<34, false>), dim3(1), dim3(MERGE_WG), lds, st, a);
    }
}

}  // namespace icem
