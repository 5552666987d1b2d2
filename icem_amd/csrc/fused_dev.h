// fused_dev.h -- device-side building blocks shared by the f32 throughput kernels (k_*.hip): packed (cost, index)
// keys and the DPP / permlane bitonic sort, the per-row colored-noise sampler (sample_row), the 16-trajectory
// matrix-pipe rollout tile (Tile16), per-workgroup candidate-list merging, and the global top-K selection
// (merge_select*).  Everything here is __device__ __forceinline__ in an anonymous namespace: every translation unit
// gets its own copy, and all of them produce the same bits for the same inputs (the shard- and path-invariance tests
// rely on that).  Internal; not part of the public ABI.
#pragma once
#include "icem_fused.h"

#include <climits>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <type_traits>
#include <utility>

#include "exchange_dev.h"
#include "philox.h"
#include "refit.h"

// shapes (H, D, O) with a compiled matrix-pipe rollout; anything else runs on the generic kernels
#ifndef ICEM_FAST_SHAPES   // (a development build may narrow the list: ICEM_DEV_SHAPES=30,6,17 python -m icem_amd.build)
#define ICEM_FAST_SHAPES(X) X(30, 6, 17) X(30, 6, 18) X(12, 6, 17) X(13, 4, 17) X(30, 17, 24)
#endif
// horizons with a compiled folded sampler
#define ICEM_FAST_HORIZONS(X) X(30) X(12) X(13) X(10)

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
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)x, (int)(unsigned)x, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(x >> 32), (int)(unsigned)(x >> 32), CTRL, 0xF, 0xF, true);
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

// which lanes keep the SMALLER key of their pair in the (SIZE, J) compare-exchange step: block (lane & SIZE) == 0 sorts
// ascending, lane (lane & J) == 0 is the pair's lower one -- a compile-time 64-bit lane mask.  The step then is: the compare's
// lane mask XOR this constant on the scalar unit, and one select per word under the resulting mask (inverse ballot) -- where
// working the two predicates out per lane cost three vector instructions a step and scalar registers spilled around them
// (the 21 steps of the selection's threshold sort were 0.6 us of a lone wave: EXPERIMENTS R6.21).  Same compares, same result.
template <int SIZE, int J>
constexpr unsigned long long take_min_lanes() {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((l & SIZE) == 0) == ((l & J) == 0)) m |= 1ull << l;
    return m;
}
// LANE_MASK: the predicate applied as the compile-time lane mask (above) -- the default; false: worked out per lane, the form the
// Door / Relocate / FetchPickAndPlace tile kernel keeps for its one slab sort per tile: with the lane-mask form in it the
// Relocate launch ran 40 -> 52 us (EXPERIMENTS R6.21; same instruction counts, no spills: not understood, measured twice).
template <int SIZE, int J, bool LANE_MASK = true>
__device__ __forceinline__ unsigned long long bitonic_step(unsigned long long k, int lane) {
    const unsigned long long o = xor_partner<J>(k, lane);
    if constexpr (LANE_MASK) {
        constexpr unsigned long long TAKE_MIN = take_min_lanes<SIZE, J>();
        const unsigned long long o_less = __builtin_amdgcn_ballot_w64(o < k);
        const bool take_o = __builtin_amdgcn_inverse_ballot_w64(~(o_less ^ TAKE_MIN));   // take_min == o_less
        return take_o ? o : k;
    } else {
        const bool up = (lane & SIZE) == 0;  // this block sorts ascending
        const bool lower = (lane & J) == 0;  // this lane keeps the smaller of the pair
        const bool take_min = (up == lower);
        const bool o_less = o < k;
        return (take_min == o_less) ? o : k;
    }
}

template <int SIZE, int J, bool LANE_MASK = true>
__device__ __forceinline__ unsigned long long bitonic_merge(unsigned long long k, int lane) {
    k = bitonic_step<SIZE, J, LANE_MASK>(k, lane);
    if constexpr (J > 1) k = bitonic_merge<SIZE, J / 2, LANE_MASK>(k, lane);
    return k;
}

// ascending bitonic sort of one key per lane across the 64-lane wave (21 compare-exchange steps)
template <bool LANE_MASK = true>
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long k, int lane) {
    k = bitonic_merge<2, 1, LANE_MASK>(k, lane);
    k = bitonic_merge<4, 2, LANE_MASK>(k, lane);
    k = bitonic_merge<8, 4, LANE_MASK>(k, lane);
    k = bitonic_merge<16, 8, LANE_MASK>(k, lane);
    k = bitonic_merge<32, 16, LANE_MASK>(k, lane);
    k = bitonic_merge<64, 32, LANE_MASK>(k, lane);
    return k;
}

// ... when only lanes < NKEYS (16 or 32) hold keys and every other lane holds KEY_SENTINEL (the maximum): block 0 of
// every stage sorts ascending, so after the stages up to NKEYS the later ones would move nothing -- 10 or 15 steps
template <int NKEYS, bool LANE_MASK = true>
__device__ __forceinline__ unsigned long long wave_sort_first(unsigned long long k, int lane) {
    static_assert(NKEYS == 16 || NKEYS == 32 || NKEYS == 64, "keys in the first 16 / 32 / 64 lanes");
    k = bitonic_merge<2, 1, LANE_MASK>(k, lane);
    k = bitonic_merge<4, 2, LANE_MASK>(k, lane);
    k = bitonic_merge<8, 4, LANE_MASK>(k, lane);
    k = bitonic_merge<16, 8, LANE_MASK>(k, lane);
    if constexpr (NKEYS >= 32) k = bitonic_merge<32, 16, LANE_MASK>(k, lane);
    if constexpr (NKEYS >= 64) k = bitonic_merge<64, 32, LANE_MASK>(k, lane);
    return k;
}

// n (wave-uniform) keys in the first lanes, sentinels behind: the shortest network that sorts them
template <bool LANE_MASK = true>
__device__ __forceinline__ unsigned long long wave_sort_n(unsigned long long k, int lane, unsigned n) {
    if (n <= 16) return wave_sort_first<16, LANE_MASK>(k, lane);
    if (n <= 32) return wave_sort_first<32, LANE_MASK>(k, lane);
    return wave_sort64<LANE_MASK>(k, lane);
}

// the same network on 32-bit keys (half the moves): used where only the ORDER STATISTIC of the keys' cost halves matters
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned x) {
    // (bound_ctrl: every lane of these permutations has a source lane, so no "old" value to preload the destination with)
    return (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, true);
}
template <int J>
__device__ __forceinline__ unsigned xor_partner32(unsigned k, int lane) {
    if constexpr (J == 1) {
        return dpp_u32<0xB1>(k);
    } else if constexpr (J == 2) {
        return dpp_u32<0x4E>(k);
    } else if constexpr (J == 4) {
        const unsigned up = dpp_u32<0x12C>(k), down = dpp_u32<0x124>(k);
        return (lane & 4) ? down : up;
    } else if constexpr (J == 8) {
        return dpp_u32<0x128>(k);
    } else if constexpr (J == 16) {
        auto l = __builtin_amdgcn_permlane16_swap(k, k, false, false);
        return (lane & 16) ? l[0] : l[1];
    } else {
        static_assert(J == 32, "xor distance");
        auto l = __builtin_amdgcn_permlane32_swap(k, k, false, false);
        return (lane & 32) ? l[0] : l[1];
    }
}
template <int SIZE, int J>
__device__ __forceinline__ unsigned bitonic_merge32(unsigned k, int lane) {
    const unsigned o = xor_partner32<J>(k, lane);
    constexpr unsigned long long TAKE_MIN = take_min_lanes<SIZE, J>();
    const unsigned long long o_less = __builtin_amdgcn_ballot_w64(o < k);
    k = __builtin_amdgcn_inverse_ballot_w64(~(o_less ^ TAKE_MIN)) ? o : k;   // take_min == (o < k)
    if constexpr (J > 1) k = bitonic_merge32<SIZE, J / 2>(k, lane);
    return k;
}
__device__ __forceinline__ unsigned wave_sort64_u32(unsigned k, int lane) {
    k = bitonic_merge32<2, 1>(k, lane);
    k = bitonic_merge32<4, 2>(k, lane);
    k = bitonic_merge32<8, 4>(k, lane);
    k = bitonic_merge32<16, 8>(k, lane);
    k = bitonic_merge32<32, 16>(k, lane);
    k = bitonic_merge32<64, 32>(k, lane);
    return k;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------------
// K1
// -------------------------------------------------------------------------------------------------
// The h colored samples of one (trajectory, action-dim) row: per-row RNG stream -> Box-Muller -> h white
// draws in registers (row_normals) -> inverse real DFT folded on its symmetry (row_synth); emit(t, y) receives sample
// y of step t.  sample_row is the two in sequence; kernels that have to wait for the distribution between the two
// (merge prologue of k_sample.hip's sample_folded_merge_kernel) call them separately -- same operations in the same order, same bits.
template <int H, int ROUNDS>
__device__ __forceinline__ void row_normals(unsigned gi, unsigned j, unsigned off_lo, unsigned off_hi, unsigned seed_lo,
                                            unsigned seed_hi, float (&g)[HMAX]) {
    static_assert(H <= 32 && H >= 2, "white draws of a row live in 32 registers");
    Xoshiro128pp rng = row_stream<ROUNDS>(gi, j, off_lo, off_hi, seed_lo, seed_hi);
#pragma unroll
    for (int m = 0; m < H; m += 2) {
        const uint32_t xa = rng.next();
        const uint32_t xb = rng.next();
        box_muller(xa, xb, g[m], g[m + 1]);
    }
}

template <int H, typename Emit>
__device__ __forceinline__ void row_synth(const float* __restrict__ W, const float (&g)[HMAX], Emit&& emit, bool white) {
    constexpr int F = H / 2 + 1;
    if (white) {  // wave-uniform: noise_beta <= 0, the draws are the samples (icem.py:77)
#pragma unroll
        for (int t = 0; t < H; ++t) emit(t, g[t]);
        return;
    }
    // Even H: the half-period shift t -> t + H/2 multiplies frequency k's cosine and sine by (-1)^k, and the partial sums
    // over even and odd k are what the two accumulators of each part hold anyway (e0 / e1, o1 / o0).  So the table row of
    // t' also yields the samples at H/2 - t' and H/2 + t': four outputs per row instead of two, half the rows (and half
    // the scalar table loads).  t = H/2 comes with t = 0.
    constexpr bool HALF = H % 2 == 0;
    constexpr int Q = H / 2;
    {  // t = 0: every sine is zero
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int m = 0; m < F; m += 2) {
            e0 = __builtin_fmaf(g[m], W[m], e0);
            if (m + 1 < F) e1 = __builtin_fmaf(g[m + 1], W[m + 1], e1);
        }
        emit(0, e0 + e1);
        if (HALF) emit(Q, e0 - e1);
    }
    auto row = [&](int tp, bool four) {
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
        if (four) {
            const float ed = e0 - e1, dd = o0 - o1;
            emit(Q - tp, ed + dd);
            emit(Q + tp, ed - dd);
        }
    };
    if constexpr (HALF) {
#pragma unroll 1
        for (int tp = 1; tp <= (Q - 1) / 2; ++tp) row(tp, true);
        if (Q % 2 == 0) row(Q / 2, false);  // its own half-period partner
    } else {
#pragma unroll 1
        for (int tp = 1; tp <= H / 2; ++tp) row(tp, false);
    }
}

template <int H, int ROUNDS, typename Emit>
__device__ __forceinline__ void sample_row(const float* __restrict__ W, unsigned gi, unsigned j, unsigned off_lo,
                                           unsigned off_hi, unsigned seed_lo, unsigned seed_hi, Emit&& emit,
                                           bool white = false) {
    float g[HMAX];
    row_normals<H, ROUNDS>(gi, j, off_lo, off_hi, seed_lo, seed_hi, g);
    row_synth<H>(W, g, emit, white);
}

// ---- one row on FOUR lanes (small populations) ---------------------------------------------------------------------
// With fewer rows than lanes on the chip a row's ~1700 instructions are a latency chain (a lone wave issues one
// instruction per ~5 cycles).  Here the four lanes of a quad share one (trajectory, dim) row: every lane runs the row's
// word stream (the generator is serial), transforms only ITS Box-Muller pairs (pair 4k + q in group k), the quad
// exchanges the draws through DPP quad broadcasts, and each lane synthesises a quarter of the outputs (t = 0 on lane 3,
// the symmetric pairs tp = q + 1, q + 5, ... on lane q) with its table rows read from an LDS copy (per-lane row, so no
// scalar loads).  Same operations per draw and per output as row_normals / row_synth: same bits.
constexpr int WQ_STRIDE = HMAX + 4;  // floats per table row in LDS: the four rows a quad reads land in different banks

template <int JQ>
__device__ __forceinline__ float quad_bcast(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), JQ * 0x55, 0xF, 0xF, false));
}

template <int H, int ROUNDS>
__device__ __forceinline__ void row_normals_quad(unsigned gi, unsigned j, unsigned off_lo, unsigned off_hi, unsigned seed_lo,
                                                 unsigned seed_hi, int q, float (&g)[HMAX]) {
    constexpr int NP = (H + 1) / 2, NG = (NP + 3) / 4;
    Xoshiro128pp rng = row_stream<ROUNDS>(gi, j, off_lo, off_hi, seed_lo, seed_hi);
    float ga[NG], gb[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        uint32_t xa = 0, xb = 0;
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
            if (4 * k + jq < NP) {
                const uint32_t wa = rng.next();
                const uint32_t wb = rng.next();
                xa = q == jq ? wa : xa;
                xb = q == jq ? wb : xb;
            }
        }
        box_muller(xa, xb, ga[k], gb[k]);  // (a lane without a pair in the last group transforms (0, 0): finite, unused)
    }
#pragma unroll
    for (int k = 0; k < NG; ++k) {
#define ICEM_QB(JQ)                                                        \
    if (4 * k + JQ < NP) {                                                 \
        g[2 * (4 * k + JQ)] = quad_bcast<JQ>(ga[k]);                       \
        if (2 * (4 * k + JQ) + 1 < HMAX) g[2 * (4 * k + JQ) + 1] = quad_bcast<JQ>(gb[k]); \
    }
        ICEM_QB(0) ICEM_QB(1) ICEM_QB(2) ICEM_QB(3)
#undef ICEM_QB
    }
}

// Wl: rows 0 .. H/2 of the synthesis table in LDS, WQ_STRIDE floats apart
template <int H, typename Emit>
__device__ __forceinline__ void row_synth_quad(const float* Wl, const float (&g)[HMAX], int q, Emit&& emit, bool white) {
    constexpr int F = H / 2 + 1;
    if (white) {  // wave-uniform: the draws are the samples (icem.py:77)
#pragma unroll
        for (int t = 0; t < H; ++t)
            if ((t & 3) == q) emit(t, g[t]);
        return;
    }
    constexpr bool HALF = H % 2 == 0;   // four outputs per table row, as row_synth
    constexpr int Q = H / 2;
    if (q == 3) {  // t = 0 (and t = H/2): every sine is zero
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int m = 0; m < F; m += 2) {
            e0 = __builtin_fmaf(g[m], Wl[m], e0);
            if (m + 1 < F) e1 = __builtin_fmaf(g[m + 1], Wl[m + 1], e1);
        }
        emit(0, e0 + e1);
        if (HALF) emit(Q, e0 - e1);
    }
    auto row = [&](int tp, bool four) {
        const float* w = Wl + tp * WQ_STRIDE;
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
        if (four) {
            const float ed = e0 - e1, dd = o0 - o1;
            emit(Q - tp, ed + dd);
            emit(Q + tp, ed - dd);
        }
    };
    if constexpr (HALF) {
#pragma unroll 1
        for (int tp = q + 1; tp <= (Q - 1) / 2; tp += 4) row(tp, true);
        if (Q % 2 == 0 && q == ((Q - 1) / 2) % 4) row(Q / 2, false);  // the lane behind the last four-output row
    } else {
#pragma unroll 1
        for (int tp = q + 1; tp <= H / 2; tp += 4) row(tp, false);
    }
}

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
    static constexpr bool LONE_STEP = false;   // (Tile16H's hand-ordered step for a lone wave: none here)
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
    template <bool PLANES_FIRST = false>   // (Tile16H's issue-order flag: nothing to reorder here, f32 MFMAs and the vector pipe are one)
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

// ---- Tile16H: the same tile with the model step on the 16-BIT matrix cores ------------------------------------------
// (EXPERIMENTS R5.1.)  Tile16's six v_mfma_f32_16x16x4_f32 are 192 cycles of the one pipe f32 MFMA and f32 VALU share
// (64 flop / clk / SIMD); v_mfma_f32_16x16x32_f16 runs at sixteen times that rate.  Every f32 operand x is carried as two
// fp16 numbers, x S = hi + lo with hi = f16(x S), lo = f16(x S - hi), both round-to-nearest-even (22 significant bits
// and lo's sign: x S to 2^-24 relative while lo is a normal fp16 number, i.e. |x S| >= 2^-3; to 2^-25 absolute below),
// and a multiply-add is the three products hi.Hi + hi.Lo + lo.Hi -- each exact in the f32 accumulator's format -- summed
// smallest first; lo.Lo (< 2^-22 of hi.Hi) is dropped.  The whole contraction of a step (16 tile columns + the extra
// entries: columns >= 16 and the actions, <= 32 in all) is ONE 32-deep MFMA per product: 3 x 16 cycles.
//   Layout: A = M^T (output column i = lane % 16, contraction slots 8g .. 8g+7 of lane group g = lane / 16), B = X^T
// (trajectory j = lane % 16, the same slots).  Slots 8g .. 8g+3 are the tile columns 4g .. 4g+3 -- the lane's own four
// accumulator registers of the previous step, as in Tile16: no cross-lane traffic -- and slots 8g+4 .. 8g+7 the extra
// entries e = 4q + g, q = 0 .. 3 (Tile16's assignment: entry e of a step sits at rd[4q] of lane group g).
//   S is ONE power of two per launch, the same in every wave, rank and tile: 2^(4 - e) with 2^e <= m < 2^(e+1),
// m = max(|obs0| entries, FastRolloutArgs::act_mag (the action bound's magnitude), 1 for a tanh model) -- from the start
// observation every trajectory shares, so a trajectory's bits do not depend on which tile, launch or GPU rolls it out.
// The model's planes are made at load time from A sM and B sB, sM / sB the powers of two (FastRolloutArgs::m_scale / b_scale,
// from the host: plan.hip) that put the largest |entry| of A / of B into [64, 128): entries down to 2^-10 of the largest keep
// normal lo planes (an UNSCALED model's small entries -- 0.05 beside 0.95 -- sit in fp16's subnormals, a systematic 3e-8
// per entry and step that adds up linearly over the horizon: measured 1e-5 of a cost at h = 30 where the actions dominate the
// state).  B's scale is free (the actions enter at T / sB); A's costs the four v_mul that bring the accumulators (T = S sM
// times the state) back to the B operand's scale S.  The state is KEPT at T (powers of two commute with every f32 operation
// here; the cost's constants are scaled once instead); states growing beyond 2^11 m overflow fp16's range and the
// trajectory's cost becomes NaN -> +inf in its key (it is dropped; an f32 chain would report a huge finite cost).  Output
// columns >= 16 stay on the VALU (f32 fmaf chain over the lane's entries + permlane reduction, as Tile16).  NOT the bits of Tile16 / Tile4: a handle rolls out in ONE
// arithmetic (FastRolloutArgs::arith), chosen from its configuration, never from a launch's row count.
__device__ __forceinline__ unsigned wave_min_u32(unsigned x);   // (below, with the merge's reductions)
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// (a, b) -> packed fp16 (hi_a | hi_b << 16) and the packed residuals: v_cvt_pk_f16_f32, 2 x v_fma_mix_f32 (the f32 value minus
// one half of the packed pair, no unpacking), v_cvt_pk_f16_f32.  The residual x - f16(x) is exact in f32.
__device__ __forceinline__ void split_pair_f16(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2_t h = __builtin_convertvector(f32x2_t{a, b}, f16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    float r0, r1;
    const float one = 1.f;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a), "v"(one), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(b), "v"(one), "v"(hi));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
}
// the same for (a sa, b sb), sa / sb powers of two (or 0): two v_mul_f32 in front of the conversion (v_fma_mixlo / mixhi_f16
// would fold the scale into it, but issue at a THIRD of v_mul's rate: tools/ubench/tile_ops_rates.hip), the residual's
// v_fma_mix_f32 takes the scale into its own multiply
__device__ __forceinline__ void split_pair_f16_scaled(float a, float sa, float b, float sb, unsigned& hi, unsigned& lo) {
    const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a * sa, b * sb}, f16x2_t));
    hi = h;
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a), "v"(sa), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(b), "v"(sb), "v"(h));
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{r0, r1}, f16x2_t));
}
__device__ __forceinline__ f32x4 mfma_f16_32(const unsigned (&a)[4], const unsigned (&b)[4], f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, u32x4_t{a[0], a[1], a[2], a[3]}),
                                                  __builtin_bit_cast(f16x8_t, u32x4_t{b[0], b[1], b[2], b[3]}), c, 0, 0, 0);
}

template <int H, int D, int O, int KIND>
struct Tile16H {
    static constexpr int NT = 1;
    static constexpr int OP = O;
    static constexpr int REM = O > 16 ? O - 16 : 0;
    static constexpr int NX = REM + D;
    static constexpr int NKX = (NX + 3) / 4;           // extra entries per lane (slots 8g+4 ..)
    static constexpr int CT4 = ((O + 3) / 4) * 4;
    static constexpr int SLACK = 4, TAIL = 8;
    static constexpr int ZROW = OP + D;
    static_assert(O >= 16 && O <= 20 && NKX <= 4, "one 16-column tile, at most 16 extra contraction entries");
    static constexpr bool LONE_STEP = REM == 1 && NKX == 2;   // shapes step_lone() is written for (o = 17, d <= 7)

    unsigned aH[4], aL[4];              // model operand planes: slots 8g .. 8g+7 as four fp16 pairs
    float wR[REM > 0 ? REM : 1][4 + NKX];   // f32 weights of this lane's entries into output column 16 + r (the extras' scaled in load_obs)
    float cw[NKX], sc[NKX];             // ctrl_w where the extra entry is an action, else 0; the entry's scale into the B operand:
                                        // S (an action), 1 (a state entry: kept scaled) or 0 (padding)
    bool is_act[NKX];
    int xoff[NKX];                      // where this lane reads extra entry q of a step, relative to its read pointer: 4q, or -- a
                                        // padding entry (zero weight, zero scale) -- the step's first action: always staged, always finite
    f32x4 obs_init;
    float rem_init[REM > 0 ? REM : 1];
    int perm_base[4], perm_rem[REM > 0 ? REM : 1];
    float pen, lin_w, ksum, flip_th, T, invT, invM, sM, sB, act_mag;
    bool ang_is_col1, use_min;
    int g;

    __device__ __forceinline__ void load(const FastRolloutArgs& a, int lane) {
        const int j = lane & 15;
        g = lane >> 4;
        sM = a.m_scale;
        sB = a.b_scale;
        float m[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 4 * g + s;
            m[s] = a.Mp[k * CT4 + j] * sM;
#pragma unroll
            for (int r = 0; r < REM; ++r) wR[r][s] = a.Mp[k * CT4 + 16 + r];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * q + g;
            const bool valid = q < NKX && e < NX;
            const int k = !valid ? ZROW : (e < REM ? 16 + e : OP + (e - REM));
            m[4 + q] = a.Mp[k * CT4 + j] * ((valid && e >= REM) ? sB : sM);   // an action's row of B, or a state row of A
            if (q < NKX) {
#pragma unroll
                for (int r = 0; r < REM; ++r) wR[r][4 + q] = a.Mp[k * CT4 + 16 + r];
                is_act[q] = valid && e >= REM;
                cw[q] = is_act[q] ? a.ctrl_w : 0.f;
                sc[q] = valid ? (is_act[q] ? 0.f : 1.f / sM) : 0.f;   // (actions: set in load_obs)
                xoff[q] = valid ? 4 * q : REM - g;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) split_pair_f16(m[2 * p], m[2 * p + 1], aH[p], aL[p]);
#pragma unroll
        for (int v = 0; v < 4; ++v) perm_base[v] = a.perm[4 * g + v];
#pragma unroll
        for (int r = 0; r < REM; ++r) perm_rem[r] = a.perm[16 + r];
        pen = (a.flip_col >= 0 && g == 0) ? a.flip_pen : 0.f;
        lin_w = g == 0 ? a.lin_w : 0.f;
        ang_is_col1 = a.flip_col == 1;
        ksum = a.cost_mode == 0 ? 1.f : 0.f;
        use_min = a.cost_mode == 1;
        flip_th = a.flip_th;
        act_mag = a.act_mag;
    }

    // obs: 32 floats in LDS (natural order, zeros behind).  Also fixes the launch's power-of-two scale S.
    __device__ __forceinline__ void load_obs(const float* obs) {
        const int lane = (int)(threadIdx.x & 63);
        float mx = __builtin_fabsf(obs[lane & 31]);
        mx = mx != mx ? 0.f : mx;   // (a NaN observation poisons every cost anyway: keep S finite)
        // max over the wave: non-negative floats order like their bit patterns
        const unsigned mb = ~wave_min_u32(~__float_as_uint(mx));
        float mm = __uint_as_float(mb);
        mm = mm > act_mag ? mm : act_mag;
        if (KIND == 1) mm = mm > 1.f ? mm : 1.f;
        int ex = (int)((__float_as_uint(mm) >> 23) & 0xFF) - 127;   // 2^ex <= mm < 2^(ex+1); mm == 0 or subnormal: -127
        ex = ex < -100 ? -100 : (ex > 100 ? 100 : ex);
        // (wave-uniform values: pinned to scalar registers -- this tile lives on the 128 registers a wave has at four per SIMD)
        auto uni = [](float x) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
        const float S = __uint_as_float((unsigned)(127 + 4 - ex) << 23);   // the B operand's scale: S x in [16, 32) at most
        T = uni(S * sM);                                                    // the accumulators' (and the kept state's): S sM x
        invT = uni(__uint_as_float((unsigned)(127 - 4 + ex) << 23) / sM);
        invM = uni(1.f / sM);
        flip_th = uni(flip_th);
        ksum = uni(ksum);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            int pb = perm_base[v];
            asm volatile("" : "+v"(pb));
            obs_init[v] = obs[pb] * T;
        }
#pragma unroll
        for (int r = 0; r < REM; ++r) {
            int pb = perm_rem[r];
            asm volatile("" : "+v"(pb));
            rem_init[r] = obs[pb] * T;
        }
        flip_th *= T;
        lin_w *= invT;
#pragma unroll
        for (int q = 0; q < NKX; ++q) {
            if (is_act[q]) {
                sc[q] = T / sB;   // (B sB) x (a T / sB) = T B a: the action's contribution at the accumulators' scale
#pragma unroll
                for (int r = 0; r < REM; ++r) wR[r][4 + q] *= T;
            }
        }
    }

    __device__ __forceinline__ const float* read_ptr(const float* buf, int lane, int stride) const {
        return buf + SLACK + (lane & 15) * stride + ((lane >> 4) - REM);
    }

    struct State {
        f32x4 cur[1];                    // T x the lane's four tile columns (T = S sM, the accumulators' scale)
        float xr[REM > 0 ? REM : 1];     // T x column 16 + r
        float acc_s, acc_b;
    };
    __device__ __forceinline__ void init(State& st) const {
        st.cur[0] = obs_init;
#pragma unroll
        for (int r = 0; r < REM; ++r) st.xr[r] = rem_init[r];
        st.acc_s = 0.f;
        st.acc_b = INFINITY;
    }
    // PLANES_FIRST: the operand planes and the MFMAs in front of the step's cost block (which needs the OLD state only and then
    // sits in the MFMAs' shadow) -- same operations, same bits, another issue order.  For the kernels with a lone tile wave per
    // SIMD and registers to spare (the single-launch kernel: c2 61.1 -> 60.4 us per MPC step, EXPERIMENTS R6.11); on the 128
    // registers of the noise-ahead launch it costs two more spilled registers and moves nothing, so that one keeps the old order.
    template <bool PLANES_FIRST = false>
    __device__ __forceinline__ void step(State& st, const float* rd) const {
        float xv[NKX];
#pragma unroll
        for (int q = 0; q < NKX; ++q) {
            float v = rd[xoff[q]];   // an action, a padding entry's finite stand-in, or (a state entry's lane) anything: replaced
#pragma unroll
            for (int r = 0; r < REM; ++r)
                if (r / 4 == q) v = (g == r % 4) ? st.xr[r] : v;
            xv[q] = v;
        }
        f32x4 nxt;
        auto model_step = [&]() {
            // the B operand planes: own columns, then the extras (x their scale)
            unsigned bH[4], bL[4];
            split_pair_f16_scaled(st.cur[0][0], invM, st.cur[0][1], invM, bH[0], bL[0]);
            split_pair_f16_scaled(st.cur[0][2], invM, st.cur[0][3], invM, bH[1], bL[1]);
            if (NKX >= 2) split_pair_f16_scaled(xv[0], sc[0], xv[NKX >= 2 ? 1 : 0], sc[NKX >= 2 ? 1 : 0], bH[2], bL[2]);
            else split_pair_f16_scaled(xv[0], sc[0], 0.f, 0.f, bH[2], bL[2]);
            if (NKX == 4) split_pair_f16_scaled(xv[NKX > 2 ? 2 : 0], sc[NKX > 2 ? 2 : 0], xv[NKX > 3 ? 3 : 0], sc[NKX > 3 ? 3 : 0], bH[3], bL[3]);
            else if (NKX == 3) split_pair_f16_scaled(xv[NKX > 2 ? 2 : 0], sc[NKX > 2 ? 2 : 0], 0.f, 0.f, bH[3], bL[3]);
            else bH[3] = bL[3] = 0u;
            nxt = f32x4{0.f, 0.f, 0.f, 0.f};
            nxt = mfma_f16_32(aL, bH, nxt);
            nxt = mfma_f16_32(aH, bL, nxt);
            nxt = mfma_f16_32(aH, bH, nxt);
        };
        if (PLANES_FIRST) model_step();
        // step cost (Tile16's, on the scaled state: flip_th and lin_w carry S).  [ang > th] + [ang < -th] = [|ang| > th]
        // for th >= 0 (update_paths keeps handles with a negative threshold on the exact tile)
        const float ang = ang_is_col1 ? st.cur[0][1] : st.cur[0][0];
        float c = (__builtin_fabsf(ang) > flip_th) ? pen : 0.f;
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
            for (int q = 0; q < NKX; ++q) p = __builtin_fmaf(xv[q], wR[r][4 + q], p);
            pr[r] = p;
        }
        if (REM >= 1)
            reduce_groups_pair(pr[0], c);
        else
            c = reduce_groups(c);
#pragma unroll
        for (int r = 1; r < REM; ++r) pr[r] = reduce_groups(pr[r]);
        st.acc_s = __builtin_fmaf(st.acc_s, ksum, c);
        st.acc_b = __builtin_fminf(c, st.acc_b);   // (a NaN step cost is skipped, as by Tile16's compare)
        if (!PLANES_FIRST) model_step();
        if (KIND == 1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) st.cur[0][v] = fast_tanh(nxt[v] * invT) * T;
#pragma unroll
            for (int r = 0; r < REM; ++r) st.xr[r] = fast_tanh(pr[r] * invT) * T;
        } else {
            st.cur[0] = nxt;
#pragma unroll
            for (int r = 0; r < REM; ++r) st.xr[r] = pr[r];
        }
    }
    // The step as a LONE wave wants it issued (one tile per CU: the single-launch kernel at the metric's population).  A lone
    // wave pays 3.5 ns for an instruction that waits for the one in front of it and 2.0-2.2 ns for one of four interleaved
    // chains (tools/ubench/lone_wave_ilp.hip), and the compiler, which schedules for a SIMD full of waves, leaves the step's
    // chains one behind the other: the 17th column's six dependent fmas, the cost's seven, the planes.  Here they are written
    // INTERLEAVED -- the column's chain (the step's longest: six fmas, the reduction, the select, the plane of the next step) as
    // the spine, a plane and a cost instruction beside each link -- and fenced (sched_barrier) so that the order survives.
    // Same operations on the same operands in the same per-value order as step(): the same bits.  `raw`: the step's two extra
    // entries as read from the tile, replaced by the next step's (requested at the top: nothing crosses a fence).
    __device__ __forceinline__ void step_lone(State& st, float (&raw)[NKX], const float* rd_next) const {
        static_assert(REM == 1 && NKX == 2, "written for one extra column and two extra entries per lane");
#define ICEM_FENCE() __builtin_amdgcn_sched_barrier(0)
        const float c0 = st.cur[0][0], c1 = st.cur[0][1], c2 = st.cur[0][2], c3 = st.cur[0][3];
        const float xv0 = (g == 0) ? st.xr[0] : raw[0], xv1 = raw[1];
        const float n0 = rd_next[xoff[0]], n1 = rd_next[xoff[1]];
        auto cvt = [](f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t)); };
        auto res = [](float x, float s_, unsigned h, bool upper) {
            float r;
            if (upper) asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s_), "v"(h));
            else asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(s_), "v"(h));
            return r;
        };
        // link 1
        const f32x2_t m01 = f32x2_t{c0 * invM, c1 * invM}, m23 = f32x2_t{c2 * invM, c3 * invM};
        float p = __builtin_fmaf(c0, wR[0][0], 0.f);
        const float ang = ang_is_col1 ? c1 : c0;
        ICEM_FENCE();
        // link 2
        unsigned bH[4], bL[4];
        bH[0] = cvt(m01);
        bH[1] = cvt(m23);
        p = __builtin_fmaf(c1, wR[0][1], p);
        const bool flipped = __builtin_fabsf(ang) > flip_th;
        ICEM_FENCE();
        // link 3
        const f32x2_t m45 = f32x2_t{xv0 * sc[0], xv1 * sc[1]};
        p = __builtin_fmaf(c2, wR[0][2], p);
        float c = flipped ? pen : 0.f;
        const float s0 = xv0 * xv0;
        ICEM_FENCE();
        // link 4
        bH[2] = cvt(m45);
        bH[3] = 0u;
        p = __builtin_fmaf(c3, wR[0][3], p);
        c = __builtin_fmaf(s0, cw[0], c);
        const float s1 = xv1 * xv1;
        ICEM_FENCE();
        f32x4 nxt = mfma_f16_32(aL, bH, f32x4{0.f, 0.f, 0.f, 0.f});
        // link 5
        const float r0 = res(c0, invM, bH[0], false), r1 = res(c1, invM, bH[0], true);
        p = __builtin_fmaf(xv0, wR[0][4], p);
        c = __builtin_fmaf(s1, cw[1], c);
        ICEM_FENCE();
        // link 6
        const float r2 = res(c2, invM, bH[1], false), r3 = res(c3, invM, bH[1], true);
        p = __builtin_fmaf(xv1, wR[0][5], p);
        c = __builtin_fmaf(lin_w, c0, c);
        ICEM_FENCE();
        // the planes' residuals beside the reduction's first exchange
        const float r4 = res(xv0, sc[0], bH[2], false), r5 = res(xv1, sc[1], bH[2], true);
        bL[0] = cvt(f32x2_t{r0, r1});
        bL[1] = cvt(f32x2_t{r2, r3});
        ICEM_FENCE();
        bL[2] = cvt(f32x2_t{r4, r5});
        bL[3] = 0u;
        nxt = mfma_f16_32(aH, bL, nxt);
        ICEM_FENCE();
        // (an MFMA that accumulates onto the one in front of it holds the wave's issue until that one is through: the
        // reduction's first exchange goes between the two) -- reduce_groups_pair(p, c), in two halves
        auto x32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(c), false, false);
        const float half = __uint_as_float(x32[0]) + __uint_as_float(x32[1]);
        ICEM_FENCE();
        nxt = mfma_f16_32(aH, bH, nxt);
        ICEM_FENCE();
        auto x16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(half), __float_as_uint(half), false, false);
        p = c = __uint_as_float(x16[0]) + __uint_as_float(x16[1]);
        st.acc_s = __builtin_fmaf(st.acc_s, ksum, c);
        st.acc_b = __builtin_fminf(c, st.acc_b);
        if (KIND == 1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) st.cur[0][v] = fast_tanh(nxt[v] * invT) * T;
            st.xr[0] = fast_tanh(p * invT) * T;
        } else {
            st.cur[0] = nxt;
            st.xr[0] = p;
        }
        raw[0] = n0;
        raw[1] = n1;
#undef ICEM_FENCE
    }
    __device__ __forceinline__ float cost(const State& st) const {
        const float v = use_min ? st.acc_b : st.acc_s;
        if (REM == 0) return v;
        auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        return __uint_as_float(r[1]);
    }
};

// which tile a kernel instantiation rolls out with: ARITH 0 = Tile16 (exact f32), 1 = Tile16H where it exists (one output tile)
template <int H, int D, int O, int KIND, int ARITH>
struct TileSel { using type = Tile16<H, D, O, KIND>; };
template <int H, int D, int O, int KIND>
struct TileSel<H, D, O, KIND, 1> {
    using type = typename std::conditional<(O <= 20), Tile16H<H, D, (O <= 20 ? O : 17), KIND>, Tile16<H, D, O, KIND>>::type;
};

// ---- Tile4: the same model step on the VALU, FOUR trajectories per wavefront ----------------------------------------
// For small populations (fewer 16-trajectory tiles than SIMDs) the rollout is a latency chain, and Tile16's chain is
// long: six DEPENDENT f32 MFMAs per step (40 cycles each) plus ~35 VALU for a lone wave = ~410 cycles.  Here a row of
// 16 lanes is one trajectory and lane c holds observation column c: the contraction is 24 v_fmac_f32 whose x operand
// arrives through the DPP row broadcast (v_fmac_f32_dpp ... row_newbcast:k: no move, no LDS), ~60 instructions per
// step, and four wavefronts share a tile's work.  An f32 MFMA is bitwise an fmaf chain over its four slots, so the SAME
// chain in the same order (k = 4g + s for s = 0..3, g = 0..3, then the extra entries) gives the SAME bits as Tile16:
// costs, elites and every downstream buffer are identical whichever tile shape a launch uses (tested).  O <= 20.
template <int K>
__device__ __forceinline__ float fmac_row_bcast(float acc, float x, float m) {
    // acc += x[lane K of this row of 16] * m   (fused)
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(K));
    return acc;
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
    return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xF, 0xF, false));
}
// (x_g + x_{g^2}) + (x_{g^1} + x_{g^3}) over the four quads g of a row -- Tile16's reduce_groups association
// (floating-point addition commutes bitwise) -- the result in every lane of the row
__device__ __forceinline__ float reduce_quads(float x) {
    const float s = x + dpp_f32<0x128>(x);  // row_ror:8
    return s + dpp_f32<0x124>(s);           // row_ror:4
}

template <int H, int D, int O, int KIND>
struct Tile4 {
    using T16 = Tile16<H, D, O, KIND>;
    static_assert(T16::NT == 1, "one 16-column tile (O <= 20)");
    static constexpr int REM = T16::REM, NKX = T16::NKX, NX = T16::NX, CT4 = T16::CT4, OP = T16::OP, ZROW = T16::ZROW;
    static constexpr int NE = 4 * NKX;     // extra contraction entries incl. padding (columns >= 16, then the actions)
    float mK[16];                          // M[k][c] for the 16 tile columns k, this lane's output column c
    float mE[NE];                          // M[row(e)][c] for the extra entries
    float wq[REM > 0 ? REM : 1][4];        // M[4g + s][16 + r]: this quad's share of extra output column r
    float wx[REM > 0 ? REM : 1][NKX];      // M[row(4q + g)][16 + r]
    float cw[NKX];
    bool is_act[NKX];
    float x_init, rem_init[REM > 0 ? REM : 1];
    int perm_c, perm_rem[REM > 0 ? REM : 1];
    float pen, lin_w, ksum, flip_th;
    bool ang_is_col1, use_min;
    int g, c;

    static __device__ __forceinline__ int entry_row(int e) { return e >= NX ? ZROW : (e < REM ? 16 + e : OP + (e - REM)); }

    __device__ __forceinline__ void load(const FastRolloutArgs& a, int lane) {
        c = lane & 15;
        g = c >> 2;
#pragma unroll
        for (int k = 0; k < 16; ++k) mK[k] = a.Mp[k * CT4 + c];
#pragma unroll
        for (int e = 0; e < NE; ++e) mE[e] = a.Mp[entry_row(e) * CT4 + c];
#pragma unroll
        for (int r = 0; r < REM; ++r) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wq[r][s] = a.Mp[(4 * g + s) * CT4 + 16 + r];
#pragma unroll
            for (int q = 0; q < NKX; ++q) wx[r][q] = a.Mp[entry_row(4 * q + g) * CT4 + 16 + r];
        }
#pragma unroll
        for (int q = 0; q < NKX; ++q) {
            const int e = 4 * q + g;
            is_act[q] = e < NX && e >= REM;
            cw[q] = is_act[q] ? a.ctrl_w : 0.f;
        }
        perm_c = a.perm[c];
#pragma unroll
        for (int r = 0; r < REM; ++r) perm_rem[r] = a.perm[16 + r];
        pen = (a.flip_col >= 0 && g == 0) ? a.flip_pen : 0.f;
        lin_w = g == 0 ? a.lin_w : 0.f;
        ang_is_col1 = a.flip_col == 1;
        ksum = a.cost_mode == 0 ? 1.f : 0.f;
        use_min = a.cost_mode == 1;
        flip_th = a.flip_th;
    }
    __device__ __forceinline__ void load_obs(const float* obs) {
        x_init = obs[perm_c];
#pragma unroll
        for (int r = 0; r < REM; ++r) rem_init[r] = obs[perm_rem[r]];
    }
    // this lane's read pointer for its quad's extra entries (entry 4q + g at rd[4q], as Tile16::read_ptr) and the
    // trajectory's action row; row = the trajectory's row in an LDS tile of `stride` floats per trajectory
    __device__ __forceinline__ const float* read_ptr(const float* buf, int row, int stride) const {
        return buf + T16::SLACK + row * stride + (g - REM);
    }
    __device__ __forceinline__ const float* row_ptr(const float* buf, int row, int stride) const {
        return buf + T16::SLACK + row * stride;
    }
    struct State {
        float x;
        float xr[REM > 0 ? REM : 1];
        float acc_s, acc_b;
    };
    __device__ __forceinline__ void init(State& st) const {
        st.x = x_init;
#pragma unroll
        for (int r = 0; r < REM; ++r) st.xr[r] = rem_init[r];
        st.acc_s = 0.f;
        st.acc_b = INFINITY;
    }
    // rd: this quad's entries of the step (rd[4q]); ar: the trajectory's D actions of the step
    __device__ __forceinline__ void step(State& st, const float* rd, const float* ar) const {
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
        float ext[NE];  // the extra entries in contraction order, the same in every lane of the row
#pragma unroll
        for (int e = 0; e < NE; ++e) ext[e] = e < REM ? st.xr[e] : (e - REM < D ? ar[e - REM] : 0.f);
        // step cost: this quad's share (the columns 4g .. 4g+3 live in the quad's four lanes), then the sum over quads
        const float col0 = dpp_f32<0x00>(st.x);  // quad_perm [0,0,0,0]: column 4g
        const float col1 = dpp_f32<0x55>(st.x);  // quad_perm [1,1,1,1]: column 4g + 1
        const float ang = ang_is_col1 ? col1 : col0;
        float cst = 0.f;
        cst += (ang > flip_th) ? pen : 0.f;
        cst += (ang < -flip_th) ? pen : 0.f;
#pragma unroll
        for (int q = 0; q < NKX; ++q) cst = __builtin_fmaf(xv[q] * xv[q], cw[q], cst);
        cst = __builtin_fmaf(lin_w, col0, cst);
        float pr[REM > 0 ? REM : 1];
        if (REM > 0) {
            const float col2 = dpp_f32<0xAA>(st.x), col3 = dpp_f32<0xFF>(st.x);
            const float cq[4] = {col0, col1, col2, col3};
#pragma unroll
            for (int r = 0; r < REM; ++r) {
                float p = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) p = __builtin_fmaf(cq[s], wq[r][s], p);
#pragma unroll
                for (int q = 0; q < NKX; ++q) p = __builtin_fmaf(xv[q], wx[r][q], p);
                pr[r] = reduce_quads(p);
            }
        }
        cst = reduce_quads(cst);
        st.acc_s = __builtin_fmaf(st.acc_s, ksum, cst);
        st.acc_b = cst < st.acc_b ? cst : st.acc_b;
        // model step: one fmaf chain per output column in the matrix pipe's order: MFMA s (s = 0..3) contracts slots
        // g = 0..3 = columns 4g + s; then the MFMAs of the extra entries
        float nxt = 0.f;
        asm volatile("s_nop 1" ::: "memory");  // st.x was written by VALU: keep the DPP read two slots behind it
#define ICEM_T4(K) nxt = fmac_row_bcast<K>(nxt, st.x, mK[K]);
        ICEM_T4(0) ICEM_T4(4) ICEM_T4(8) ICEM_T4(12)
        ICEM_T4(1) ICEM_T4(5) ICEM_T4(9) ICEM_T4(13)
        ICEM_T4(2) ICEM_T4(6) ICEM_T4(10) ICEM_T4(14)
        ICEM_T4(3) ICEM_T4(7) ICEM_T4(11) ICEM_T4(15)
#undef ICEM_T4
#pragma unroll
        for (int e = 0; e < NE; ++e) nxt = __builtin_fmaf(mE[e], ext[e], nxt);
        st.x = act_fn(nxt, std::integral_constant<int, KIND>{});
#pragma unroll
        for (int r = 0; r < REM; ++r) st.xr[r] = act_fn(pr[r], std::integral_constant<int, KIND>{});
    }
    __device__ __forceinline__ float cost(const State& st) const { return use_min ? st.acc_b : st.acc_s; }
};

// a tile's 16 keys (lanes 0..15, the rest sentinels) join the wave's running sorted top-K: lanes 16..16+K-1 carry
// the running list, one sort
template <bool LANE_MASK = true>
__device__ __forceinline__ unsigned long long topk_push16(unsigned long long run_key, unsigned long long key, bool first,
                                                          int K, int lane) {
    if (!first) {
        const unsigned long long prev = __shfl(run_key, lane - 16, 64);
        if (lane >= 16 && lane < 16 + K) key = prev;
    }
    // keys in lanes 0..15 (first tile) or 0..15+K, sentinels behind: the shortest network that covers them
    return wave_sort_n<LANE_MASK>(key, lane, first ? 16u : 16u + (unsigned)K);
}

// one sorted list per workgroup out of the first WAVES waves' running lists (K <= 32 keys each, ascending in lanes 0..K-1):
// list `wg` of `n_wg`.  Every thread of the workgroup must call it.
// No sort: the lists go to LDS once, and the place of a key in the merged order is the number of smaller keys among the
// WAVES x K -- real keys are all different (a key embeds its row), and the empty slots are filled with keys that are all
// different too and lie behind every real one ((+inf, INT_MAX - 511 ..): they come out as the (+inf, INT_MAX) padding
// again).  Every wave counts for its own keys (lane = key + 16 x the quarter of the comparands it reads) and the keys
// with place < K go straight to the output.  One barrier and a few dozen independent compares per lane where the tree
// of 64-lane bitonic networks on 64-bit keys (8 -> 2 -> 1 lists) was two dependent sorts and three barriers -- at the
// end of EVERY rollout launch (EXPERIMENTS R4.13).
constexpr unsigned KEY_FILL_LO = 0x7FFFFE00u;   // fillers: (+inf, KEY_FILL_LO + slot), slot < 512 (16 waves x 32)
template <int WAVES>
__device__ __forceinline__ void wg_merge_emit(unsigned long long (*wg_keys)[WAVES][32], unsigned long long run_key, int K,
                                              int lane, int wave, const FastRolloutArgs& a, int wg = blockIdx.x,
                                              int n_wg = gridDim.x) {
    static_assert(WAVES * 32 <= 512, "filler keys: one per slot");
    auto emit = [&](unsigned long long key, int place) {
        if (a.part_k) {
            a.part_k[(size_t)place * n_wg + wg] = key;
        } else {
            a.part_c[(size_t)wg * K + place] = key_cost(key);
            a.part_i[(size_t)wg * K + place] = key_idx(key);
        }
    };
    if (WAVES == 1) {
        if (wave == 0 && lane < K) emit(run_key, lane);
        return;
    }
    // candidates and comparands: slots 0 .. K-1 of every row (WAVES x K >= K keys, all different: places 0 .. K-1 are all
    // taken); comparands are read in pairs, so an odd K also reads slot K -- a filler behind its row's candidates, whose
    // own place would be >= K
    const int kp = (K + 1) & ~1, half = kp >> 1;
    unsigned long long* keys = &wg_keys[0][0][0];
    if (wave < WAVES && lane < kp) {
        unsigned long long v = lane < K ? run_key : KEY_SENTINEL;
        if (v == KEY_SENTINEL) v = (KEY_SENTINEL & 0xFFFFFFFF00000000ull) | (KEY_FILL_LO + (unsigned)(wave * 32 + lane));
        keys[wave * 32 + lane] = v;
    }
    __syncthreads();
    if (wave < WAVES) {
        const int q = lane >> 4;   // rows q, q + 4, ..
        for (int s0 = 0; s0 < K; s0 += 16) {
            const int i = s0 + (lane & 15);
            const unsigned long long mine = keys[wave * 32 + (i < 32 ? i : 31)];
            unsigned place = 0;
            for (int r = q; r < WAVES; r += 4) {
                const unsigned long long* b = keys + r * 32;
#pragma unroll 4
                for (int c = 0; c < half; ++c) place += (b[2 * c] < mine ? 1u : 0u) + (b[2 * c + 1] < mine ? 1u : 0u);
            }
            place += (unsigned)__shfl_xor((int)place, 16, 64);
            place += (unsigned)__shfl_xor((int)place, 32, 64);
            if (q == 0 && i < K && place < (unsigned)K) {
                const bool filler = (mine >> 32) == (KEY_SENTINEL >> 32) && (unsigned)mine >= KEY_FILL_LO;
                emit(filler ? KEY_SENTINEL : mine, (int)place);
            }
        }
    }
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

// The same minimum for (cost | index) keys in a fraction of the dependent instructions: the 32-bit cost halves first
// (one v_min_u32 with a DPP operand per step), then the index among the lanes that hold that cost -- one lane in all
// but tied costs (ballot), where a second 32-bit reduction over the indices decides, as the 64-bit compare would.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned min_dpp32(unsigned x) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
    return o < x ? o : x;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned x) {
    x = min_dpp32<0xB1, 0xF>(x);
    x = min_dpp32<0x4E, 0xF>(x);
    x = min_dpp32<0x141, 0xF>(x);
    x = min_dpp32<0x140, 0xF>(x);
    x = min_dpp32<0x142, 0xA>(x);
    x = min_dpp32<0x143, 0xC>(x);
    return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ unsigned long long wave_min_key(unsigned long long x) {
    const unsigned hi = (unsigned)(x >> 32), lo = (unsigned)x;
    const unsigned mh = wave_min_u32(hi);
    const unsigned long long tie = __ballot(hi == mh);
    unsigned ml;
    if (__popcll(tie) == 1)   // (wave-uniform)
        ml = (unsigned)__builtin_amdgcn_readlane((int)lo, __ffsll((long long)tie) - 1);
    else
        ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
    return ((unsigned long long)mh << 32) | ml;
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

// index the first kept candidate carries in its key: n_global (kept / shifted elites of the merges), or keep_base where
// the caller indexes differently (a sharded rank's pack: its local pool row n_loc)
__device__ __forceinline__ int keep_index0(const MergeSingleArgs& a) { return a.keep_base >= 0 ? a.keep_base : a.n_global; }

// Global sorted top-K of the candidate lists (+ kept elites): ONE wavefront; sel[0..K) receives the keys.
// Lane t owns lists t, t+64, t+128, t+192 (each sorted) in registers; key r of list w sits at
// part_k[r * n_lists + w], so every load is one contiguous 512 bytes.
// KEPT_APART: the kept elites are offered as candidates of their own behind the lists' survivors instead of being
// inserted into their lanes' first lists (a KREG-step compare-exchange chain, 0.56 us).  Measured per caller: the last
// merge 6.52 -> 6.21 us and the one-tile single-launch kernel 10.07 -> 9.80 us with it, but the two- and four-tile
// single-launch kernels +0.8 / +0.5 us per launch (N = 8192: 83.3 -> 87.7 us per MPC step) -- those keep the insertion.
// COH: every load of another workgroup's data bypasses this CU's L1 (sc1: served by the L2) -- for callers INSIDE a launch whose
// lists were written by other workgroups of the same launch (step_xcd_kernel); LISTS: lists per lane (64 x LISTS >= n_lists).
template <bool COH, class T>
__device__ __forceinline__ T ld_coh(const T* p) {
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
template <int KREG, bool KEPT_APART = false, bool COH = false, int LISTS = LPL>
__device__ __forceinline__ void merge_select(const MergeSingleArgs& a, int lane, unsigned long long* cand,
                                             unsigned long long* sel) {
    unsigned long long k[LISTS][KREG];
    // (the kept elite's cost comes from another buffer: requested first, consumed behind the lists)
    const float keep_cost = a.elites_cost_cur ? ld_coh<COH>(a.elites_cost_cur + (lane < a.n_keep ? lane : 0)) : 0.f;
#pragma unroll
    for (int l = 0; l < LISTS; ++l) {
        const int list = lane + l * 64;
#pragma unroll
        for (int i = 0; i < KREG; ++i)
            k[l][i] = ld_coh<COH>(a.part_k + (size_t)(i < a.K ? i : 0) * a.n_lists + (list < a.n_lists ? list : 0));
    }
#pragma unroll
    for (int l = 0; l < LISTS; ++l) {
        const bool has_list = lane + l * 64 < a.n_lists;
#pragma unroll
        for (int i = 0; i < KREG; ++i) k[l][i] = (has_list && i < a.K) ? k[l][i] : KEY_SENTINEL;
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[1] = wall_clock64();
    // kept elite `lane` (icem.py:143-145): joins this lane's first list, order preserved -- or (KEPT_APART) stays one
    // more candidate of this lane, offered behind the lists' survivors (the threshold below then comes from the lists
    // alone: still an upper bound of the K-th smallest key overall)
    unsigned long long kept = KEY_SENTINEL;
    if constexpr (KEPT_APART) {
        kept = lane < a.n_keep ? make_key(keep_cost, keep_index0(a) + lane) : KEY_SENTINEL;
    } else if (lane < a.n_keep) {
        unsigned long long v = make_key(keep_cost, keep_index0(a) + lane);
#pragma unroll
        for (int i = 0; i < KREG; ++i) {
            const bool sw = v < k[0][i];
            const unsigned long long t = sw ? k[0][i] : v;
            k[0][i] = sw ? v : k[0][i];
            v = t;
        }
    }
    // selection by threshold: the K-th smallest of the 64 lane minima bounds the K-th smallest key overall, so the
    // global top-K is among the keys <= T; those (usually K..2K of them) are compacted into one key per lane and
    // sorted.  The threshold only needs the minima's COST halves (32-bit sort; keys that tie with T in cost all
    // survive), and every list is sorted, so a list's survivors are a prefix: the compaction walks the lists depth
    // by depth and stops at the first depth without a survivor (usually the second).
    if (a.dbg && threadIdx.x == 0) a.dbg[2] = wall_clock64();
    unsigned long long mine = k[0][0];
#pragma unroll
    for (int l = 1; l < LISTS; ++l) mine = k[l][0] < mine ? k[l][0] : mine;
    const unsigned srt = wave_sort64_u32((unsigned)(mine >> 32), lane);
    const unsigned T = __shfl(srt, a.K - 1, 64);
    unsigned n_cand = 0;  // wave-uniform
#pragma unroll
    for (int i = 0; i < KREG; ++i) {
        bool p[LISTS];
        bool any_lane = false;
#pragma unroll
        for (int l = 0; l < LISTS; ++l) {
            p[l] = (unsigned)(k[l][i] >> 32) <= T && k[l][i] != KEY_SENTINEL;
            any_lane |= p[l];
        }
        if (__ballot(any_lane) == 0) break;
#pragma unroll
        for (int l = 0; l < LISTS; ++l) {
            const unsigned long long m = __ballot(p[l]);
            if (m != 0) {
                const unsigned pos = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (p[l] && pos < 64) cand[pos] = k[l][i];
                n_cand += (unsigned)__popcll(m);
            }
        }
    }
    if constexpr (KEPT_APART) {   // the kept elites at or below the threshold
        const bool p = (unsigned)(kept >> 32) <= T && kept != KEY_SENTINEL;
        const unsigned long long m = __ballot(p);
        if (m != 0) {
            const unsigned pos = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (p && pos < 64) cand[pos] = kept;
            n_cand += (unsigned)__popcll(m);
        }
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[3] = wall_clock64();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (n_cand <= 64) {
        unsigned long long key = lane < (int)n_cand ? *((volatile unsigned long long*)&cand[lane]) : KEY_SENTINEL;
        key = wave_sort_n(key, lane, n_cand);
        if (lane < a.K) sel[lane] = key;
    } else {
        // more than 64 keys tie at or below T: K tournament rounds over the list heads (and the kept elites)
        for (int r = 0; r < a.K; ++r) {
            unsigned long long head = k[0][0];
#pragma unroll
            for (int l = 1; l < LISTS; ++l) head = k[l][0] < head ? k[l][0] : head;
            if constexpr (KEPT_APART) head = kept < head ? kept : head;
            const unsigned long long best = wave_min_u64(head);
            if (best != KEY_SENTINEL) {  // keys embed the trajectory index: exactly one (lane, list) matches
                if constexpr (KEPT_APART) {
                    if (kept == best) kept = KEY_SENTINEL;
                }
#pragma unroll
                for (int l = 0; l < LISTS; ++l) {
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
    const unsigned long long kept = lane < a.n_keep ? make_key(a.elites_cost_cur[lane], keep_index0(a) + lane) : KEY_SENTINEL;
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
        key = wave_sort_n(key, lane, n_cand);
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

// The selection for ONE wave of a launch that has 128 registers (the noise-ahead launch's merge prologue): the lists' first
// DEPTH keys in registers -- one load round, like merge_select; the survivors at or below the threshold are a prefix of every
// (sorted) list and rarely reach past the second depth -- and whatever lies deeper fetched only where a list still has a
// survivor at depth DEPTH - 1 (dependent loads, L2-resident by then).  More than 64 survivors: the streaming form's tournament.
// Same result as merge_select (the K smallest keys, ascending; keys are all different).
template <int DEPTH>
__device__ __forceinline__ void merge_select_shallow(const MergeSingleArgs& a, int lane, unsigned long long* cand,
                                                     unsigned long long* sel) {
    const int K = a.K, nl = a.n_lists;
    const float keep_cost = a.elites_cost_cur ? a.elites_cost_cur[lane < a.n_keep ? lane : 0] : 0.f;
    unsigned long long k[LPL][DEPTH];
#pragma unroll
    for (int l = 0; l < LPL; ++l) {
        const int list = lane + l * 64;
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) k[l][i] = a.part_k[(size_t)(i < K ? i : 0) * nl + (list < nl ? list : 0)];
    }
#pragma unroll
    for (int l = 0; l < LPL; ++l) {
        const bool has_list = lane + l * 64 < nl;
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) k[l][i] = (has_list && i < K) ? k[l][i] : KEY_SENTINEL;
    }
    const unsigned long long kept = lane < a.n_keep ? make_key(keep_cost, keep_index0(a) + lane) : KEY_SENTINEL;
    unsigned long long mine = k[0][0];
#pragma unroll
    for (int l = 1; l < LPL; ++l) mine = k[l][0] < mine ? k[l][0] : mine;
    const unsigned srt = wave_sort64_u32((unsigned)(mine >> 32), lane);
    const unsigned T = __shfl(srt, K - 1, 64);   // (cost halves: keys that tie with T in cost all survive)
    unsigned n_cand = 0;  // wave-uniform
    auto offer = [&](unsigned long long key) {
        const bool p = (unsigned)(key >> 32) <= T && key != KEY_SENTINEL;
        const unsigned long long m = __ballot(p);
        if (m != 0) {
            const unsigned pos = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (p && pos < 64) cand[pos] = key;
            n_cand += (unsigned)__popcll(m);
        }
        return m != 0;
    };
    bool deeper[LPL];   // (wave-uniform) list group l still had a survivor at the last depth held in registers
#pragma unroll
    for (int l = 0; l < LPL; ++l) {
        deeper[l] = true;
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            if (deeper[l]) deeper[l] = offer(k[l][i]);
    }
#pragma unroll 1
    for (int l = 0; l < LPL; ++l) {
        if (!deeper[l]) continue;
#pragma unroll 1
        for (int i = DEPTH; i < K; ++i) {
            const int list = lane + l * 64;
            const unsigned long long v = list < nl ? a.part_k[(size_t)i * nl + list] : KEY_SENTINEL;
            if (!offer(v)) break;
        }
    }
    offer(kept);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (n_cand <= 64) {
        unsigned long long key = lane < (int)n_cand ? *((volatile unsigned long long*)&cand[lane]) : KEY_SENTINEL;
        key = wave_sort_n(key, lane, n_cand);
        if (lane < K) sel[lane] = key;
    } else {
        merge_select_stream(a, lane, cand, sel);   // (its own threshold pass again, then the tournament: more than 64 ties)
    }
}

// Sharded runs: the candidates are n_rec <= 128 all-gathered records {cost, gidx, actions} (+ kept elites).  Two
// records and one kept elite per lane; same threshold selection; slot[r] receives the record number of selected key
// r (n_rec + e for kept elite e).  Ties cannot occur: keys embed the global trajectory index.
__device__ __forceinline__ void merge_select_records(const MergeSingleArgs& a, int lane, unsigned long long* cand,
                                                     unsigned long long* sel, int* slot, long long* stamp = nullptr) {
    const int K = a.K, rs = a.h * a.d + 2;
    xchg_wait(a.xw, lane);  // in-library exchange: the peers' records of this iteration have landed
    if (stamp && lane == 0) *stamp = wall_clock64();
    auto rec_key = [&](int e) -> unsigned long long {
        if (e >= a.n_rec) return KEY_SENTINEL;
        const float* rec = a.records + (size_t)e * rs;
        return make_key(rec[0], reinterpret_cast<const int*>(rec + 1)[0]);
    };
    unsigned long long k[3] = {rec_key(lane), rec_key(lane + 64),
                               lane < a.n_keep ? make_key(a.elites_cost_cur[lane], a.n_global + lane) : KEY_SENTINEL};
    // K rounds of "the smallest key of the wave" (wave_min_key).  The selected key is wave-uniform, so the lane that holds
    // it records its slot on the spot.  (The earlier form -- sort the lane minima for a threshold, compact through LDS,
    // sort again, look every selected key up -- took the same 2.5 us beside the sampling waves: EXPERIMENTS.md R3.14.)
    const unsigned long long k0 = k[0], k1 = k[1], k2 = k[2];
    for (int r = 0; r < K; ++r) {
        unsigned long long head = k[0] < k[1] ? k[0] : k[1];
        head = k[2] < head ? k[2] : head;
        const unsigned long long best = wave_min_key(head);
        if (best != KEY_SENTINEL) {   // (keys embed the global trajectory index: exactly one lane and slot matches)
            if (k0 == best) { slot[r] = lane; k[0] = KEY_SENTINEL; }
            if (k1 == best) { slot[r] = lane + 64; k[1] = KEY_SENTINEL; }
            if (k2 == best) { slot[r] = a.n_rec + lane; k[2] = KEY_SENTINEL; }
        } else if (lane == 0) {
            slot[r] = 0;  // fewer than K live candidates: reference behaviour undefined, stay in bounds
        }
        if (lane == 0) sel[r] = best;
    }
    (void)cand;
}

// The same selection by the WHOLE workgroup (round 6): the K rounds of a wave minimum above are a chain of K dependent
// reductions -- 3.7 us for K = 10 on the critical path of every sharded launch (stamps: EXPERIMENTS R6.5) -- while the
// workgroup's other waves wait at the barrier behind it.  Here ONE wave (`stager`) waits for the exchange and parks the keys
// (<= 128 records + <= 64 kept elites) in LDS; then every thread counts for one (candidate, part): a key's place is the
// number of smaller keys (keys embed the global row: all different), places < K go straight to sel / slot.  Two barriers
// and a dozen compares per thread.  Called by ALL `nthr` threads of the workgroup at one point; ends behind a barrier.
__device__ __forceinline__ void merge_select_records_wg(const MergeSingleArgs& a, bool stager, int lane, int tid, int nthr,
                                                        unsigned long long* sel, int* slot, long long* stamp = nullptr) {
    __shared__ unsigned long long rk_keys[192];
    __shared__ unsigned rk_rank[192];
    const int K = a.K, rs = a.h * a.d + 2;
    if (stager) {
        xchg_wait(a.xw, lane);  // in-library exchange: the peers' records of this iteration have landed
        if (stamp && lane == 0) *stamp = wall_clock64();
        auto rec_key = [&](int e) -> unsigned long long {
            if (e >= a.n_rec) return KEY_SENTINEL;
            const float* rec = a.records + (size_t)e * rs;
            return make_key(rec[0], reinterpret_cast<const int*>(rec + 1)[0]);
        };
        rk_keys[lane] = rec_key(lane);
        rk_keys[64 + lane] = rec_key(lane + 64);
        rk_keys[128 + lane] = lane < a.n_keep ? make_key(a.elites_cost_cur[lane], a.n_global + lane) : KEY_SENTINEL;
        rk_rank[lane] = rk_rank[64 + lane] = rk_rank[128 + lane] = 0u;
        sel[lane] = KEY_SENTINEL;   // fewer than K live candidates: reference behaviour undefined, stay in bounds
        slot[lane] = 0;
    }
    __syncthreads();
    // live candidate c: record c (c < n_rec) at rk_keys[c], kept elite c - n_rec at rk_keys[128 + c - n_rec]
    const int n_rec = a.n_rec, n_live = n_rec + a.n_keep;
    auto at = [&](int c) { return c < n_rec ? c : 128 + (c - n_rec); };
    int parts = n_live > 0 ? nthr / n_live : 1;
    parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
    if (n_live > 0 && tid < n_live * parts) {
        const int c = tid % n_live, p = tid / n_live;
        const unsigned long long mine = rk_keys[at(c)];
        unsigned cnt = 0;
        for (int j = p; j < n_live; j += parts) cnt += rk_keys[at(j)] < mine ? 1u : 0u;
        if (cnt) atomicAdd(&rk_rank[at(c)], cnt);
    }
    if (n_live > nthr)   // (never with the shipped workgroup sizes: n_live <= 192; kept for completeness)
        for (int c = nthr + tid; c < n_live; c += nthr) {
            const unsigned long long mine = rk_keys[at(c)];
            unsigned cnt = 0;
            for (int j = 0; j < n_live; ++j) cnt += rk_keys[at(j)] < mine ? 1u : 0u;
            rk_rank[at(c)] = cnt;
        }
    __syncthreads();
    for (int c = tid; c < n_live; c += nthr) {
        const unsigned long long mine = rk_keys[at(c)];
        const unsigned r = rk_rank[at(c)];
        if (mine != KEY_SENTINEL && r < (unsigned)K) {
            sel[r] = mine;
            slot[r] = c;   // (record number; n_rec + e for kept elite e)
        }
    }
    __syncthreads();
}

// The same selection spread over the NW wavefronts of a workgroup, for a merge prologue that has NOTHING to hide behind
// (rollout16_ahead_kernel: every wave waits for the distribution).  merge_select_stream walks the lists with dependent
// loads -- a dozen L2 round trips, 12-14 us when exposed (measured) -- and merge_select holds all 4 x K keys of a lane in
// registers (174 VGPRs); here every wave takes n_lists / NW lists, ONE list per lane, all K keys of it requested at
// once (24 registers), reduces them to its own K best (threshold on the lane minima, depth-wise compaction, one sort) and
// parks those in LDS; behind a workgroup barrier wave 0 selects the K best of the NW x K (+ kept elites).  One cold
// round trip + four sorts.  Same result: the K smallest keys overall, ascending (keys are unique: they embed the row).
// Stage 1: call from EVERY wave (w = wave index); then __syncthreads(); stage 2: call from wave 0.
template <int KREG>
__device__ __forceinline__ void merge_select_split_stage1(const MergeSingleArgs& a, int lane, int w, int nw, unsigned long long* cand /* [64] of this wave */,
                                                          unsigned long long* wsel /* [nw][16] */) {
    const int per = (a.n_lists + nw - 1) / nw;  // <= 64 (n_lists <= 256, nw >= 4)
    const int base = w * per;
    const int cnt = base < a.n_lists ? (a.n_lists - base < per ? a.n_lists - base : per) : 0;
    const bool has = lane < cnt;
    const int list = has ? base + lane : 0;
    unsigned long long k[KREG];
#pragma unroll
    for (int i = 0; i < KREG; ++i) k[i] = a.part_k[(size_t)(i < a.K ? i : 0) * a.n_lists + list];
#pragma unroll
    for (int i = 0; i < KREG; ++i) k[i] = (has && i < a.K) ? k[i] : KEY_SENTINEL;
    const unsigned srt = wave_sort64_u32((unsigned)(k[0] >> 32), lane);
    const unsigned T = __shfl(srt, a.K - 1, 64);  // (fewer lists than K: the sentinel's cost half -- everything survives)
    unsigned n_cand = 0;
#pragma unroll
    for (int i = 0; i < KREG; ++i) {
        const bool p = (unsigned)(k[i] >> 32) <= T && k[i] != KEY_SENTINEL;
        const unsigned long long m = __ballot(p);
        if (m == 0) break;
        const unsigned pos = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (p && pos < 64) cand[pos] = k[i];
        n_cand += (unsigned)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    unsigned long long* out = wsel + w * 16;
    if (n_cand <= 64) {
        unsigned long long key = lane < (int)n_cand ? *((volatile unsigned long long*)&cand[lane]) : KEY_SENTINEL;
        key = wave_sort_n(key, lane, n_cand);
        if (lane < 16) out[lane] = lane < a.K ? key : KEY_SENTINEL;
    } else {
        // more than 64 keys tie at or below the threshold: K tournament rounds over the list heads
        for (int r = 0; r < a.K; ++r) {
            const unsigned long long best = wave_min_u64(k[0]);
            if (best != KEY_SENTINEL && k[0] == best) {
#pragma unroll
                for (int i = 0; i + 1 < KREG; ++i) k[i] = k[i + 1];
                k[KREG - 1] = KEY_SENTINEL;
            }
            if (lane == 0) out[r] = best;
        }
        if (lane >= a.K && lane < 16) out[lane] = KEY_SENTINEL;
    }
}
// the kept elite's cost of lane `lane` (icem.py:143-145) for stage 2: a cold load of its own -- callers request it in front
// of stage 1 so that it travels with the lists' keys instead of behind the workgroup barrier
__device__ __forceinline__ float merge_keep_cost(const MergeSingleArgs& a, int lane) {
    return a.elites_cost_cur ? a.elites_cost_cur[lane < a.n_keep ? lane : 0] : 0.f;
}
__device__ __forceinline__ void merge_select_split_stage2(const MergeSingleArgs& a, int lane, int nw, const unsigned long long* wsel,
                                                          unsigned long long* cand, unsigned long long* sel, float keep_cost) {
    // nw * 16 <= 256 slots (sentinels behind each wave's K keys): four per lane, + kept elite `lane` (icem.py:143-145)
    unsigned long long k[5];
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = (lane + 64 * j < nw * 16) ? *((volatile const unsigned long long*)&wsel[lane + 64 * j]) : KEY_SENTINEL;
    k[4] = (lane < a.n_keep && a.elites_cost_cur) ? make_key(keep_cost, keep_index0(a) + lane) : KEY_SENTINEL;
    unsigned long long mine = k[0];
#pragma unroll
    for (int j = 1; j < 5; ++j) mine = k[j] < mine ? k[j] : mine;
    const unsigned long long srt = wave_sort64(mine, lane);
    const unsigned long long T = __shfl(srt, a.K - 1, 64);
    unsigned n_cand = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const bool p = k[j] <= T && k[j] != KEY_SENTINEL;
        const unsigned long long m = __ballot(p);
        const unsigned pos = n_cand + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (p && pos < 64) cand[pos] = k[j];
        n_cand += (unsigned)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if (n_cand <= 64) {
        unsigned long long key = lane < (int)n_cand ? *((volatile unsigned long long*)&cand[lane]) : KEY_SENTINEL;
        key = wave_sort_n(key, lane, n_cand);
        if (lane < a.K) sel[lane] = key;
    } else {
        for (int r = 0; r < a.K; ++r) {
            unsigned long long head = k[0];
#pragma unroll
            for (int j = 1; j < 5; ++j) head = k[j] < head ? k[j] : head;
            const unsigned long long best = wave_min_u64(head);
            if (best != KEY_SENTINEL) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (k[j] == best) k[j] = KEY_SENTINEL;
            }
            if (lane == 0) sel[r] = best;
        }
    }
}

// Stage 2 without a sort, spread over ALL waves of the workgroup.  Behind stage 1 the candidates are the NW x 16 slots of
// wsel: every wave's K best and, in the slots behind them, the kept elites (icem.py:143-145; wave w parks elites w,
// w + NW, ..: merge_select_split_keep, in front of the workgroup barrier) -- at most NW x 16 keys, all different (a key
// embeds its row).  A key's place in the ascending order is the number of smaller keys: wave w counts that for the 16
// slots of its row (lane = slot + 16 x the quarter of the comparands it reads, two per ds_read_b128) and writes the key
// to sel[place] if place < K.  32 independent compare-and-adds per lane where stage 2 is one wave's ~600-instruction
// dependent chain (two bitonic networks on 64-bit keys).  Same result: the K smallest keys, ascending.
// Needs n_keep <= (16 - K) x NW (merge_select_split_by_rank; callers fall back to stage 2 otherwise).
template <int NW>
__device__ __forceinline__ bool merge_select_split_by_rank(const MergeSingleArgs& a) { return a.n_keep <= (16 - a.K) * NW; }
// the cost of the kept elite lane `lane` of wave w parks (requested in front of stage 1: it travels with the lists' keys)
template <int NW>
__device__ __forceinline__ float merge_keep_cost_split(const MergeSingleArgs& a, int lane, int w) {
    const int j = w + NW * (lane - a.K);
    return a.elites_cost_cur ? a.elites_cost_cur[(lane >= a.K && lane < 16 && j < a.n_keep) ? j : 0] : 0.f;
}
template <int NW>
__device__ __forceinline__ void merge_select_split_keep(const MergeSingleArgs& a, int lane, int w, unsigned long long* wsel,
                                                        unsigned long long* sel, float keep_cost) {
    if (lane >= a.K && lane < 16) {   // (behind stage 1's own writes of these slots: same wave, program order)
        const int j = w + NW * (lane - a.K);
        wsel[w * 16 + lane] = (j < a.n_keep && a.elites_cost_cur) ? make_key(keep_cost, keep_index0(a) + j) : KEY_SENTINEL;
    }
    if (w == 0 && lane < 16) sel[lane] = KEY_SENTINEL;   // (fewer than K candidates: the places behind them)
}
template <int NW>
__device__ __forceinline__ void merge_select_split_rank(const MergeSingleArgs& a, int lane, int w, const unsigned long long* wsel,
                                                        unsigned long long* sel) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const int i = lane & 15, q = lane >> 4;
    const unsigned long long mine = wsel[w * 16 + i];
    const u64x2* pairs = reinterpret_cast<const u64x2*>(wsel) + q;
    unsigned place = 0;
#pragma unroll
    for (int c = 0; c < 2 * NW; ++c) {
        const u64x2 b = pairs[4 * c];
        place += (b[0] < mine ? 1u : 0u) + (b[1] < mine ? 1u : 0u);
    }
    place += (unsigned)__shfl_xor((int)place, 16, 64);
    place += (unsigned)__shfl_xor((int)place, 32, 64);
    if (q == 0 && mine != KEY_SENTINEL && place < (unsigned)a.K) sel[place] = mine;
}

// pointers to the K selected rows (icem.py:201): pool rows, or kept elites behind index n_global
template <int KREG, bool REC>
__device__ __forceinline__ void merge_rows(const MergeSingleArgs& a, const unsigned long long* sel, const int* slot,
                                           const float* (&rows)[KREG]) {
    const int hd = a.h * a.d;
    // ONE LDS read per lane (lane r: selected key / slot r), then a v_readlane per row: read one by one through the same
    // uniform address, the K entries were K dependent LDS round trips in front of the gather's loads (0.3 us of the merge
    // prologue's chain: EXPERIMENTS R6.20); the pointers themselves are scalar arithmetic either way.
    // ... and lane r also works out row r's address (one set of vector instructions for all K rows instead of K sets of a
    // dozen scalar ones in front of the first load); the 64-bit addresses come back through v_readlane.
    const int lane = (int)(threadIdx.x & 63);
    const int v = REC ? slot[lane < a.K ? lane : 0] : key_idx(sel[lane < a.K ? lane : 0]);
    const float* mine;
    if (REC) mine = v < a.n_rec ? a.records + (size_t)v * (hd + 2) + 2 : a.elites_cur + (size_t)(v - a.n_rec) * hd;   // sharded run: candidates are records
    else mine = v < a.n_pool ? a.actions + (size_t)v * hd : a.elites_cur + (size_t)(v - a.n_global) * hd;
    const unsigned long long bits = (unsigned long long)(size_t)mine;
#pragma unroll
    for (int r = 0; r < KREG; ++r) {
        const int rr = r < a.K ? r : 0;
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, rr);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), rr);
        rows[r] = reinterpret_cast<const float*>((size_t)(((unsigned long long)hi << 32) | lo));
    }
}

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

// sample_into_tile with the row on a quad of lanes (q = lane & 3), table from LDS (row_synth_quad).  `publish()` runs
// between the draws and the synthesis, in EVERY lane of the wave (live = false: a lane without a row): the caller
// uses it to finish its wave's private copy of the table, whose global loads it issued in front of the draws.
template <int H, int D, int ROUNDS, bool RAW, typename Publish>
__device__ __forceinline__ void sample_into_tile_quad(const FastSampleArgs& sa, bool live, int n_rows, int r_mine, int jd, int q,
                                                      float* trow, const float* mrow, const float* Wl, Publish&& publish) {
    constexpr int HD = H * D;
    float g[HMAX];
    const int kind = !live ? 3 : (r_mine < sa.n ? 0 : ((r_mine < n_rows && !RAW) ? 1 : 2));
    if (kind == 0)
        row_normals_quad<H, ROUNDS>((unsigned)(sa.first_index + r_mine), (unsigned)jd, sa.off_lo, sa.off_hi, sa.seed_lo, sa.seed_hi, q, g);
    else if (kind == 1)  // shifted elite e = r_mine - sa.n: stream off2 (icem.py:91-104)
        row_normals_quad<H, ROUNDS>((unsigned)(r_mine - sa.n), (unsigned)jd, sa.off2_lo, sa.off2_hi, sa.seed_lo, sa.seed_hi, q, g);
    publish();
    if (kind == 0) {
        if (RAW) {
            row_synth_quad<H>(Wl, g, q, [&](int t, float y) { trow[t * D] = y; }, sa.white != 0);
        } else {
            const float lo = sa.low[jd], hi = sa.high[jd];
            row_synth_quad<H>(Wl, g, q, [&](int t, float y) {
                const float v = __builtin_fmaf(y, mrow[HD + t * D], mrow[t * D]);
                trow[t * D] = __builtin_amdgcn_fmed3f(v, lo, hi);
            }, sa.white != 0);
        }
    } else if (kind == 1) {
        // elites[e, 1:, j] and a last action from the full (n_shift, d, h) noise batch (only t = h-1 is used)
        const int e = r_mine - sa.n;
        const float lo = sa.low[jd], hi = sa.high[jd];
        row_synth_quad<H>(Wl, g, q, [&](int t, float y) {
            if (t == H - 1) {
                const float v = __builtin_fmaf(y, mrow[HD + t * D], mrow[t * D]);
                trow[(H - 1) * D] = __builtin_amdgcn_fmed3f(v, lo, hi);
            }
        }, sa.white != 0);
        const float* src = sa.elites_src + (size_t)e * HD + jd;
        for (int t = q; t < H - 1; t += 4) trow[t * D] = src[(t + 1) * D];
    } else if (kind == 2) {
        for (int t = q; t < H; t += 4) trow[t * D] = 0.f;  // past the end: rolled out, dropped
    }
}

// A trajectory whose cost came out NaN (a state left the arithmetic's range, or the inputs were not finite): ranked last by
// its key as always -- and counted in the handle's status word (icem_nonfinite_costs; icem_get_action reports it)
__device__ __forceinline__ void note_nonfinite(const FastRolloutArgs& a, float cost, bool stored) {
    if (stored && cost != cost) atomicAdd(a.nonfinite, 1u);
}

// One wave rolls its 16 trajectories of the slab out of the LDS tile, stores the costs and folds them into its
// running candidate list.
template <typename Tile, int H, int D, bool PLANES_FIRST = false>
__device__ __forceinline__ unsigned long long rollout_slab(Tile& tile, const FastRolloutArgs& ra, const float* rd0, int row,
                                                           int n_rows, unsigned long long run_key, bool first, int lane) {
    const bool live = row < n_rows;
    typename Tile::State st;
    tile.init(st);
    if constexpr (PLANES_FIRST && Tile::LONE_STEP) {
        float raw[2] = {rd0[tile.xoff[0]], rd0[tile.xoff[1]]};
#pragma unroll
        for (int t = 0; t < H; ++t) tile.step_lone(st, raw, rd0 + (t + 1 < H ? t + 1 : t) * D);
    } else {
#pragma unroll
        for (int t = 0; t < H; ++t) tile.template step<PLANES_FIRST>(st, rd0 + t * D);
    }
    const float cost = tile.cost(st);
    if (live && lane < 16) ra.costs[row] = cost;
    note_nonfinite(ra, cost, live && lane < 16);
    if (ra.K > 0) {
        const unsigned long long key = (lane < 16 && live && row < ra.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
        run_key = topk_push16(run_key, key, first, ra.K, lane);
    }
    return run_key;
}

// One wave rolls ONE 16-trajectory tile out of the action tensor in HBM / L2: the tile's actions are one contiguous
// 16 x H x D block; the wave fetches it with full-width coalesced loads, chunk by chunk (one chunk prefetched in
// registers), into its own LDS staging buffer; each lane then reads the one or two entries it feeds to the MFMAs.
// Only this wave touches the buffer and a wave's LDS operations execute in order: no barriers.  Shared by
// rollout16_kernel (k_rollout.hip) and rollout16_ahead_kernel (k_rollout_ahead.hip).
template <typename TileT, int H, int D>
struct StreamT {
    using Tile = TileT;
    static constexpr int HD = H * D;
    static constexpr int VW = HD % 4 == 0 ? 4 : 2;    // floats per load: rows are 16-byte aligned only if h*d % 4 == 0
    static_assert(HD % 2 == 0, "8-byte aligned action rows");
    using Vec = typename VecOf<VW>::type;
    static constexpr int TC = r16_chunk_steps(H, D, VW);  // steps per action chunk
    static_assert(TC > 0, "no aligned action chunk for this (H, D)");
    static constexpr int CB = TC * D;                 // floats per row and chunk
    static constexpr int C4 = CB / VW;                // vectors per row and chunk
    static constexpr int CBP = (C4 % 2) ? CB : CB + VW;  // LDS row stride: odd number of vectors
    static constexpr int NCH = H / TC;
    static constexpr int F4 = 16 * C4;                // vectors per chunk of a 16-trajectory tile
    static constexpr int NLD = (F4 + 63) / 64;        // cooperative load instructions per chunk
    static constexpr int STG = Tile::SLACK + 16 * CBP + Tile::TAIL;  // floats of staging per wave

    // cooperative loads: vector number f = m * 64 + lane of a chunk is row f / C4, vector f % C4 of that row
    int ld_row[NLD], ld_c4[NLD];
    bool ld_on[NLD];
    const float* rd0;
    float* stage;

    __device__ __forceinline__ void init(const Tile& tile, float* stage_buf, int lane) {
        stage = stage_buf;
        rd0 = tile.read_ptr(stage_buf, lane, CBP);
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int f = m * 64 + lane;
            ld_on[m] = f < F4;
            ld_row[m] = ld_on[m] ? f / C4 : 0;
            ld_c4[m] = ld_on[m] ? f % C4 : 0;
        }
    }

    // costs[row] of the tile's live rows; the tile's keys join the wave's running sorted top-K.  `pre`: the tile's first
    // chunk, requested by the caller (first_loads) -- for a wave's first tile in front of the workgroup barrier that waits
    // for the model operands, so that the two cold round trips overlap instead of following each other.
    __device__ __forceinline__ unsigned long long run(const Tile& tile, const FastRolloutArgs& a, int tile_id, int lane,
                                                      unsigned long long run_key, bool first, Vec (&pre)[NLD]) const {
        const int row = tile_id * 16 + (lane & 15);
        const bool live = row < a.n_rows;
        const Vec* src[NLD];
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int r = tile_id * 16 + ld_row[m];
            src[m] = reinterpret_cast<const Vec*>(a.actions + (size_t)(r < a.n_rows ? r : 0) * HD) + ld_c4[m];
        }
        typename Tile::State st;
        tile.init(st);
#pragma unroll
        for (int t = 0; t < H; ++t) {
            if (t % TC == 0) {
                // next chunk: registers -> this wave's LDS buffer, then start fetching the one after
#pragma unroll
                for (int m = 0; m < NLD; ++m)
                    if (ld_on[m]) *reinterpret_cast<Vec*>(&stage[Tile::SLACK + ld_row[m] * CBP + VW * ld_c4[m]]) = pre[m];
                if (t / TC + 1 < NCH) {
#pragma unroll
                    for (int m = 0; m < NLD; ++m) pre[m] = src[m][(t / TC + 1) * C4];
                }
            }
            tile.step(st, rd0 + (t % TC) * D);
        }
        const float cost = tile.cost(st);
        if (live && lane < 16) a.costs[row] = cost;
        note_nonfinite(a, cost, live && lane < 16);
        if (a.K > 0) {
            const unsigned long long key = (lane < 16 && live && row < a.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
            run_key = topk_push16(run_key, key, first, a.K, lane);
        }
        return run_key;
    }

    // ---- noise-ahead form (k_rollout_ahead.hip) --------------------------------------------------------------------
    // The tile's rows hold RAW colored noise y (written ahead of time by noise_rows_kernel: it needs no distribution,
    // icem.py:73-75).  Every vector becomes an action on its way from the prefetch registers to the staging buffer --
    // clip(y * std + mean, low, high) (icem.py:79), the same fmaf + v_med3 the samplers apply, so the same bits; row 0 <-
    // mean on the last iteration (icem.py:87-88) -- and is written back in place: when the launch is over the pool holds
    // the actions and everything downstream (elite gather, record pack, the caller) reads it as before.
    //   dist    LDS [mean (HD) | std (HD)]
    //   lo, hi  the action bounds, the same for every action dimension (wave-uniform scalars; environments with
    //           per-dimension bounds stay on the sampler + rollout pair -- plan.hip checks)
    // Rows >= n_xf are taken as they are (rows that already hold actions).  One load group at a time (the empty asm
    // statements keep the compiler from batching all groups' LDS reads: 16 registers this kernel does not have).
    __device__ __forceinline__ void first_loads(const float* pool, int n_rows, int tile_id, Vec (&pre)[NLD]) const {
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int r = tile_id * 16 + ld_row[m];
            pre[m] = (reinterpret_cast<const Vec*>(pool + (size_t)(r < n_rows ? r : 0) * HD) + ld_c4[m])[0];
        }
    }
    __device__ __forceinline__ unsigned long long run_xf(const Tile& tile, const FastRolloutArgs& a, float* pool, int n_xf,
                                                         bool row0_mean, bool store_back, const float* dist, float lo, float hi, int tile_id,
                                                         int lane, unsigned long long run_key, bool first, Vec (&pre)[NLD]) const {
        const int row = tile_id * 16 + (lane & 15);
        const bool live = row < a.n_rows;
        unsigned voff[NLD];  // float offset of this lane's vector of load group m inside the pool (chunk 0)
        bool xf_on[NLD], is0[NLD], st_on[NLD];
#pragma unroll
        for (int m = 0; m < NLD; ++m) {
            const int r = tile_id * 16 + ld_row[m];
            voff[m] = (unsigned)(r < a.n_rows ? r : 0) * (unsigned)HD + (unsigned)(VW * ld_c4[m]);
            xf_on[m] = ld_on[m] && r < a.n_rows && r < n_xf;
            is0[m] = xf_on[m] && row0_mean && r == 0;
            st_on[m] = xf_on[m] && store_back;
        }
        typename Tile::State st;
        tile.init(st);
        float xraw[2] = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < H; ++t) {
            if (t % TC == 0) {
                const int ch = t / TC;
#pragma unroll
                for (int m = 0; m < NLD; ++m) {
                    asm volatile("" ::: "memory");
                    const int eo = ch * CB + VW * ld_c4[m];
                    const Vec mu = *reinterpret_cast<const Vec*>(dist + eo);
                    const Vec sg = *reinterpret_cast<const Vec*>(dist + HD + eo);
                    float y[VW], vm[VW], vs[VW];
                    __builtin_memcpy(y, &pre[m], sizeof(Vec));
                    __builtin_memcpy(vm, &mu, sizeof(Vec));
                    __builtin_memcpy(vs, &sg, sizeof(Vec));
#pragma unroll
                    for (int k = 0; k < VW; ++k) {
                        const float v = __builtin_amdgcn_fmed3f(__builtin_fmaf(y[k], vs[k], vm[k]), lo, hi);
                        y[k] = xf_on[m] ? (is0[m] ? vm[k] : v) : y[k];
                    }
                    Vec out;
                    __builtin_memcpy(&out, y, sizeof(Vec));
                    if (ld_on[m]) *reinterpret_cast<Vec*>(&stage[Tile::SLACK + ld_row[m] * CBP + VW * ld_c4[m]]) = out;
                    // (the next chunk's request goes out in front of this chunk's store: on this ISA one counter tracks
                    //  loads and stores in order, and a wait for a load issued behind a store waits for the store as well)
                    if (ch + 1 < NCH) pre[m] = *reinterpret_cast<const Vec*>(pool + voff[m] + (ch + 1) * CB);
                    if (st_on[m]) *reinterpret_cast<Vec*>(pool + voff[m] + ch * CB) = out;
                }
                asm volatile("" ::: "memory");
                if constexpr (Tile::LONE_STEP) { xraw[0] = rd0[tile.xoff[0]]; xraw[1] = rd0[tile.xoff[1]]; }
            }
            if constexpr (Tile::LONE_STEP) tile.step_lone(st, xraw, rd0 + ((t + 1) % TC ? (t + 1) % TC : t % TC) * D);
            else tile.step(st, rd0 + (t % TC) * D);
        }
        const float cost = tile.cost(st);
        if (live && lane < 16) a.costs[row] = cost;
        note_nonfinite(a, cost, live && lane < 16);
        if (a.K > 0) {
            const unsigned long long key = (lane < 16 && live && row < a.n_cand) ? make_key(cost, row) : KEY_SENTINEL;
            run_key = topk_push16(run_key, key, first, a.K, lane);
        }
        return run_key;
    }
};
template <int H, int D, int O, int KIND, int ARITH = 0>
using Stream16 = StreamT<typename TileSel<H, D, O, KIND, ARITH>::type, H, D>;

// Sharded runs: this rank's K best candidates (merge_select's selection, already in `sel`) packed as records
// {cost, gidx, actions[h*d]} for the exchange.  Local pool row li is global trajectory shard_lo + li, or
// n_global + (li - n_loc) for the replicated shifted elites behind the shard (icem_amd/distributed.py).  With
// px.peers the records (staged in `stage`, [K, h*d + 2] floats of LDS) also go into this rank's slot of every rank's
// exchange block as 16-byte peer-to-peer stores (xGMI between GPUs), the flags follow behind a system-scope fence.
// Called by all `nthr` threads of ONE workgroup (pack_records_kernel, or workgroup 0 of a sample_rollout launch).
template <int KREG>
__device__ __forceinline__ void pack_records_body(const MergeSingleArgs& a, int n_loc, int shard_lo, float* records, const XchgPush& px,
                                                  float* stage, const unsigned long long* sel, int tid, int nthr) {
    const int hd = a.h * a.d;
    const int rs = hd + 2;
    const bool push = px.peers != nullptr;
    auto put = [&](int off, float v) {
        records[off] = v;
        if (push) stage[off] = v;
    };
    // headers by the first K threads; rows: element e of all K rows per thread, every load in flight before a store
    if (tid < a.K) {
        const unsigned long long key = sel[tid];
        float c = INFINITY;
        int g = INT_MAX;
        if (key != KEY_SENTINEL) {  // (else: fewer than K candidates on this rank)
            const int li = key_idx(key);
            c = key_cost(key);
            g = li < n_loc ? shard_lo + li : a.n_global + (li - n_loc);
        }
        put(tid * rs, c);
        put(tid * rs + 1, __int_as_float(g));
    }
    const float* rows[KREG];
    bool dead[KREG];
#pragma unroll
    for (int r = 0; r < KREG; ++r) {
        const unsigned long long key = sel[r < a.K ? r : 0];
        dead[r] = key == KEY_SENTINEL;
        rows[r] = a.actions + (size_t)(dead[r] ? 0 : key_idx(key)) * hd;
    }
    for (int e = tid; e < hd; e += nthr) {
        float xs[KREG];
#pragma unroll
        for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
#pragma unroll
        for (int r = 0; r < KREG; ++r)
            if (r < a.K) put(r * rs + 2 + e, dead[r] ? 0.f : xs[r]);
    }
    if (push) {
        __syncthreads();
        const int words = a.K * rs;  // K * rs * 4 bytes: a multiple of 8; the tail goes out as dwords
        const int vecs = words / 4;
        for (int p = 0; p < px.world; ++p) {
            float* dst = reinterpret_cast<float*>(px.peers[p] + px.rec_byte_off);
            for (int v = tid; v < vecs; v += nthr) reinterpret_cast<float4*>(dst)[v] = reinterpret_cast<const float4*>(stage)[v];
            for (int e = 4 * vecs + tid; e < words; e += nthr) dst[e] = stage[e];
        }
        // Every thread waits for ITS record stores to be acknowledged, the workgroup meets, then the ONE wave that
        // raises the flags releases at system scope (L2 write-back + the flag stores).  (A system-scope fence in every
        // wave of the workgroup, as it used to be, is 7..13 write-backs of the whole L2 where one is needed: the
        // barrier alone does not wait for outstanding global stores on this target, hence the explicit wait.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < px.world)
            __hip_atomic_store(reinterpret_cast<unsigned*>(px.peers[tid]) + px.flag_idx, px.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- argument blocks read from DEVICE memory (the batched launches of icem_plan_step_batch) -----------------------------
// A pointer that arrives through the kernel-argument segment is known to be a global address; one LOADED from memory is
// generic, and every access through it becomes a flat_load / flat_store (both counters, no overlap with LDS traffic).  gptr
// reads the 8 bytes AS a global-address-space pointer, so the accesses stay global_load / global_store: the same device code
// as the kernels that take their block by value.
template <class T>
__device__ __forceinline__ T* gptr(T* const& field) {
    typedef T __attribute__((address_space(1))) * GP;
    return (T*)(*static_cast<const GP*>(static_cast<const void*>(&field)));   // (the 8 bytes re-read with a global pointer's type)
}
__device__ __forceinline__ FastSampleArgs from_device(const FastSampleArgs& m) {
    FastSampleArgs s = m;
    s.W = gptr(m.W), s.mean = gptr(m.mean), s.std = gptr(m.std), s.low = gptr(m.low), s.high = gptr(m.high), s.out = gptr(m.out);
    s.elites_src = gptr(m.elites_src), s.raw_src = gptr(m.raw_src);
    return s;
}
__device__ __forceinline__ FastRolloutArgs from_device(const FastRolloutArgs& m) {
    FastRolloutArgs r = m;
    r.Mp = gptr(m.Mp), r.perm = gptr(m.perm), r.obs0 = gptr(m.obs0), r.actions = gptr(m.actions), r.costs = gptr(m.costs);
    r.part_c = gptr(m.part_c), r.part_i = gptr(m.part_i), r.part_k = gptr(m.part_k), r.dbg = nullptr, r.nonfinite = gptr(m.nonfinite);
    return r;
}
__device__ __forceinline__ MergeSingleArgs from_device(const MergeSingleArgs& m) {
    MergeSingleArgs a = m;
    a.part_k = gptr(m.part_k), a.records = gptr(m.records), a.actions = gptr(m.actions);
    a.xw.flags = nullptr, a.xw.status = nullptr, a.xw.records = nullptr;   // (world 1: nothing to wait for)
    a.elites_cur = gptr(m.elites_cur), a.elites_cost_cur = gptr(m.elites_cost_cur);
    a.elites_next = gptr(m.elites_next), a.elites_cost_next = gptr(m.elites_cost_next);
    a.mean = gptr(m.mean), a.std = gptr(m.std), a.mean_out = gptr(m.mean_out), a.std_out = gptr(m.std_out);
    a.low = gptr(m.low), a.high = gptr(m.high), a.executed = gptr(m.executed), a.best_cost = gptr(m.best_cost), a.dbg = nullptr;
    return a;
}
__device__ __forceinline__ void add_base64(uint32_t& lo, uint32_t& hi, unsigned long long base) {
    const unsigned long long v = (((unsigned long long)hi << 32) | lo) + base;
    lo = (uint32_t)v;
    hi = (uint32_t)(v >> 32);
}

// rollout launch shape: one 16-trajectory tile per wave while they fit, at most FAST_MAX_LISTS workgroups (= lists)
inline void r16_shape(int n_rows, int* grid, int* waves) {
    const int tiles = std::max(1, (n_rows + 15) / 16);
    const int g = std::min(tiles, FAST_MAX_LISTS);
    int w = 1;
    while (w < 16 && w * g < tiles) w *= 2;
    *grid = g;
    *waves = w;
}

}  // namespace

}  // namespace icem
