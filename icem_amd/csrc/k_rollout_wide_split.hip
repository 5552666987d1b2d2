// k_rollout_wide_split.hip -- K2 + K3 for WIDE observations (32 < o <= 384; HumanoidStandup's o = 378, d = 17:
// icem/environments/mujoco.py:241-277) with the model step on the bf16 matrix cores: rollout_wide_split_kernel.
//
// The step [n, o + d] x [o + d, o] is a GEMM three orders of magnitude above the f32 ridge, and the exact-f32 MFMA of
// k_rollout_wide.hip runs at 1/16 of the 16-bit rate.  Two splits of the f32 operands onto the 16-bit matrix cores, one
// kernel template (F16):
//   * fp16, two planes, THREE products (the default; icem_set_wide_exact 0).  x S = hi + lo with hi = f16(x S), lo =
//     f16(x S - hi), both round-to-nearest-even: 11 + 11 significant bits and lo's sign, x S to 2^-24 relative.  S is a
//     power of two per TRAJECTORY ROW and step, taken from the row's largest entry so that |x| S < 2^15 (fp16 ends at
//     65 504; found by four threads per row at the step's top), and one per model (pack_wide_model_split); both are taken
//     out of the accumulators again at the write-back, exactly.  A product x m = hi Hi + hi Lo + lo Hi (+ lo Lo, dropped:
//     below 2^-22 of hi Hi, i.e. 2^-24-class like the operands' own rounding), each of the three EXACT in the f32
//     accumulator's format (22 significant bits), summed smallest first on v_mfma_f32_16x16x32_f16.  The model is carried
//     EQUILIBRATED (pack_wide_model_split): row k as M[k][:] 2^-e_k with entry k as x_k 2^e_k, column j as M[:][j] 2^-f_j with
//     the accumulators x 2^f_j at the write-back -- rows, then columns, peak in [0.5, 1): one sweep of a matrix balancing,
//     exact (powers of two).  S therefore follows a row's largest CONTRIBUTION, and what fp16's subnormals cost -- an
//     entry 2^-r of the largest keeps 2^-24 relative up to r = 13, beyond that an ABSOLUTE 2^-40 of the largest -- is
//     relative to each output column's own scale: the same dynamics in other units (D^-1 A D: observation entries 10^6 apart
//     with weights to match) come out as they went in (test_wide_split_planes_in_mixed_units), and an entry the model
//     ignores takes no part.  What is left outside: contributions that differ by more than 2^13 inside a BALANCED model
//     (one state entry of 10^6 feeding a column beside entries of 1 feeding others): there the small columns see 1e-6
//     relative per step instead of 6e-8 -- the bf16 planes below (exact operands) are for such a model.  Measured against
//     the float64 oracle over 30 tanh steps at o = 378: 1-4 x 10^-6 relative, the exact-f32 kernel's own distance
//     (tools/dbg/split_tile_errors.py; tests: observations from 1e-30 to 1e9 in test_wide_fp16_planes_follow_the_magnitudes).
//   * bf16, three planes, SIX products (icem_set_wide_exact 2; round 4's first form).  x = hi + mid + lo EXACTLY (3 x 8
//     significant bits, no scale needed: bf16 has f32's exponent), products hi*Hi, hi*Mid, mid*Hi, hi*Lo, lo*Hi, mid*Mid
//     on v_mfma_f32_16x16x32_bf16; what is dropped is below 2^-31 of a product.  Two thirds of the fp16 form's speed.
// Either way the result is an f32 dot product with f32-class rounding, not the bits of an fmaf chain (error analysis:
// DESIGN.md section 4); a row's result depends on that row and the model only, wherever the row sits in a launch.
//
// Tiling (numbers for the bf16 form; the fp16 one has two planes where it has three, three products where it has six).  A
// 16-row tile per wave (k_rollout_wide.hip) streams the whole model through every wave every step; in three bf16 planes
// that is 958 KB per 16 rows and step -- more than a CU's L2 port delivers.  So a WORKGROUP owns up to 80
// trajectories (five 16-row tiles; four is the regular batch, the fifth takes the few rows a population leaves behind
// whole batches -- the shifted elites of iteration 0, icem.py:131-137 -- instead of a second round of workgroups):
//   * their contraction vectors [obs | action | 0-pad] live in LDS as f32 rows X[80][XS] (134 KB: one workgroup per CU);
//   * wave w of 8 (two per SIMD) owns the output column tiles NCT w .. NCT w + NCT - 1 for ALL the workgroup's
//     trajectories -- NCT x 4 (5) accumulator tiles -- and streams only ITS eighth of the model: per 32-deep contraction
//     block 3 planes x NCT 16-byte loads per lane into a second register set, requested BETWEEN the MFMAs of the block
//     before, one load per 8 MFMAs (all nine at the block's top queue up in front of the CU's one texture path, ~21 cycles
//     per 1 KB, and a wave whose load is not accepted yet does not issue the MFMAs behind it: 900 -> 823 us per launch);
//   * the B operand's planes are made ONCE per block for the whole workgroup -- thread u of the first 4 x rows splits
//     the 8 contraction entries (row u / 4, slot u % 4) -- and shared through a double-buffered 24 KB of LDS behind X: block
//     kb + 1 is split beside block kb's MFMAs, one barrier per block.  (Every wave splitting the same values for itself, the
//     first form of this kernel, asked the issue port for 3.3 VALU per MFMA where 3 fit: EXPERIMENTS R4.4.)
//   * per block a wave reads the planes of all its tiles (12 ds_read_b128) and runs column tile by column tile, 6 products x
//     4 tiles, consecutive MFMAs on different accumulators; per accumulator the order of the six products never changes
//     (Lo*hi, Hi*lo, Mid*mid, Mid*hi, Hi*mid, Hi*hi), so every form of this kernel returned the same bits;
//   * a launch in which some workgroup is left with FIVE tiles runs the FIVE instantiation: the five-tile batch in two tile
//     groups (3 + 2) with ONE operand set (requests over the operands just used).  Kept out of the regular kernel because
//     its 60 accumulators set the whole kernel's register allocation (+14 % on launches that never see a fifth tile), and
//     kept to one set because two spill > 100 VGPRs at 256 (11 with one);
//   * the step's actions are requested one step ahead and held in registers across the model loop;
//   * two workgroup barriers per step around the model loop's (X read by everybody -> X rewritten column block by column block);
//   * step cost, candidate lists and the running top-K as in k_rollout_wide.hip (trajectory tile tt is scored by wave tt's
//     lanes 0..15 from X); icem_cost_terms included (EXT).
// Model operand layout (pack_wide_model_split): Mb[kb][wave][ct][plane][lane] = 8 x 16 bits = M[32 kb + 8 (lane / 16) + v]
// [16 (NCT wave + ct) + lane % 16], planes in the order lo, (mid,) hi.
#include <cmath>
#include <type_traits>
#include "fused_dev.h"
#include "wide_dev.h"

namespace icem {

namespace {

constexpr int SPLIT_TT = 5;      // trajectory tiles per workgroup batch (regular batches take 4)
constexpr int SPLIT_WAVES = 8;
constexpr int SPLIT_KMAX = 416;   // contraction length (o + d, padded to 32) the fp16 form's per-entry scales have LDS for
// the shared planes: two buffers of (up to) 3 planes x 64 rows x 32 x 16 bits for the regular batches (one of 80 rows for the five-tile batch fits inside)
constexpr size_t SPLIT_PLANE_BYTES = (size_t)2 * 3 * 16 * (SPLIT_TT - 1) * 64;
static_assert(SPLIT_PLANE_BYTES >= (size_t)3 * 16 * SPLIT_TT * 64 && SPLIT_PLANE_BYTES >= 2 * SPLIT_WAVES * 32 * sizeof(unsigned long long), "planes buffer");

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x -> (hi, residual): hi = bf16(x) round-to-nearest-even, both of a pair in one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned split_pair(float& a, float& b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    const unsigned u = __builtin_bit_cast(unsigned, h);
    // exact: the residual has at most 16 significant bits.  (Spelled as two v_sub_f32: the compiler's v_pk_add_f32 is
    // an expensive neighbour of MFMAs -- MI355X guide, "price of one filler beside MFMAs".)
    const float fa = __uint_as_float(u << 16), fb = __uint_as_float(u & 0xFFFF0000u);
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(fa));
    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(b) : "v"(fb));
    return u;
}

struct Planes {
    u32x4 hi, mid, lo;
};
// the three bf16 planes of 8 f32 values (a lane's share of one 32-deep contraction block)
__device__ __forceinline__ Planes split8(float4 p, float4 q) {
    float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
    Planes r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.hi[i] = split_pair(x[2 * i], x[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) r.mid[i] = split_pair(x[2 * i], x[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x2 h = __builtin_convertvector(f32x2{x[2 * i], x[2 * i + 1]}, bf16x2);
        r.lo[i] = __builtin_bit_cast(unsigned, h);
    }
    return r;
}

// The two fp16 planes of 8 f32 values scaled by S (a power of two chosen per trajectory row so that |x| S < 2^15):
// hi = f16(x S), lo = f16(x S - hi), both round-to-nearest-even (v_cvt_pk_f16_f32).  hi + lo is x S to 2^-24 relative
// (11 + 11 significant bits and lo's sign), lo is a normal fp16 number for every entry within 2^-13 of the row's largest
// and carries an absolute error below 2^-40 of that largest entry otherwise.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split8h(float4 p, float4 q, float4 cp, float4 cq, float S, u32x4& hi, u32x4& lo) {
    const float x[8] = {p.x * cp.x, p.y * cp.y, p.z * cp.z, p.w * cp.w, q.x * cq.x, q.y * cq.y, q.z * cq.z, q.w * cq.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 v = {x[2 * i] * S, x[2 * i + 1] * S};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f32x2 b = __builtin_convertvector(h, f32x2);
        const f16x2 l = __builtin_convertvector(f32x2{v[0] - b[0], v[1] - b[1]}, f16x2);
        hi[i] = __builtin_bit_cast(unsigned, h);
        lo[i] = __builtin_bit_cast(unsigned, l);
    }
}

template <bool F16>
__device__ __forceinline__ f32x4 mma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One batch of NTT trajectory tiles (rows [row0, row0 + 16 ntt), ntt <= NTT: tiles beyond ntt are computed on whatever their
// LDS rows hold and dropped -- no predicate inside the model loop) through all H steps.  m0 holds contraction block 0 of the
// wave's share of the model on entry and on exit.
template <int NCT, int NTT, int KIND, bool EXT, bool ONESET, bool F16, typename Req>
__device__ __forceinline__ void split_batch(const WideRolloutArgs& a, float* X, unsigned char* P, float* rscale, float* rinv, const float* ksc, const float* csc,
                                            const CostArgs<float>& cs_s,
                                            int row0, int ntt, int tid, int lane, int wave, u32x4 (&m0)[NCT * (F16 ? 2 : 3)],
                                            u32x4 (&m1)[NCT * (F16 ? 2 : 3)], Req&& request1, unsigned long long& run_key, bool& first) {
    constexpr int NPL = F16 ? 2 : 3;   // planes per operand: fp16 (hi, lo) / bf16 (hi, mid, lo)
    const int j = lane & 15, g = lane >> 4;
    const int XS = a.xs, KB = a.kb, o = a.o, d = a.d, H = a.h;
    const WideCost wc{a.lin_idx, a.flip_idx, a.ctrl_w, a.lin_w, a.flip_pen, a.flip_th};
    const bool sweep = EXT && cs_s.health_idx >= 0, diff = EXT && cs_s.diff_idx >= 0;
    const int nrow = 16 * ntt;
    __syncthreads();   // (the previous batch's readers are done with X)
    for (int e = tid; e < 16 * NTT * XS; e += 64 * SPLIT_WAVES) {
        const int c = e % XS;
        X[e] = c < o ? a.obs0[c] : 0.f;
    }
    // cost bookkeeping of the trajectory tile this wave scores (tt = wave), in lanes 0..15
    constexpr int NS = (SPLIT_TT + SPLIT_WAVES - 1) / SPLIT_WAVES;
    float acc_c[NS] = {}, c_prev[NS] = {}, dold[NS] = {};
    auto score = [&](int t) {   // finish step t - 1 (its difference term reads the observation now in X), start step t
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int tt = wave + SPLIT_WAVES * s;
            if (tt >= ntt) continue;   // (wave-uniform)
            const float* xt = X + (size_t)(16 * tt) * XS;
            if (t > 0 && lane < 16) {
                float c = c_prev[s];
                if (diff) c += wide_diff_cost(cs_s, xt[lane * XS + cs_s.diff_idx], dold[s]);
                acc_c[s] = wide_accumulate(acc_c[s], c, t - 1, a.cost_mode);
            }
            if (t < H) {
                bool bad = false;
                if (sweep) {   // lane (j, g) sweeps entries g, g + 4, .. of row j
                    const float* xj = xt + j * XS;
                    bool b = false;
                    for (int k = g; k < o; k += 4) b |= wide_bad_entry(cs_s, xj[k], k);
                    const unsigned long long mk = __ballot(b);
                    bad = ((mk >> (lane & 15)) & 0x0001000100010001ull) != 0ull;
                }
                if (lane < 16) c_prev[s] = wide_step_cost(wc, EXT, cs_s, xt + lane * XS, o, d, bad, dold[s]);
            }
        }
    };
    // The step's actions -> X[:, o .. o + d): element e = tid + 256 i of the batch's [nrow, d] block.  Step t + 1's are
    // requested at the top of step t and stay in registers across the model loop (a load -> LDS store -> barrier sequence
    // per step would expose one global round trip per step and element: measured, a third of the launch).
    constexpr int AE = 4;   // elements per thread held ahead: covers d <= 25 at 80 rows, d <= 32 at 64
    const bool ahead = nrow * d <= 64 * SPLIT_WAVES * AE;
    int aoff[AE], xoff[AE];
    float an[AE];
#pragma unroll
    for (int i = 0; i < AE; ++i) {
        const int e = tid + 64 * SPLIT_WAVES * i;
        const int r = e / d, c = e - r * d;
        const bool in = e < nrow * d;
        xoff[i] = in ? r * XS + o + c : -1;
        aoff[i] = (in && row0 + r < a.n_rows) ? (r * H) * d + c : -1;   // relative to the batch's first row
    }
    const float* abase = a.actions + (size_t)row0 * H * d;
    auto load_actions = [&](int t) {
#pragma unroll
        for (int i = 0; i < AE; ++i) an[i] = aoff[i] >= 0 ? abase[aoff[i] + t * d] : 0.f;
    };
    auto store_actions = [&](int t) {
        if (ahead) {
#pragma unroll
            for (int i = 0; i < AE; ++i)
                if (xoff[i] >= 0) X[xoff[i]] = an[i];
        } else {
            for (int e = tid; e < nrow * d; e += 64 * SPLIT_WAVES) {
                const int r = e / d, c = e - r * d;
                const int row = row0 + r;
                X[r * XS + o + c] = row < a.n_rows ? a.actions[((size_t)row * H + t) * d + c] : 0.f;
            }
        }
    };
    if (ahead) load_actions(0);
    __syncthreads();   // the zeros above and the actions below meet in X's action slots, written by different threads
    store_actions(0);
    long long* st = (a.dbg && blockIdx.x == 3 && tid == 64) ? a.dbg : nullptr;   // development: phase stamps of step 5 (tools/dbg/split_stamps.py)
    for (int t = 0; t < H; ++t) {
        if (st && t == 5) st[0] = wall_clock64();
        __syncthreads();
        if (st && t == 5) st[1] = wall_clock64();
        score(t);
        if (st && t == 5) st[2] = wall_clock64();
        if (ahead && t + 1 < H) load_actions(t + 1);
        f32x4 acc[NCT][NTT];
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) acc[c][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the shared planes P[buffer][plane][row][32 bf16]: regular batches double-buffer (block kb + 1 is split while block kb
        // is multiplied, one barrier per block); the five-tile batch has room for one buffer (two barriers per block)
        constexpr bool DB = NTT < SPLIT_TT;
        constexpr int ROWS = 16 * NTT;
        float S = 1.f;   // (fp16 planes) the power of two of this splitter thread's row
        auto split_row = [&](int r, int gg, int kbn, int buf) {
            const float* xp = X + (size_t)r * XS + 32 * kbn + 8 * gg;
            unsigned char* q = P + ((size_t)(buf * NPL) * ROWS + r) * 64 + gg * 16;
            if constexpr (F16) {
                u32x4 hi, lo;
                const float* cp = ksc + 32 * kbn + 8 * gg;
                split8h(*reinterpret_cast<const float4*>(xp), *reinterpret_cast<const float4*>(xp + 4), *reinterpret_cast<const float4*>(cp),
                        *reinterpret_cast<const float4*>(cp + 4), S, hi, lo);
                *reinterpret_cast<u32x4*>(q) = hi;
                *reinterpret_cast<u32x4*>(q + (size_t)ROWS * 64) = lo;
            } else {
                const Planes b = split8(*reinterpret_cast<const float4*>(xp), *reinterpret_cast<const float4*>(xp + 4));
                *reinterpret_cast<u32x4*>(q) = b.hi;
                *reinterpret_cast<u32x4*>(q + (size_t)ROWS * 64) = b.mid;
                *reinterpret_cast<u32x4*>(q + (size_t)2 * ROWS * 64) = b.lo;
            }
        };
        // blocks 1 .. : bf16 planes by the first 4 x ROWS threads, 8 entries each; fp16 planes by ALL threads, 4 entries each
        // (EIGHT per row) -- with the entries' scales to read and apply, four waves splitting for eight made the block's
        // critical path: their MFMAs began when their SIMD partners' had ended (k-loop 13.3 -> 15.3 us per step)
        float S8[F16 ? (NTT < SPLIT_TT ? 1 : 2) : 1] = {};   // the power of two of this thread's row (rows: one per 512 threads' pass)
        auto splitn = [&](int kbn, int buf) {
            if constexpr (!F16) {
                if (tid < 4 * ROWS) split_row(tid >> 2, tid & 3, kbn, buf);
            } else {
#pragma unroll
                for (int ps = 0; ps < (NTT < SPLIT_TT ? 1 : 2); ++ps) {
                    const int u = tid + 64 * SPLIT_WAVES * ps;
                    if (u < 8 * ROWS) {
                        const int r = u >> 3, g8 = u & 7;
                        const float4 x4 = *reinterpret_cast<const float4*>(X + (size_t)r * XS + 32 * kbn + 4 * g8);
                        const float4 c4 = *reinterpret_cast<const float4*>(ksc + 32 * kbn + 4 * g8);
                        const float sc = S8[ps];
                        const f32x2 v0 = {x4.x * c4.x * sc, x4.y * c4.y * sc}, v1 = {x4.z * c4.z * sc, x4.w * c4.w * sc};
                        const f16x2 h0 = __builtin_convertvector(v0, f16x2), h1 = __builtin_convertvector(v1, f16x2);
                        const f32x2 b0 = __builtin_convertvector(h0, f32x2), b1 = __builtin_convertvector(h1, f32x2);
                        const f16x2 l0 = __builtin_convertvector(f32x2{v0[0] - b0[0], v0[1] - b0[1]}, f16x2);
                        const f16x2 l1 = __builtin_convertvector(f32x2{v1[0] - b1[0], v1[1] - b1[1]}, f16x2);
                        unsigned char* q = P + ((size_t)(buf * NPL) * ROWS + r) * 64 + g8 * 8;
                        *reinterpret_cast<uint2*>(q) = uint2{__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1)};
                        *reinterpret_cast<uint2*>(q + (size_t)ROWS * 64) = uint2{__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1)};
                    }
                }
            }
        };
        // Block 0 of a step.  bf16 planes: like every block.  fp16 planes: by the threads four waves on (the waves that
        // score nothing at the step's top), which first find the row's scale -- four threads per row scan it, S = 2^(141 - e)
        // for a largest entry 1.m x 2^(e - 127), so |x| S < 2^15 (fp16 holds 65 504), exact powers of two throughout;
        // 1 / (S x the model's scale) waits in rinv[] for the write-back.  A NaN entry does not move the maximum and poisons
        // its own trajectory only (one B-operand row = one output column); an infinite one makes S 2^-114 and stays infinite.
        auto split_first = [&](int t) {
            if constexpr (!F16) {
                splitn(0, 0);
            } else {
                const int u = (tid + 4 * 64) & (64 * SPLIT_WAVES - 1);
                if (u < 4 * ROWS) {
                    const int r = u >> 2, gg = u & 3;
                    const float* xr = X + (size_t)r * XS + 4 * gg;
                    const float* cr = ksc + 4 * gg;
                    // (tanh model, t > 0: the state entries are below 1 -- their scaled bound a.sbound stands in for them, only
                    //  the actions behind them are scanned: an upper bound of the largest contribution is all S needs)
                    const bool bounded = KIND == 1 && t > 0;
                    float mx = bounded ? a.sbound : 0.f;
                    for (int k = bounded ? ((o / 4) & ~3) : 0; k < 8 * KB; k += 4) {   // float4s gg, gg + 4, ..: the row's 32 KB entries, a quarter each
                        if (4 * (gg + k) < 32 * KB) {
                            float4 v = *reinterpret_cast<const float4*>(xr + 4 * k);
                            const float4 c = *reinterpret_cast<const float4*>(cr + 4 * k);
                            v.x *= c.x; v.y *= c.y; v.z *= c.z; v.w *= c.w;
                            mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(v.x)), __builtin_fmaxf(__builtin_fabsf(v.y), __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w))));
                        }
                    }
                    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 1));
                    mx = __builtin_fmaxf(mx, __shfl_xor(mx, 2));
                    int e = (int)(__float_as_uint(mx) >> 23);
                    e = e < 40 ? 40 : e;   // (an all-zero row: any scale)
                    const float Sr = __uint_as_float((unsigned)(268 - e) << 23);
                    if (gg == 0) {
                        rscale[r] = Sr;
                        rinv[r] = __uint_as_float((unsigned)(e - 14) << 23) * a.minv;
                    }
                    const float keep = S;
                    S = Sr;
                    split_row(r, gg, 0, 0);
                    S = keep;
                }
            }
        };
        // A block: the planes of all NTT tiles are read once, then column tile by column tile -- 6 products x NTT tiles on NTT
        // different accumulators -- with the NEXT block's operands of that column tile requested between its MFMAs (two sets:
        // every operand is asked for exactly one block before its use; one set: over itself, behind its last use).
        static_assert(ONESET || NTT < SPLIT_TT, "the two-set form has one tile group");
        auto block = [&](const u32x4 (&m)[NCT * NPL], int buf, u32x4 (&mn)[NCT * NPL], int kbn) {
            const unsigned char* q0 = P + ((size_t)(buf * NPL) * ROWS + j) * 64 + g * 16;
            auto feed = [&](int e) {
                __builtin_amdgcn_sched_barrier(0);
                mn[e] = request1(kbn, e);
                __builtin_amdgcn_sched_barrier(0);
            };
            // (the five-tile batch in two groups of tiles, 3 + 2: see the head of the file)
            constexpr int T0 = NTT < SPLIT_TT ? NTT : 3;
            auto group = [&](auto t_lo, auto t_n, bool first, bool last) {
                constexpr int TL = decltype(t_lo)::value, TN = decltype(t_n)::value;
                // planes of the group's tiles: [0] hi, [1] mid (bf16) / lo (fp16), [2] lo (bf16)
                u32x4 bp[NPL][TN];
                if (!first) __builtin_amdgcn_sched_barrier(0);   // (or the second group's planes are read beside the first's)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const int src = pl == 0 ? 0 : (NPL - pl);   // read order hi, lo, (mid): the order the products want them in
#pragma unroll
                    for (int tt = 0; tt < TN; ++tt) bp[src][tt] = *reinterpret_cast<const u32x4*>(q0 + (size_t)(16 * (TL + tt) + src * ROWS) * 64);
                }
                // products, smallest first; model operands m[NPL c + ..]: planes in the order lo, (mid), hi
                constexpr int NPROD = F16 ? 3 : 6;
                constexpr int PA[6] = {0, NPL - 1, F16 ? 1 : 1, 1, 2, 2};          // A: Lo, Hi, [Hi] | Mid, Mid, Hi, Hi
                constexpr int PB[6] = {0, NPL - 1, F16 ? 0 : 1, 0, 1, 0};          // B: hi, lo, [hi] | mid, hi, mid, hi
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
#pragma unroll
                    for (int p = 0; p < NPROD; ++p) {
                        // product by product over the group's tiles: consecutive MFMAs write different accumulators
#pragma unroll
                        for (int tt = 0; tt < TN; ++tt) acc[c][TL + tt] = mma<F16>(m[NPL * c + PA[p]], bp[PB[p]][tt], acc[c][TL + tt]);
                        if (!ONESET && first && last) {   // the next block's operands of this column tile, spread over its products
#pragma unroll
                            for (int i = 0; i < NPL; ++i)
                                if (p == (i + 1) * NPROD / NPL - 1) feed(NPL * c + i);
                        }
                    }
                    if (ONESET && last) {   // over the operands just used for the last time
#pragma unroll
                        for (int i = 0; i < NPL; ++i) feed(NPL * c + i);
                    }
                }
            };
            group(std::integral_constant<int, 0>{}, std::integral_constant<int, T0>{}, true, T0 == NTT);
            if constexpr (T0 < NTT) group(std::integral_constant<int, T0>{}, std::integral_constant<int, NTT - T0>{}, false, true);
        };
        // behind a block: everybody is done with its planes (and the next block's are complete, if they were made beside it)
        auto behind = [&](int kbn) {
            __syncthreads();
            if (!DB && kbn >= 0) {
                splitn(kbn, 0);
                __syncthreads();
            }
        };
        split_first(t);
        __syncthreads();
        if constexpr (F16) {
#pragma unroll
            for (int ps = 0; ps < (NTT < SPLIT_TT ? 1 : 2); ++ps)
                if (tid + 64 * SPLIT_WAVES * ps < 8 * ROWS) S8[ps] = rscale[(tid + 64 * SPLIT_WAVES * ps) >> 3];
        }
        int kb = 0;
        if constexpr (!ONESET) {
            // two register sets of model operands, a block's requested while the block before it runs; beside the last block: block 0 of the NEXT step
#pragma unroll 1
            for (; kb + 1 < KB; kb += 2) {
                splitn(kb + 1, 1);
                block(m0, 0, m1, kb + 1);
                behind(kb + 1);
                if (kb + 2 < KB) splitn(kb + 2, 0);
                block(m1, 1, m0, kb + 2 < KB ? kb + 2 : 0);
                behind(kb + 2 < KB ? kb + 2 : -1);
            }
            if (kb < KB) {   // odd block count: the last one (outside the loop: a conditional block inside it costs a copy of every accumulator per trip)
                block(m0, 0, m1, 0);   // (the next step's block 0 lands in m1: moved to m0 at the step's end, when it has long arrived)
            }
        } else {
#pragma unroll 1
            for (; kb < KB; ++kb) {
                if (DB && kb + 1 < KB) splitn(kb + 1, (kb + 1) & 1);
                block(m0, DB ? (kb & 1) : 0, m0, kb + 1 < KB ? kb + 1 : 0);
                behind(kb + 1 < KB ? kb + 1 : -1);
            }
        }
        if (st && t == 5) st[3] = wall_clock64();
        if (a.dbg && blockIdx.x == 3 && lane == 0 && t == 5) a.dbg[8 + wave] = wall_clock64();   // every wave's loop end
        __syncthreads();   // everybody has read X: the new observation may go in
        if (st && t == 5) st[4] = wall_clock64();
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int col = 16 * (NCT * wave + c) + 4 * g;
#pragma unroll
            for (int tt = 0; tt < NTT; ++tt) {
                f32x4 v = acc[c][tt];
                if constexpr (F16) {   // the row's, the model's and the column's powers of two taken out again (exact)
                    const float iv = rinv[16 * tt + j];
                    const float4 cs4 = *reinterpret_cast<const float4*>(csc + col);
                    v[0] *= iv * cs4.x;
                    v[1] *= iv * cs4.y;
                    v[2] *= iv * cs4.z;
                    v[3] *= iv * cs4.w;
                }
                if (KIND == 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fast_tanh(v[k]);
                }
                float* dst = X + (size_t)(16 * tt + j) * XS + col;
                if (col + 3 < o) {
                    *reinterpret_cast<f32x4*>(dst) = v;
                } else {   // the column group that straddles o: the action slots behind it belong to other threads
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (col + k < o) dst[k] = v[k];
                }
            }
        }
        if (st && t == 5) st[5] = wall_clock64();
        if (t + 1 < H) store_actions(t + 1);
        if (!ONESET && (KB & 1)) {
#pragma unroll
            for (int e = 0; e < NCT * NPL; ++e) m0[e] = m1[e];
        }
        if (st && t == 5) st[6] = wall_clock64();
        if (st && t == 6) st[7] = wall_clock64();
    }
    __syncthreads();
    score(H);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int tt = wave + SPLIT_WAVES * s;
        if (tt >= ntt) continue;
        const int row = row0 + 16 * tt + (lane & 15);
        const bool live = row < a.n_rows;
        if (live && lane < 16) a.costs[row] = acc_c[s];
        if (a.K > 0) {
            const unsigned long long key = (lane < 16 && live && row < a.n_cand) ? make_key(acc_c[s], row) : KEY_SENTINEL;
            run_key = topk_push16(run_key, key, first, a.K, lane);
            first = false;
        }
    }
}

// FIVE: some workgroup's tile count leaves a remainder of five (one batch instead of 4 + 1).  Its own instantiation: the
// five-tile batch's 60 accumulators set the register allocation of the whole kernel, and the four-tile batches of a launch
// that never sees one were 14 % slower for carrying it.
template <int NCT, int KIND, bool EXT, bool FIVE, bool F16>
__global__ __launch_bounds__(64 * SPLIT_WAVES) void rollout_wide_split_kernel(WideRolloutArgs a) {
    constexpr int NPL = F16 ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) float X[];  // [16 * SPLIT_TT][XS] f32 rows, then the planes' buffers
    __shared__ CostArgs<float> cs_s;
    __shared__ float rscale[F16 ? 16 * SPLIT_TT : 1], rinv[F16 ? 16 * SPLIT_TT : 1];   // fp16 planes: the rows' powers of two
    __shared__ __attribute__((aligned(16))) float ksc_s[F16 ? SPLIT_KMAX : 4];       // ... the contraction entries' ...
    __shared__ __attribute__((aligned(16))) float csc_s[F16 ? 16 * SPLIT_WAVES * NCT : 4];   // ... and the output columns' (pack_wide_model_split)
    if (F16) {   // (split_batch opens with a barrier)
        for (int e = threadIdx.x; e < 32 * a.kb; e += 64 * SPLIT_WAVES) ksc_s[e] = a.ksc[e];
        for (int e = threadIdx.x; e < 16 * SPLIT_WAVES * NCT; e += 64 * SPLIT_WAVES) csc_s[e] = a.csc[e];
    }
    unsigned char* P = reinterpret_cast<unsigned char*>(X + (size_t)16 * SPLIT_TT * a.xs);   // SPLIT_PLANE_BYTES
    // (the workgroup's candidate-list scratch lies over the planes: used behind the last batch only)
    auto wg_keys = reinterpret_cast<unsigned long long(*)[SPLIT_WAVES][32]>(P);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (EXT) wide_stage_terms(cs_s, a.cs, tid, 64 * SPLIT_WAVES);
    // this wave's share of the model: [kb][wave][ct][plane][lane] 16-byte vectors
    typedef const __attribute__((address_space(1))) u32x4* gvec;
    gvec Mw = (gvec)a.Mp + (size_t)wave * NCT * NPL * 64 + lane;
    const size_t kb_stride = (size_t)SPLIT_WAVES * NCT * NPL * 64;
    auto request1 = [&](int kb, int e) -> u32x4 { return Mw[(size_t)kb * kb_stride + (size_t)e * 64]; };
    auto request = [&](u32x4 (&m)[NCT * NPL], int kb) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < NCT * NPL; ++e) m[e] = request1(kb, e);
        __builtin_amdgcn_sched_barrier(0);
    };
    // tiles of this workgroup: T tiles over the grid, the remainder one each to the first workgroups
    const int tiles = (a.n_rows + 15) / 16;
    const int base = tiles / (int)gridDim.x, extra = tiles % (int)gridDim.x;
    int t_begin = (int)blockIdx.x * base + ((int)blockIdx.x < extra ? (int)blockIdx.x : extra);
    int cnt = base + ((int)blockIdx.x < extra ? 1 : 0);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    u32x4 m0[NCT * NPL], m1[NCT * NPL];
    request(m0, 0);
    while (cnt > 0) {   // batches of four tiles; a remainder of five is one batch
        int ntt = cnt < SPLIT_TT - 1 ? cnt : SPLIT_TT - 1;
        if constexpr (FIVE) {
            if (cnt == SPLIT_TT) ntt = SPLIT_TT;
        }
        if constexpr (FIVE) {
            if (ntt == SPLIT_TT)
                split_batch<NCT, SPLIT_TT, KIND, EXT, true, F16>(a, X, P, rscale, rinv, ksc_s, csc_s, cs_s, t_begin * 16, ntt, tid, lane, wave, m0, m1, request1, run_key, first);
            else
                split_batch<NCT, SPLIT_TT - 1, KIND, EXT, true, F16>(a, X, P, rscale, rinv, ksc_s, csc_s, cs_s, t_begin * 16, ntt, tid, lane, wave, m0, m1, request1, run_key, first);
        } else {
            split_batch<NCT, SPLIT_TT - 1, KIND, EXT, false, F16>(a, X, P, rscale, rinv, ksc_s, csc_s, cs_s, t_begin * 16, ntt, tid, lane, wave, m0, m1, request1, run_key, first);
        }
        cnt -= ntt;
        t_begin += ntt;
    }
    __syncthreads();   // (the planes are dead: their LDS becomes the list scratch)
    if (a.K > 0) {
        FastRolloutArgs fr{};  // wg_merge_emit only looks at the candidate outputs
        fr.part_k = a.part_k;
        fr.part_c = a.part_c;
        fr.part_i = a.part_i;
        wg_merge_emit<SPLIT_WAVES>(wg_keys, run_key, a.K, lane, wave, fr);
    }
}

__host__ unsigned short bf16_rne(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ float bf16_f32(unsigned short b) {
    const unsigned u = (unsigned)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

}  // namespace

int wide_split_kb(int o, int d) { return (o + d + 31) / 32; }
int wide_split_xs(int o, int d) { return 32 * wide_split_kb(o, d) + 4; }
// the workgroup's LDS: 80 rows of 32 kb + 4 floats, the planes' buffers and ~3 KB of static arrays in 160 KB: o + d <= 416
bool wide_split_fits(int o, int d) { return (size_t)16 * SPLIT_TT * wide_split_xs(o, d) * sizeof(float) + SPLIT_PLANE_BYTES + 4608 <= 160 * 1024 && 32 * wide_split_kb(o, d) <= SPLIT_KMAX; }
static int wide_split_nct(int o) { const int nt = (o + 15) / 16; return nt <= 8 ? 1 : nt <= 16 ? 2 : 3; }

// workgroups (= candidate lists): whole batches of four tiles, at most FAST_MAX_LISTS
int wide_split_lists(int n_rows) {
    const int tiles = std::max(1, (n_rows + 15) / 16);
    return std::min(std::max(1, tiles / (SPLIT_TT - 1)), FAST_MAX_LISTS);
}

// How far the EQUILIBRATED model (rows, then columns scaled by powers of two as pack_wide_model_split does for the fp16
// planes) stays from balanced: the largest log2(global max / line max) over its live rows and live columns.  0-1 for a
// model one sweep balances; a row or column whose LARGEST weight sits 2^13 below the model's largest has every entry where
// the fp16 pair no longer carries f32's 22+ bits relative to that line (the lo plane in fp16's subnormals) -- the model
// ICEM_WIDE_AUTO hands to the bf16 planes, whose operands are exact at any magnitude.  Incidental small entries inside a
// line that has a large one do not count (their error is absolute, 2^-40 of that line's largest; a dense random model has
// some 0.05 % of its entries 2^13 below its largest; the benchmark's 0.95 I + 0.05 N / sqrt(o) 3.5 % of every row) -- a
// STRUCTURAL spread does: a row or column in which MOST nonzero weights lie that far below its largest (a block of the
// model in other units than diagonal scaling can take out); the states that exercise those weights carry 10^-6 relative
// per step.  Returned: the larger of the two measures (log2 largest / weakest line's largest, log2 line's largest / its
// median nonzero weight, worst line).
int wide_model_imbalance_log2(int o, int d, const double* A, const double* B) {
    auto M = [&](int r, int c) -> double { return r < o ? A[(size_t)r * o + c] : B[(size_t)(r - o) * o + c]; };
    std::vector<int> ek((size_t)o + d, 0), fj((size_t)o, 0);
    std::vector<char> dead((size_t)o + d, 1);
    for (int r = 0; r < o + d; ++r) {
        float wmax = 0.f;
        bool any = false;
        for (int c = 0; c < o; ++c) {
            const float m = std::fabs((float)M(r, c));
            if (std::isfinite(m) && m > wmax) wmax = m;
            any = any || M(r, c) != 0.0;
        }
        if (!any) continue;
        int e = 0;
        if (wmax > 0.f) (void)std::frexp(wmax, &e);
        ek[r] = e < -100 ? -100 : (e > 100 ? 100 : e);
        dead[r] = 0;
    }
    std::vector<float> cmax((size_t)o, 0.f), rmax((size_t)o + d, 0.f);
    for (int c = 0; c < o; ++c) {
        for (int r = 0; r < o + d; ++r) {
            if (dead[r]) continue;
            const float m = std::fabs(std::ldexp((float)M(r, c), -ek[r]));
            if (std::isfinite(m) && m > cmax[c]) cmax[c] = m;
        }
        int f = 0;
        if (cmax[c] > 0.f) (void)std::frexp(cmax[c], &f);
        fj[c] = f < -100 ? -100 : (f > 100 ? 100 : f);
    }
    float gmax = 0.f;
    std::fill(cmax.begin(), cmax.end(), 0.f);
    for (int r = 0; r < o + d; ++r) {
        if (dead[r]) continue;
        for (int c = 0; c < o; ++c) {
            const float m = std::fabs(std::ldexp((float)M(r, c), -ek[r] - fj[c]));
            if (!std::isfinite(m)) continue;
            rmax[r] = std::max(rmax[r], m);
            cmax[c] = std::max(cmax[c], m);
            gmax = std::max(gmax, m);
        }
    }
    if (!(gmax > 0.f)) return 0;
    int worst = 0;
    {   // the structural spread: a line's largest weight over the MEDIAN of its nonzero weights
        std::vector<float> mags;
        auto spread = [&](int line_is_row, int idx) {
            mags.clear();
            float mx = 0.f;
            const int n = line_is_row ? o : o + d;
            for (int t = 0; t < n; ++t) {
                const int r = line_is_row ? idx : t, c = line_is_row ? t : idx;
                if (dead[r]) continue;
                const float m = std::fabs(std::ldexp((float)M(r, c), -ek[r] - fj[c]));
                if (std::isfinite(m) && m > 0.f) {
                    mags.push_back(m);
                    mx = std::max(mx, m);
                }
            }
            if (mags.size() < 2) return;
            std::nth_element(mags.begin(), mags.begin() + mags.size() / 2, mags.end());
            int em = 0, eq = 0;
            (void)std::frexp(mx, &em);
            (void)std::frexp(mags[mags.size() / 2], &eq);
            worst = std::max(worst, em - eq);
        };
        for (int r = 0; r < o + d; ++r)
            if (!dead[r]) spread(1, r);
        for (int c = 0; c < o; ++c) spread(0, c);
    }
    auto line = [&](float mx) {
        if (!(mx > 0.f)) return;   // (a column nothing feeds: zero in every arithmetic)
        int eg = 0, el = 0;
        (void)std::frexp(gmax, &eg);
        (void)std::frexp(mx, &el);
        worst = std::max(worst, eg - el);
    };
    for (int r = 0; r < o + d; ++r)
        if (!dead[r]) line(rmax[r]);
    for (int c = 0; c < o; ++c) line(cmax[c]);
    return worst;
}

// Mb[kb][wave][ct][plane][lane][v] = plane of (float)M[32 kb + 8 (lane / 16) + v][16 (NCT wave + ct) + lane % 16],
// M = [A ; B] ([o + d, o], zero padded).  planes = 3: bf16 lo, mid, hi.  planes = 2: fp16 lo, hi of M x 2^k, k such that the
// largest entry stays below 2^15; *minv = 2^-k.
void pack_wide_model_split(int o, int d, const double* A, const double* B, int planes, std::vector<unsigned short>& Mb, float* minv,
                           std::vector<float>* ksc, std::vector<float>* csc, float* sbound) {
    const int KB = wide_split_kb(o, d), NCT = wide_split_nct(o);
    Mb.assign((size_t)KB * SPLIT_WAVES * NCT * planes * 64 * 8, 0);
    auto M = [&](int r, int c) -> double {
        if (c >= o) return 0.0;
        if (r < o) return A[(size_t)r * o + c];
        if (r < o + d) return B[(size_t)(r - o) * o + c];
        return 0.0;
    };
    float SM = 1.f;
    // fp16 planes: contraction entry k (an observation / action entry) is carried as x_k 2^e_k and the model's row k as
    // M[k][:] 2^-e_k, 2^e_k the power of two of the row's largest weight -- every row of the scaled model peaks in [0.5, 1),
    // and the planes' per-trajectory scale follows the largest CONTRIBUTION x_k max|M[k][:]|, not the largest entry: an
    // observation in mixed units (positions of 1, forces of 10^4 with weights of 10^-4) keeps its small entries' bits, and an
    // entry the model ignores (a zero row: 2^e_k = 0) takes no part.  Exact: powers of two.
    std::vector<int> ek((size_t)32 * KB, 0);
    std::vector<char> dead((size_t)32 * KB, 1);
    if (planes == 2) {
        if (ksc) ksc->assign((size_t)32 * KB, 0.f);
        for (int r = 0; r < o + d; ++r) {
            float wmax = 0.f;
            for (int c = 0; c < o; ++c) {
                const float m = std::fabs((float)M(r, c));
                if (std::isfinite(m) && m > wmax) wmax = m;   // (a NaN / infinite weight: the scale of the finite ones, the weight stays what it is)
            }
            bool any = false;
            for (int c = 0; c < o; ++c) any = any || M(r, c) != 0.0;   // (NaN != 0: a row with a NaN weight is not dead)
            if (!any) continue;
            int e = 0;
            if (wmax > 0.f) (void)std::frexp(wmax, &e);
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            ek[r] = e;
            dead[r] = 0;
            if (ksc) (*ksc)[r] = std::ldexp(1.f, e);
        }
    }
    if (sbound) {   // the largest scaled state entry a tanh model can hand on: max over the observation entries of 2^e_k x 1
        *sbound = 0.f;
        if (planes == 2)
            for (int r = 0; r < o; ++r)
                if (!dead[r]) *sbound = std::max(*sbound, std::ldexp(1.f, ek[r]));
    }
    // ... and output column j as (sum) 2^f_j, 2^f_j the power of two of the column's largest (row-scaled) weight: the planes'
    // absolute accuracy (2^-40 of the row's largest contribution) is then relative to every COLUMN's own scale.  Rows, then
    // columns: the first sweep of a matrix balancing -- the same dynamics in other units (D^-1 A D) come out as A.
    std::vector<int> fj((size_t)o, 0);
    if (planes == 2) {
        if (csc) csc->assign((size_t)16 * SPLIT_WAVES * NCT, 1.f);
        for (int c = 0; c < o; ++c) {
            float cmax = 0.f;
            for (int r = 0; r < o + d; ++r) {
                if (dead[r]) continue;
                const float m = std::fabs(std::ldexp((float)M(r, c), -ek[r]));
                if (std::isfinite(m) && m > cmax) cmax = m;
            }
            int f = 0;
            if (cmax > 0.f) (void)std::frexp(cmax, &f);
            f = f < -100 ? -100 : (f > 100 ? 100 : f);
            fj[c] = f;
            if (csc) (*csc)[c] = std::ldexp(1.f, f);
        }
    }
    auto Ms = [&](int r, int c) -> float {   // the model as the fp16 planes see it
        if (r >= o + d || c >= o || dead[r]) return 0.f;
        return std::ldexp((float)M(r, c), -ek[r] - fj[c]);
    };
    if (planes == 2) {
        float mx = 0.f;
        for (int r = 0; r < o + d; ++r)
            for (int c = 0; c < o; ++c) {
                const float m = std::fabs(Ms(r, c));
                if (std::isfinite(m) && m > mx) mx = m;
            }
        int e = 0;
        if (mx > 0.f && std::isfinite(mx)) (void)std::frexp(mx, &e);   // mx = f x 2^e, f in [0.5, 1)
        else e = 1;
        SM = std::ldexp(1.f, 15 - e);
        if (minv) *minv = std::ldexp(1.f, e - 15);
    } else if (minv) {
        *minv = 1.f;
    }
    for (int kb = 0; kb < KB; ++kb)
        for (int w = 0; w < SPLIT_WAVES; ++w)
            for (int ct = 0; ct < NCT; ++ct)
                for (int lane = 0; lane < 64; ++lane)
                    for (int v = 0; v < 8; ++v) {
                        const float m = (float)M(32 * kb + 8 * (lane / 16) + v, 16 * (NCT * w + ct) + lane % 16);
                        if (planes == 2) {
                            const float ms = Ms(32 * kb + 8 * (lane / 16) + v, 16 * (NCT * w + ct) + lane % 16) * SM;
                            const _Float16 hi = (_Float16)ms;
                            const _Float16 lo = (_Float16)(ms - (float)hi);
                            const size_t at = ((((size_t)kb * SPLIT_WAVES + w) * NCT + ct) * 2) * 64 * 8 + (size_t)lane * 8 + v;
                            std::memcpy(&Mb[at], &lo, 2);
                            std::memcpy(&Mb[at + 64 * 8], &hi, 2);
                            continue;
                        }
                        const unsigned short hi = bf16_rne(m);
                        const float r1 = m - bf16_f32(hi);
                        const unsigned short mid = bf16_rne(r1);
                        const float r2 = r1 - bf16_f32(mid);
                        const unsigned short lo = bf16_rne(r2);
                        const size_t at = ((((size_t)kb * SPLIT_WAVES + w) * NCT + ct) * 3) * 64 * 8 + (size_t)lane * 8 + v;
                        Mb[at] = lo;
                        Mb[at + 64 * 8] = mid;
                        Mb[at + 2 * 64 * 8] = hi;
                    }
}

void launch_rollout_wide_split(const WideRolloutArgs& a, int kind, hipStream_t st) {
    const int grid = wide_split_lists(a.n_rows);
    const size_t lds = (size_t)16 * SPLIT_TT * a.xs * sizeof(float) + SPLIT_PLANE_BYTES;
    const int NCT = wide_split_nct(a.o);
    // a batch of five: some workgroup holds 4 q + 1 tiles, q >= 1
    const int tiles = (a.n_rows + 15) / 16, base = tiles / grid, extra = tiles % grid;
    const bool five = (base >= SPLIT_TT && base % (SPLIT_TT - 1) == 1) || (extra > 0 && base + 1 >= SPLIT_TT && (base + 1) % (SPLIT_TT - 1) == 1);
#define XW3(NV, KINDV, EXTV, FV, HV)                                                                        \
    {                                                                                                       \
        auto kfn = rollout_wide_split_kernel<NV, KINDV, EXTV, FV, HV>;                                      \
        (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);  \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * SPLIT_WAVES), lds, st, a);                            \
    }
#define XW2(NV, KINDV, EXTV, FV) \
    if (a.planes == 2) XW3(NV, KINDV, EXTV, FV, true) else XW3(NV, KINDV, EXTV, FV, false)
#define XW1(NV, KINDV, EXTV) \
    if (five) XW2(NV, KINDV, EXTV, true) else XW2(NV, KINDV, EXTV, false)
#define XW(NV)                                                     \
    if (NCT == NV) {                                               \
        if (kind == 1) {                                           \
            if (a.cs) XW1(NV, 1, true) else XW1(NV, 1, false)      \
        } else {                                                   \
            if (a.cs) XW1(NV, 0, true) else XW1(NV, 0, false)      \
        }                                                          \
        return;                                                    \
    }
    XW(1) XW(2) XW(3)
#undef XW
#undef XW1
#undef XW2
#undef XW3
}

}  // namespace icem
