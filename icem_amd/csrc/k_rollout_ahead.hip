// k_rollout_ahead.hip -- the noise-ahead iteration launch (large populations, world == 1; plan.hip::plan_step_ahead): ONE launch
// per CEM iteration whose workgroups play different roles,
//   rollout  (the first `n_roll` workgroups)  rollout16_kernel's work (one wavefront per 16 trajectories on
//            v_mfma_f32_16x16x4_f32, Tile16 / Stream16 of fused_dev.h) on a pool that holds RAW colored noise: the PREVIOUS
//            iteration's K3 + K4 (top-K of its candidate lists, elite gather, refit; icem.py:194-211) in the prologue, the
//            selection shared by all waves (merge_select_split: nothing hides it here), then every vector a wave loads
//            becomes clip(y * std + mean) (icem.py:79) between the prefetch registers and its LDS staging buffer
//            (Stream16::run_xf).  The last iteration writes the actions back in place (the caller's pool); the others
//            leave the noise where it is and the NEXT prologue maps the K elite rows once more (MergeSingleArgs::n_raw);
//   noise    (the next `n_noise` workgroups)  the raw colored noise of the NEXT sampling call (iteration i + 1, or
//            iteration 0 of the next MPC step) into the next pool: powerlaw_psd_gaussian needs no distribution
//            (icem.py:73-79), so it runs beside the rollout whose waves leave half of every SIMD's registers free;
//   shift    (one more workgroup, iteration 0 of every MPC step but the first)  the shifted elites (icem.py:91-104,
//            131-137): built like the sampler's extra workgroup builds them, rolled out by the workgroup's first wave, and
//            handed to the merge through the cost array (its kept-elite slot), as the single-launch kernel's tail rows.
// All roles in one launch on ONE stream: no events, no second queue (the two-stream form of this pipeline lost 10-15 us
// per cross-stream wait; profiles/r03_noise_ahead_*).  Same operations in the same order as sample_folded(_merge)_kernel +
// rollout16_kernel, so the same bits in every buffer (tests: test_noise_ahead_pipeline_equals_the_default_path).
#include "fused_dev.h"

namespace icem {

namespace {

constexpr int AHEAD_KREG = 12;

// LDS of a workgroup (dynamic, one buffer overlaid by the three roles), in floats
template <int H, int D, int O, int KIND, int WAVES>
struct AheadLds {
    using Stream = Stream16<H, D, O, KIND>;   // (the staging layout is the same for every tile arithmetic)
    using T16 = Tile16<H, D, O, KIND>;
    static constexpr int HD = H * D, NT = 64 * WAVES;
    // rollout role
    static constexpr int STAGE = 0;                                   // [WAVES][STG]; the selection's scratch lies over it
    static constexpr int KEYS = STAGE + WAVES * Stream::STG + (WAVES * Stream::STG) % 2;   // u64 [2][WAVES][32]
    static constexpr int DIST = KEYS + 2 * (2 * WAVES * 32);          // [2 HD]
    static constexpr int OBS = DIST + 2 * HD + (2 * HD) % 4;          // [32]
    static constexpr int SEL = OBS + 32;                              // u64 [64]
    static constexpr int WSEL = SEL + 2 * 64;                         // u64 [WAVES * 16]: every wave's K best, the kept elites behind them
    static constexpr int SLOT = WSEL + 2 * WAVES * 16;                // int [64]
    static_assert(WSEL % 4 == 0, "wsel is read two keys at a time");
    static constexpr int ROLL_END = SLOT + 64;
    static_assert(WAVES * 64 * 2 <= WAVES * Stream::STG, "per-wave compaction scratch fits the staging buffers");
    // noise role: [tpw, HD] tile of NT / D rows
    static constexpr int TPW = NT / D;
    static constexpr int NOISE_END = TPW * HD;
    // shift role: mean | std, then a 16-row tile the first wave rolls out
    static constexpr int SH_DIST = 0;
    static constexpr int SH_TILE = 2 * HD + (2 * HD) % 4;
    static constexpr int SHIFT_END = SH_TILE + T16::SLACK + 16 * HD + T16::TAIL + 32;
    static constexpr int FLOATS = ROLL_END > NOISE_END ? (ROLL_END > SHIFT_END ? ROLL_END : SHIFT_END) : (NOISE_END > SHIFT_END ? NOISE_END : SHIFT_END);
};

// (at most 128 registers whatever the workgroup size: the rollout waves share their SIMDs with the noise role's.  96 -- a
//  fifth wave per SIMD -- spills 32 registers in the rollout role: measured 220 instead of 185 us per MPC step at N = 65 536)
// PM: 0 = no merge (iteration 0), 1 = lists merge in every rollout workgroup's prologue, 2 = sharded: pack role + published
// records merge
template <int H, int D, int O, int KIND, int WAVES, int PM, int ARITH>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) void iter_ahead_kernel(IterAheadArgs args) {
#include "iter_ahead_body.h"
}

// B problems in one launch (blockIdx.y): the block of problem y from device memory, its pointers re-read as global addresses
// (fused_dev.h: from_device), the noise calls' stream offsets stored relative to the step's base of that problem
template <int H, int D, int O, int KIND, int WAVES, int PM, int ARITH>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) void iter_ahead_batch_kernel(
    const IterAheadArgs* __restrict__ arr, BatchBases bases) {
    static_assert(PM != 2, "sharded launches are not batched");
    const IterAheadArgs& g = arr[blockIdx.y];
    IterAheadArgs args = g;
    args.r = from_device(g.r);
    args.m = from_device(g.m);
    args.z = from_device(g.z);
    args.s = from_device(g.s);
    args.pool = gptr(g.pool);
    args.mean = gptr(g.mean);
    args.std = gptr(g.std);
    const unsigned long long base = bases.v[blockIdx.y];
    add_base64(args.z.off_lo, args.z.off_hi, base);
    add_base64(args.s.off2_lo, args.s.off2_hi, base);
#include "iter_ahead_body.h"
}

// Launch shape of the rollout role: rollout16's (one 16-trajectory tile per wave while they fit, at most FAST_MAX_LISTS
// workgroups), but at most 8 wavefronts per workgroup -- two per SIMD, each taking its tiles one after the other: the
// rollout leaves half of every SIMD's registers (and 100 of the CU's 160 KB of LDS) to a noise workgroup beside it.
void ahead_shape(int n_rows, int* grid, int* waves) {
    r16_shape(n_rows, grid, waves);
    if (*waves > 8) *waves = 8;
    if (g_batch.mult > 1 && g_batch.ahead) {
        // a batch of `mult` problems of this size in one launch (icem_plan_step_batch): waves per workgroup as for all their rows
        // together, the rollout workgroups of ONE problem (blockIdx.y is the problem)
        int g_all, w_all;
        r16_shape(n_rows * g_batch.mult, &g_all, &w_all);
        *waves = std::max(*waves, std::min(w_all, 8));
        *grid = (std::max(1, (n_rows + 15) / 16) + *waves - 1) / *waves;
    }
}

}  // namespace

// populations whose rollout launch has at least 4 waves per workgroup (more than 512 tiles): below that the
// single-launch kernel of k_iter_small.hip is the shorter chain
// ... and one-tile observation widths (O <= 20) only: with two output tiles (O = 24: 26 MFMAs and twice the state per
// step) the rollout role does not fit the 128 registers it shares its SIMDs on -- 112-158 spilled VGPRs, and the launch
// LOSES to the sampler + rollout16 pair: 236.1 vs 171.4 us per MPC step at N = 16 384 (d = 17, 3 iterations), 657.7 vs
// 473.7 at 65 536 (EXPERIMENTS.md R4.6).  Those shapes are not instantiated.
bool rollout_ahead_ok(int h, int d, int O, int K, int n_rows) {
    int grid, waves;
    r16_shape(n_rows * ((g_batch.mult > 1 && g_batch.ahead) ? g_batch.mult : 1), &grid, &waves);   // (a batch: the rows of all its problems fill the chip)
    return O <= 20 && K + 1 <= AHEAD_KREG && waves >= 4 && fast_rollout_supported(h, d, O, K) && fast_sample_supported(h, d);
}

int ahead_roll_workgroups(int n_rows) {
    int grid, waves;
    ahead_shape(n_rows, &grid, &waves);
    return grid;
}

void launch_iter_ahead(const IterAheadArgs& a_in, int h, int d, int O, int kind, hipStream_t st) {
    IterAheadArgs a = a_in;
    int grid, waves;
    ahead_shape(a.r.n_rows, &grid, &waves);
    a.n_roll = grid;
    if (g_batch.rec) {   // icem_plan_step_batch: recorded, launched for all problems at once (launch_iter_ahead_batch)
        if (a.has_merge == 2 || O > 20 || (waves != 4 && waves != 8)) {
            g_batch.unsupported = true;
            return;
        }
        BatchRecord r;
        r.kind = 4;
        a.n_noise = a.z.n > 0 ? (a.z.n + (64 * waves) / d - 1) / ((64 * waves) / d) : 0;   // (AheadLds::TPW = threads / d)
        r.ia = a;
        r.h = h, r.d = d, r.O = O, r.model_kind = kind, r.rw = waves, r.grid = grid;
        g_batch.rec->push_back(r);
        return;
    }
#define XK(HH, DD, OO, KK, WW, PP)                                                                                         \
    {                                                                                                                      \
        using L = AheadLds<HH, DD, OO, KK, WW>;                                                                            \
        a.n_noise = a.z.n > 0 ? (a.z.n + L::TPW - 1) / L::TPW : 0;                                                         \
        const int total = grid + a.n_noise + (a.s.n_shift > 0 ? 1 : 0) + (PP == 2 ? 1 : 0);                                \
        if (a.r.arith == 1)                                                                                                \
            hipLaunchKernelGGL((iter_ahead_kernel<HH, DD, OO, KK, WW, PP, 1>), dim3(total), dim3(64 * WW), L::FLOATS * sizeof(float), st, a); \
        else                                                                                                               \
            hipLaunchKernelGGL((iter_ahead_kernel<HH, DD, OO, KK, WW, PP, 0>), dim3(total), dim3(64 * WW), L::FLOATS * sizeof(float), st, a); \
    }
#define XW(HH, DD, OO, WW)                          \
    if (waves == WW) {                              \
        if (kind == 1) {                            \
            if (a.has_merge == 2) {                 \
                XK(HH, DD, OO, 1, WW, 2)            \
            } else if (a.has_merge) {               \
                XK(HH, DD, OO, 1, WW, 1)            \
            } else {                                \
                XK(HH, DD, OO, 1, WW, 0)            \
            }                                       \
        } else {                                    \
            if (a.has_merge == 2) {                 \
                XK(HH, DD, OO, 0, WW, 2)            \
            } else if (a.has_merge) {               \
                XK(HH, DD, OO, 0, WW, 1)            \
            } else {                                \
                XK(HH, DD, OO, 0, WW, 0)            \
            }                                       \
        }                                           \
        return;                                     \
    }
#define XR(HH, DD, OO)                       \
    if constexpr (OO <= 20) {                \
        if (h == HH && d == DD && O == OO) { \
            XW(HH, DD, OO, 4)                \
            XW(HH, DD, OO, 8)                \
        }                                    \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XW
#undef XK
}

// ... and the same launch for n problems (blockIdx.y); shape = one problem's record (all equal: plan.hip checked)
void launch_iter_ahead_batch(const BatchRecord& sr, const IterAheadArgs* args_dev, const BatchBases& bases, int n, hipStream_t st) {
    const int h = sr.h, d = sr.d, O = sr.O, kind = sr.model_kind, waves = sr.rw;
    const IterAheadArgs& a = sr.ia;
#define XK(HH, DD, OO, KK, WW, PP)                                                                                         \
    {                                                                                                                      \
        using L = AheadLds<HH, DD, OO, KK, WW>;                                                                            \
        const int n_noise = a.z.n > 0 ? (a.z.n + L::TPW - 1) / L::TPW : 0;                                                 \
        const dim3 grid(a.n_roll + n_noise + (a.s.n_shift > 0 ? 1 : 0), n);                                                \
        if (a.r.arith == 1)                                                                                                \
            hipLaunchKernelGGL((iter_ahead_batch_kernel<HH, DD, OO, KK, WW, PP, 1>), grid, dim3(64 * WW), L::FLOATS * sizeof(float), st, args_dev, bases); \
        else                                                                                                               \
            hipLaunchKernelGGL((iter_ahead_batch_kernel<HH, DD, OO, KK, WW, PP, 0>), grid, dim3(64 * WW), L::FLOATS * sizeof(float), st, args_dev, bases); \
    }
#define XW(HH, DD, OO, WW)                          \
    if (waves == WW) {                              \
        if (kind == 1) {                            \
            if (a.has_merge) {                      \
                XK(HH, DD, OO, 1, WW, 1)            \
            } else {                                \
                XK(HH, DD, OO, 1, WW, 0)            \
            }                                       \
        } else {                                    \
            if (a.has_merge) {                      \
                XK(HH, DD, OO, 0, WW, 1)            \
            } else {                                \
                XK(HH, DD, OO, 0, WW, 0)            \
            }                                       \
        }                                           \
        return;                                     \
    }
#define XR(HH, DD, OO)                       \
    if constexpr (OO <= 20) {                \
        if (h == HH && d == DD && O == OO) { \
            XW(HH, DD, OO, 4)                \
            XW(HH, DD, OO, 8)                \
        }                                    \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XW
#undef XK
}

}  // namespace icem
