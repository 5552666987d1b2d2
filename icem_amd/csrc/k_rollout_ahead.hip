// k_rollout_ahead.hip -- K2 + K3 of the noise-ahead pipeline (large populations, world == 1): rollout16_ahead_kernel =
// rollout16_kernel (one wavefront per 16 trajectories on v_mfma_f32_16x16x4_f32, Tile16 / Stream16 of fused_dev.h) whose
// pool holds RAW colored noise drawn ahead of time (noise_rows_kernel, k_sample.hip, on a second stream while the
// previous rollout ran).  What the sampler used to do with the distribution happens here:
//   * PM: the PREVIOUS iteration's K3 + K4 (top-K of its candidate lists, elite gather, refit; icem.py:194-211) in the
//     prologue -- all wavefronts share the selection (merge_select_split: nothing hides it here), then all threads gather +
//     refit (refit.h: every workgroup gets the same bits), workgroup 0 publishes;
//   * every vector a wave loads becomes clip(y * std + mean) (icem.py:79) between the prefetch registers and its LDS
//     staging buffer and is written back in place (Stream16::run_xf): after the launch the pool holds the actions.
// One launch per CEM iteration on the critical path instead of two; same operations in the same order as
// sample_folded(_merge)_kernel + rollout16_kernel, so the same bits in every buffer (tests: the at-size loops, the
// plan_step == split-API checks).
#include "fused_dev.h"

namespace icem {

namespace {

// (at most 128 registers whatever the workgroup size: the waves share their SIMDs with the noise kernel's)
template <int H, int D, int O, int KIND, int WAVES, bool PM>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) void rollout16_ahead_kernel(RolloutAheadArgs args) {
    using Tile = Tile16<H, D, O, KIND>;
    using Stream = Stream16<H, D, O, KIND>;
    constexpr int HD = H * D, NTT = 64 * WAVES, KREG = 12;
    __shared__ __attribute__((aligned(16))) float stage[WAVES][Stream::STG];
    __shared__ unsigned long long wg_keys[2][WAVES][32];
    __shared__ float obs_stage[32];
    __shared__ __attribute__((aligned(16))) float dist[2 * HD];  // mean | std this iteration samples from
    __shared__ unsigned long long sel[PM ? 64 : 1];
    __shared__ unsigned long long cand[PM ? WAVES : 1][PM ? 64 : 1];  // compaction scratch, one per wave
    __shared__ unsigned long long wsel[PM ? WAVES * 16 : 1];         // every wave's K best (merge_select_split)
    __shared__ int slot[PM ? 64 : 1];
    const FastRolloutArgs& a = args.r;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // model operands, start observation and this wave's first noise vectors in flight together, in front of the merge
    const float obs_reg = a.obs0[(tid < 32 && tid < a.o) ? tid : 0];
    Tile tile;
    tile.load(a, lane);
    Stream stream;
    stream.init(tile, stage[wave], lane);
    const int tiles = (a.n_rows + 15) / 16;
    const int tile0 = wave * gridDim.x + blockIdx.x;
    typename Stream::Vec pre[Stream::NLD];
    if (tile0 < tiles) stream.first_loads(args.pool, a.n_rows, tile0, pre);
    if constexpr (PM) {
        // nothing to hide the selection behind here: all waves share it (one cold round trip instead of a dozen)
        merge_select_split_stage1<KREG>(args.m, lane, wave, WAVES, cand[wave], wsel);
        __syncthreads();
        if (wave == 0) merge_select_split_stage2(args.m, lane, WAVES, wsel, cand[0], sel);
    } else {
        for (int e = tid; e < HD; e += NTT) {
            dist[e] = args.mean[e];
            dist[HD + e] = args.std[e];
        }
    }
    if (tid < 32) obs_stage[tid] = tid < a.o ? obs_reg : 0.f;
    __syncthreads();
    if constexpr (PM) {
        const MergeSingleArgs& m = args.m;
        const float* rows[KREG];
        merge_rows<KREG, false>(m, sel, slot, rows);
        for (int e = tid; e < HD; e += NTT) {
            float xs[KREG];
#pragma unroll
            for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
            float nm, ns;
            refit_element_regs<float, KREG>(m.K, m.alpha, m.mean[e], m.std[e], xs, nm, ns);
            dist[e] = nm;
            dist[HD + e] = ns;
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
    }
    tile.load_obs(obs_stage);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    // tile t of the launch belongs to wave t / gridDim.x of workgroup t % gridDim.x (as rollout16_kernel)
    for (int tile_id = tile0; tile_id < tiles; tile_id += WAVES * gridDim.x) {
        if (!first) stream.first_loads(args.pool, a.n_rows, tile_id, pre);
        run_key = stream.run_xf(tile, a, args.pool, args.n_xf, args.row0_mean != 0, args.store_back != 0, dist, args.lo, args.hi, tile_id, lane, run_key, first, pre);
        first = false;
    }
    if (a.K > 0) wg_merge_emit<WAVES>(wg_keys, run_key, a.K, lane, wave, a);
}

}  // namespace

// Launch shape: rollout16's (one 16-trajectory tile per wave while they fit, at most FAST_MAX_LISTS workgroups), but at
// most AHEAD_MAX_WAVES wavefronts per workgroup -- two per SIMD, each taking its tiles one after the other: the rollout
// must leave half of every SIMD's registers to the noise kernel that runs beside it (a 16-wave workgroup owns its CU, and
// noise workgroups back-filling every freed slot then starve the rollout's: measured 751 us per MPC step at N = 65 536
// against 214 for the sampler + rollout pair).
static int ahead_max_waves() {
    static const int w = [] { const char* e = getenv("ICEM_AHEAD_MAXW"); const int v = e ? atoi(e) : 8; return v >= 16 ? 16 : (v >= 8 ? 8 : 4); }();
    return w;
}
static void ahead_shape(int n_rows, int* grid, int* waves) {
    r16_shape(n_rows, grid, waves);
    if (*waves > ahead_max_waves()) *waves = ahead_max_waves();
}

// populations whose rollout launch has at least 4 waves per workgroup (more than 512 tiles): below that the
// single-launch kernel of k_iter_small.hip is the shorter chain
bool rollout_ahead_ok(int h, int d, int O, int K, int n_rows) {
    int grid, waves;
    r16_shape(n_rows, &grid, &waves);
    return K + 1 <= 12 && waves >= 4 && fast_rollout_supported(h, d, O, K) && fast_sample_supported(h, d);
}

void launch_rollout_ahead(const RolloutAheadArgs& a, int h, int d, int O, int kind, hipStream_t st) {
    int grid, waves;
    ahead_shape(a.r.n_rows, &grid, &waves);
#define XK(HH, DD, OO, KK, WW, PP) \
    hipLaunchKernelGGL((rollout16_ahead_kernel<HH, DD, OO, KK, WW, PP>), dim3(grid), dim3(64 * WW), 0, st, a);
#define XW(HH, DD, OO, WW)                          \
    if (waves == WW) {                              \
        if (kind == 1) {                            \
            if (a.has_merge) {                      \
                XK(HH, DD, OO, 1, WW, true)         \
            } else {                                \
                XK(HH, DD, OO, 1, WW, false)        \
            }                                       \
        } else {                                    \
            if (a.has_merge) {                      \
                XK(HH, DD, OO, 0, WW, true)         \
            } else {                                \
                XK(HH, DD, OO, 0, WW, false)        \
            }                                       \
        }                                           \
        return;                                     \
    }
#define XR(HH, DD, OO)                   \
    if (h == HH && d == DD && O == OO) { \
        XW(HH, DD, OO, 4)                \
        XW(HH, DD, OO, 8)                \
        XW(HH, DD, OO, 16)               \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XW
#undef XK
}

}  // namespace icem
