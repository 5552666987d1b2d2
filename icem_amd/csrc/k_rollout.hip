// k_rollout.hip -- K2 + K3: rollout16_kernel<H, D, O, KIND, WAVES>: one wavefront per 16 trajectories on
// v_mfma_f32_16x16x4_f32 (Tile16, fused_dev.h), cost on the VALU, one sorted candidate list per workgroup.
#include "fused_dev.h"

namespace icem {

namespace {

template <int H, int D, int O, int KIND, int WAVES, int ARITH>
__global__ __launch_bounds__(64 * WAVES) void rollout16_kernel(FastRolloutArgs a) {
    using Stream = Stream16<H, D, O, KIND, ARITH>;
    using Tile = typename Stream::Tile;
    __shared__ __attribute__((aligned(16))) float stage[WAVES][Stream::STG];
    __shared__ unsigned long long wg_keys[2][WAVES][32];
    __shared__ float obs_stage[32];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // model operands and start observation in flight together: one wait at the barrier
    const float obs_reg = a.obs0[(threadIdx.x < 32 && (int)threadIdx.x < a.o) ? threadIdx.x : 0];
    Tile tile;
    tile.load(a, lane);
    Stream stream;
    stream.init(tile, stage[wave], lane);
    // tile t of the launch belongs to wave t / gridDim.x of workgroup t % gridDim.x: a short launch thins every CU
    const int tiles = (a.n_rows + 15) / 16;
    const int tile0 = wave * gridDim.x + blockIdx.x;
    // (requesting the first tile's actions HERE, next to the model operands, so that the two cold round trips overlap,
    //  was measured and lost: 26.5 instead of 24.5 us per launch at N = 65 536 -- 4 096 waves asking HBM for their first
    //  chunk in the same microsecond queue up behind each other, and the model loads behind them)
    typename Stream::Vec pre[Stream::NLD];
    if (threadIdx.x < 32) obs_stage[threadIdx.x] = (int)threadIdx.x < a.o ? obs_reg : 0.f;
    __syncthreads();
    tile.load_obs(obs_stage);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    for (int tile_id = tile0; tile_id < tiles; tile_id += WAVES * gridDim.x) {
        stream.first_loads(a.actions, a.n_rows, tile_id, pre);
        run_key = stream.run(tile, a, tile_id, lane, run_key, first, pre);
        first = false;
    }
    if (a.K > 0) wg_merge_emit<WAVES>(wg_keys, run_key, a.K, lane, wave, a);
}

}  // namespace

bool fast_rollout_supported(int h, int d, int O, int K) {
    if (K > 32) return false;
#define X(HH, DD, OO) \
    if (h == HH && d == DD && O == OO) return true;
    ICEM_FAST_SHAPES(X)
#undef X
    return false;
}

int rollout_lists(int h, int d, int O, int n_rows) {
    int g, w;
    r16_shape(n_rows, &g, &w);
    return g;
}

void launch_rollout16(const FastRolloutArgs& a, int h, int d, int O, int kind, hipStream_t st) {
    int grid, waves;
    r16_shape(a.n_rows, &grid, &waves);
    // two output tiles (O > 20): a 16-wave workgroup's 128 registers spill ~100 of the two-tile step's; at most 8 waves per
    // workgroup (256 registers, no spills), each taking its tiles one after the other (EXPERIMENTS.md R4.6)
    if (O > 20 && waves > 8) waves = 8;
#define XA(HH, DD, OO, KK, WW)                                                                                  \
    {                                                                                                           \
        if constexpr (OO <= 20) {                                                                               \
            if (a.arith == 1) {                                                                                 \
                hipLaunchKernelGGL((rollout16_kernel<HH, DD, OO, KK, WW, 1>), dim3(grid), dim3(64 * WW), 0, st, a); \
                return;                                                                                         \
            }                                                                                                   \
        }                                                                                                       \
        hipLaunchKernelGGL((rollout16_kernel<HH, DD, OO, KK, WW, 0>), dim3(grid), dim3(64 * WW), 0, st, a);     \
        return;                                                                                                 \
    }
#define XW(HH, DD, OO, WW)                     \
    if constexpr (OO <= 20 || WW <= 8) {       \
        if (waves == WW) {                     \
            if (kind == 1) XA(HH, DD, OO, 1, WW) \
            else XA(HH, DD, OO, 0, WW)         \
        }                                      \
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
#undef XA
}

}  // namespace icem
