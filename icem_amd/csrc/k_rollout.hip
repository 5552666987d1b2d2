// k_rollout.hip -- K2 + K3: rollout16_kernel<H, D, O, KIND, WAVES>: one wavefront per 16 trajectories on
// v_mfma_f32_16x16x4_f32 (Tile16, fused_dev.h), cost on the VALU, one sorted candidate list per workgroup.
#include "fused_dev.h"

namespace icem {

namespace {

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

}  // namespace icem
