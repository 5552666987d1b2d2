// Fused rollout of the declared recurrent state-space model (BASELINE configs[4]: learned dynamics, N=1024, h=12) on
// the bf16 matrix cores.  Architecture (icem_amd/models.py::declared_rssm): planner observation [h (200) | z (30)],
//   x  = relu(W1 [z, a] + b1)                200
//   h' = GRUCell(x, h)                       200      (torch gate order r, u, n)
//   z' = W5 relu(W4 h' + b4) + b5            30
//   reward(h, z) = W8 relu(W7 relu(W6 [h, z] + b6) + b7) + b8,   cost of a step = -reward of the state it starts from
// Output widths are padded to multiples of 16 (200 -> 208, 30 -> 32), contraction widths to multiples of 32 (200 -> 224,
// 30 -> 32, 6 -> 32); the packed parameter buffer holds, per layer, the weight as v_mfma_f32_16x16x32_bf16 A-operand
// blocks [out block][k block][lane 64][8 bf16] followed by the f32 bias.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace icem {
namespace rssm {
constexpr int DET = 200, STOCH = 30, HID = 200, ACT = 6;
constexpr int DETB = 13, STB = 2, HIDB = 13;             // 16-wide OUTPUT blocks (200 -> 208, 30 -> 32)
constexpr int DETK = 7, STK = 1, ACTK = 1, HIDK = 7;     // 32-wide K blocks (200 -> 224, 30 -> 32, 6 -> 32)
constexpr int K1K = STK + ACTK;                          // [z | a]
constexpr int K6K = DETK + STK;                          // [h | z]
constexpr int BLK = 64 * 8;                              // bf16 elements of one 16 x 32 A-operand block
// element offsets (in bf16 units; biases are f32 = 2 units each) of the packed parameter buffer
constexpr size_t W1 = 0, B1 = W1 + (size_t)HIDB * K1K * BLK;
constexpr size_t WGI = B1 + 2 * 16 * HIDB, BGI = WGI + (size_t)3 * DETB * HIDK * BLK;
constexpr size_t WGH = BGI + 2 * 16 * 3 * DETB, BGH = WGH + (size_t)3 * DETB * DETK * BLK;
constexpr size_t W4 = BGH + 2 * 16 * 3 * DETB, B4 = W4 + (size_t)HIDB * DETK * BLK;
constexpr size_t W5 = B4 + 2 * 16 * HIDB, B5 = W5 + (size_t)STB * HIDK * BLK;
constexpr size_t W6 = B5 + 2 * 16 * STB, B6 = W6 + (size_t)HIDB * K6K * BLK;
constexpr size_t W7 = B6 + 2 * 16 * HIDB, B7 = W7 + (size_t)HIDB * HIDK * BLK;
constexpr size_t W8 = B7 + 2 * 16 * HIDB, B8 = W8 + (size_t)1 * HIDK * BLK;
constexpr size_t TOTAL = B8 + 2 * 16;  // bf16 units
constexpr int SPLIT_TILE_LIMIT = 8192;  // what the staging bookkeeping is sized for
constexpr int SPLIT_TT1_TILES = 256;    // up to here one tile per recurrence workgroup, two beyond
constexpr int SPLIT_MAX_TILES = 4096;  // populations up to 65 536: recurrence and reward head as separate workgroups (icem_rssm_split.hip)
}  // namespace rssm

// costs[i] = reduce_t -reward(state_t) along the rollout of actions[i] from obs0 (cost_mode: 0 sum, 1 best, 2 final)
hipError_t launch_rssm_rollout(int n, int horizon, int cost_mode, const unsigned short* params, const float* obs0,
                               const float* actions, float* costs, hipStream_t st);
// n <= 16 * SPLIT_MAX_TILES: one launch of recurrence workgroups + reward-head workgroups (ICEM_RSSM_SPLIT=0 turns it off)
bool rssm_split_ok(int n, int horizon);
void rssm_split_trim();                     // frees the split launch's per-(device, stream) staging areas
void rssm_set_stamps(long long* dev_ptr);   // development aid: 16 int64 of wall_clock64 phase stamps of tile 0 (NULL = off)
hipError_t launch_rssm_split(int n, int horizon, int cost_mode, const unsigned short* params, const float* obs0,
                             const float* actions, float* costs, hipStream_t st);
}  // namespace icem
