// icem_fused.h -- interface between the C-ABI translation unit and the fused f32 kernels
// (icem_fused.hip).  Internal; not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icem {

constexpr int FUSED_WG = 256;      // threads per workgroup (4 wavefronts)
constexpr int FUSED_MAX_GRID = 1024;  // one sorted candidate list per workgroup; merged by 1024 threads

// One CEM iteration's data-parallel part for this rank's shard (f32):
//   tile of TPW trajectories per workgroup pass:  sample (Philox -> Box-Muller -> folded inverse
//   real DFT -> affine -> clip) into an LDS tile  ->  tile to HBM as coalesced stores  ->  rollout
//   + cost with L lanes per trajectory straight out of LDS  ->  wave-level bitonic merge into the
//   workgroup's running sorted top-K.  Workgroups stride over tiles and emit K candidates each.
struct FusedArgs {
    int n;        // sampled trajectories of this rank (rows [0, n) of actions)
    int n_extra;  // pre-filled rows [n, n + n_extra) (shifted elites): rolled out, not sampled
    int n_cand;   // rows [0, n_cand) are top-k candidates (n, or n + n_extra on rank 0)
    int h, d, F, o;
    int tpw;          // trajectories per tile
    int tile_stride;  // floats per trajectory in the LDS tile (h*d padded)
    int K;
    int cost_mode;
    int row0_mean;
    long long first_index;
    const float* W;  // [h, HMAX]
    const float* mean;
    const float* std;
    const float* low;
    const float* high;
    uint32_t seed_lo, seed_hi, off_lo, off_hi;
    const float* A;  // [O, O] padded
    const float* B;  // [d, O] padded
    const float* obs0;
    float ctrl_w, lin_w, flip_pen, flip_th;
    int lin_idx, flip_idx;
    float* actions;  // [n + n_extra, h, d]
    float* costs;    // [n + n_extra]
    float* part_c;   // [grid, K] sorted candidates of each workgroup
    int* part_i;
    long long* dbg;  // optional [grid, 8] phase cycle stamps (nullptr in production)
};

// Returns 0 when a kernel for (O, d, model kind, rounds) exists and was launched, 1 when the
// combination is not compiled (caller falls back to the unfused kernels).
int launch_fused_iter(const FusedArgs& a, int O, int kind, int rounds, int grid, hipStream_t st);
bool fused_supported(int O, int d, int h, int K);
int fused_tile_traj(int h, int d);
int fused_tile_stride(int h, int d);

// world == 1: global sorted top-K straight from the workgroups' candidate lists (+ kept elites),
// gather of the elite rows from the pool, refit, and the last-iteration epilogue.
struct MergeSingleArgs {
    int n_lists;   // candidate lists (one per fused workgroup), each K long and sorted
    int n_keep;    // kept elites appended as candidates (icem.py:143-145), gidx = n_pool + e
    int n_pool;    // sampled + shifted rows in `actions` this iteration (local == global, world 1)
    int n_global;  // N_it: index offset of shifted (it == 0) or kept (it > 0) elites
    int K, h, d, last;
    float alpha, init_std;
    const float* part_c;
    const int* part_i;
    const float* actions;
    const float* elites_cur;
    const float* elites_cost_cur;
    float* elites_next;
    float* elites_cost_next;
    float* mean;
    float* std;
    const float* low;
    const float* high;
    float* executed;
    float* best_cost;
};
void launch_merge_single(const MergeSingleArgs& a, hipStream_t st);

}  // namespace icem
