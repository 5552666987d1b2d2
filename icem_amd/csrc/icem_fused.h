// icem_fused.h -- interface between the host-side translation units (plan.hip, abi.hip) and the f32 throughput
// kernels (k_sample.hip, k_rollout.hip, k_rollout_ahead.hip, k_rollout_wide.hip, k_iter_small.hip, k_merge.hip):
// argument blocks and launchers.  Internal; not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "cost_args.h"
#include "options.h"

namespace icem {

constexpr int FAST_MAX_LISTS = 256;  // candidate lists (one per rollout workgroup) the merge accepts

// In-library elite exchange (exchange.hip): what a merge needs to wait for the ranks' records of one exchange.
constexpr int XCHG_MAX_WORLD = 16;
struct XchgWait {
    const unsigned* flags = nullptr;  // [world] sequence flags of the exchange's parity in THIS rank's block; nullptr: no wait
    unsigned* status = nullptr;       // set to 1 when a wait times out
    unsigned seq = 0;                 // the value every flag must reach
    int world = 0;
    unsigned max_polls = 0;           // bound of the wait (polls of ~0.5 us); then status = 1 and the wait gives up
    const void* records = nullptr;    // [world * K] gathered records of that parity (handle dtype)
};
// ... and what the producer needs to push its K records into every rank's block from inside the pack kernel
struct XchgPush {
    unsigned char* const* peers = nullptr;  // [world] device pointers of the ranks' blocks; nullptr: no push
    size_t rec_byte_off = 0;                // byte offset of this rank's slot (parity applied) inside a block
    int world = 0;
    int flag_idx = 0;                       // flag word of (parity, this rank)
    unsigned seq = 0;
};

// K1 fast: colored-noise sampling with the inverse real DFT folded on its symmetry (f32, Philox).
struct FastSampleArgs {
    int n, h, d;
    long long first_index;
    const float* W;  // [h, 32]
    const float* mean;
    const float* std;
    const float* low;
    const float* high;
    uint32_t seed_lo, seed_hi, off_lo, off_hi;
    int row0_mean;
    float* out;  // [n (+ n_shift), h, d]
    // optional shifted elites (icem.py:91-104), handled by one extra workgroup of the same launch:
    // rows [n, n + n_shift) <- elites_src[e, 1:, :] ++ one freshly sampled last action (stream off2)
    int n_shift;
    const float* elites_src;  // [>= n_shift, h, d]
    uint32_t off2_lo, off2_hi;
    int white;  // noise_beta <= 0 (icem.py:77): normal t of a row is its sample at step t, no synthesis
    // single-launch kernel, iteration 0 only: the raw colored noise of this call -- rows [0, n) and, behind them, the
    // n_shift shifted elites' rows of stream off2 -- was drawn ahead (beside the previous MPC step's last merge,
    // merge_noise_kernel) and only has to be mapped; nullptr: draw it here
    const float* raw_src = nullptr;
};
bool fast_sample_supported(int h, int d);
void launch_sample_folded(const FastSampleArgs& a, int rounds, hipStream_t st);

// K2+K3 fast: rollout on the matrix pipe (v_mfma_f32_16x16x4, exact f32), cost on the VALU, and a
// sorted top-K per workgroup; one wavefront per 16 trajectories.
struct FastRolloutArgs {
    int n_rows;   // trajectories to roll out (rows of `actions`)
    int n_cand;   // rows [0, n_cand) enter the top-K
    int K;        // 0 = no candidate output
    int o;        // true observation width (<= O)
    int cost_mode;
    const float* Mp;    // [O + D + 1, 4*ceil(O/4)] permuted, zero padded model [A ; B], then one zero row
    const int* perm;    // [32] permuted column k holds observation entry perm[k]; 31 (a zero slot) for k >= o
    const float* obs0;
    float ctrl_w, lin_w, flip_pen, flip_th;
    int flip_col;       // column holding obs[flip_idx] after the permutation, -1 = no flip term
    const float* actions;
    float* costs;
    float* part_c;  // [grid, K] sorted candidate costs / row indices of every workgroup ...
    int* part_i;
    unsigned long long* part_k;  // ... or, if not null, their packed keys, key r of workgroup w at [r * grid + w]
    int list_wgs = 0;   // > 0: only the first list_wgs workgroups emit a list (sample_rollout_lists' tail rows)
    long long* dbg;  // development: [waves, 8] cycle stamps, nullptr in production
    // which arithmetic the model step of the 16-trajectory tile runs in (icem_set_tile_arith; fused_dev.h):
    // 0 = v_mfma_f32_16x16x4_f32, bitwise an fmaf chain (Tile16; Tile4 is its VALU twin); 1 = two fp16 planes of every f32
    // operand, three products per multiply-add on v_mfma_f32_16x16x32_f16 (Tile16H; one-tile observation widths only)
    int arith = 0;
    float act_mag = 1.f;   // arith 1: magnitude of the action bounds, max(|low|, |high|) -- with |obs0| it fixes the launch's scale
    float m_scale = 1.f, b_scale = 1.f;   // arith 1: powers of two that put the largest |entry| of A / of B into [64, 128)
    unsigned* nonfinite = nullptr;        // counts the trajectories whose cost came out NaN (icem_nonfinite_costs); never NULL in a launch
};
bool fast_rollout_supported(int h, int d, int O, int K);
void launch_rollout16(const FastRolloutArgs& a, int h, int d, int O, int kind, hipStream_t st);
// workgroups (= candidate lists) the rollout of n_rows trajectories is launched with
int rollout_lists(int h, int d, int O, int n_rows);

// K2+K3 for the narrow shapes beside HalfCheetah that the reference ships (Door, Relocate, FetchPickAndPlace: k_rollout_hn.hip):
// TileHN -- up to three output tiles on the 16-bit matrix cores (the fp16-plane arithmetic of Tile16H), icem_cost_terms
// evaluated across the four lanes of a trajectory.  A [o, lda], B [d, ldb]: the row-major f32 model; cs: device copy of the
// cost terms (nullptr: none).  r: n_rows / n_cand / K / o / cost_mode / obs0 / actions / costs / part_* / the cost spec's
// weights / act_mag, m_scale, b_scale.
template <typename T> struct CostArgs;
bool hn_rollout_supported(int h, int d, int o, int K);
int hn_rollout_lists(int n_rows);
int hn_tail_rows(int n_rows, int n_tail);   // trailing shifted-elite rows scored through the cost array (0: none)
// cs: terms sorted into the program prog = (N32, N4, NP) and padded with null terms (kind -1): hn_cost_program says whether a
// list of n32 long slices (5 .. 32 entries), n4 short ones and np point terms has a compiled program, and which
bool hn_cost_program(int n32, int n4, int np, int* prog);
void launch_rollout_hn(const FastRolloutArgs& r, int h, int d, int o, int kind, const float* A, int lda, const float* B, int ldb,
                       int lin_idx, int flip_idx, const CostArgs<float>* cs, const int* prog, hipStream_t st);

// K2+K3 for wide observations (32 < o <= 384; k_rollout_wide.hip): the model step as an f32 matrix-pipe GEMM per
// 16-trajectory tile, contraction vectors in LDS; same candidate-list outputs as the kernels above.
struct WideRolloutArgs {
    int n_rows, n_cand, K;
    int o, d, h;
    int kb, xs;           // contraction blocks of 4, LDS row stride (wide_kb / wide_xs)
    int planes;           // rollout_wide_split_kernel: 3 = bf16 planes (six products per multiply-add), 2 = fp16 planes (three)
    float minv;           // ... fp16 planes: 1 / the model's power-of-two scale (pack_wide_model_split)
    const float* ksc;     // ... the contraction entries' powers of two, [32 kb] (device memory)
    const float* csc;     // ... and the output columns', [16 x 8 x NCT]
    float sbound;         // ... the largest scaled state entry of a tanh model (max over the observation entries of their power of two)
    int cost_mode;
    int lin_idx, flip_idx;
    float ctrl_w, lin_w, flip_pen, flip_th;
    const CostArgs<float>* cs;   // icem_cost_terms on: the same + the terms, in device memory (NULL: off)
    long long* dbg;       // development: 8 wall_clock64 phase stamps of one wave (rollout_wide_split_kernel; tools/dbg/split_stamps.py), nullptr in production
    const float* Mp;      // pack_wide_model / pack_wide_model_split
    const float* obs0;
    const float* actions;
    float* costs;
    float* part_c;
    int* part_i;
    unsigned long long* part_k;
};
bool wide_rollout_supported(int o, int d, int K);
bool gemm_rollout_supported(int o, int d, int K);   // the same kernels at ANY observation width 1..384 (narrow models the tile kernels do not serve)
int wide_rollout_lists(int n_rows);
int wide_kb(int o, int d);
int wide_xs(int o, int d);
void pack_wide_model(int o, int d, const double* A, const double* B, std::vector<float>& Mp);
void launch_rollout_wide(const WideRolloutArgs& a, int kind, hipStream_t st);
// ... the same on the bf16 matrix cores (k_rollout_wide_split.hip): every f32 operand as three bf16 planes, six products
// per MAC, up to 80 trajectories per workgroup; a.Mp = pack_wide_model_split's planes, a.kb / a.xs = wide_split_kb / _xs
int wide_split_kb(int o, int d);
int wide_split_xs(int o, int d);
int wide_split_lists(int n_rows);
bool wide_split_fits(int o, int d);   // the workgroup's rows + planes in 160 KB of LDS (o + d <= 416); else the exact-f32 kernel
int wide_model_imbalance_log2(int o, int d, const double* A, const double* B);   // ICEM_WIDE_AUTO's criterion (k_rollout_wide_split.hip)
void pack_wide_model_split(int o, int d, const double* A, const double* B, int planes, std::vector<unsigned short>& Mb, float* minv,
                           std::vector<float>* ksc, std::vector<float>* csc, float* sbound);
void launch_rollout_wide_split(const WideRolloutArgs& a, int kind, hipStream_t st);
// rows [row0, row0 + n_tail) of the same pool one workgroup each, from the row-major f32 model (A [o, o], B [d, o]); costs only
void launch_rollout_rows_wide(const WideRolloutArgs& a, int row0, int n_tail, const float* A, const float* B, int kind,
                              hipStream_t st);

// world == 1: global sorted top-K straight from the waves' candidate lists (+ kept elites), gather of
// the elite rows from the pool, refit, and the last-iteration epilogue.
struct MergeSingleArgs {
    int n_lists;   // candidate lists, each K long and sorted
    int n_keep;    // kept elites appended as candidates (icem.py:143-145), gidx = n_global + e
    int keep_base = -1;  // >= 0: index of kept candidate 0 in its key instead of n_global (pack of a sharded rank: n_loc)
    int n_pool;    // sampled + shifted rows in `actions` this iteration (local == global, world 1)
    int n_global;  // N_it: index offset of shifted (it == 0) or kept (it > 0) elites
    int K, h, d, last;
    float alpha, init_std;
    const unsigned long long* part_k;  // [K, n_lists] packed candidate keys (FastRolloutArgs::part_k)
    // sharded runs (merge prologues only): the candidates are the all-gathered records {cost, gidx, actions[h*d]}
    // instead of lists + pool (part_k / actions / n_lists / n_pool unused)
    const float* records;  // [n_rec, 2 + h*d] or nullptr
    int n_rec;
    XchgWait xw;           // records arrive through the in-library exchange: wait for them first (flags == nullptr: no)
    const float* actions;
    const float* elites_cur;
    const float* elites_cost_cur;
    float* elites_next;
    float* elites_cost_next;
    const float* mean;  // distribution before the refit (momentum term)
    const float* std;
    float* mean_out;    // ... and after (may alias mean / std)
    float* std_out;
    const float* low;
    const float* high;
    float* executed;
    float* best_cost;
    long long* dbg;  // development: 8 wall_clock64 stamps (100 MHz) of thread 0, nullptr in production
    // noise-ahead pipeline, non-last iterations (iter_ahead_kernel's prologue only): pool rows [0, n_raw) still hold the
    // RAW colored noise y -- the launch that rolled them out did not write the actions back -- and an elite row among
    // them is clip(y * std + mean, xf_lo, xf_hi) with THIS merge's input distribution (mean / std above: what the
    // iteration sampled from).  0: the pool holds actions.
    int n_raw = 0;
    float xf_lo = 0.f, xf_hi = 0.f;
};
void launch_merge_single(const MergeSingleArgs& a, hipStream_t st);
// ... with noise workgroups beside it (noise-ahead pipeline: z.n rows of raw colored noise -> z.out; see merge_noise_kernel)
bool merge_noise_ok(const MergeSingleArgs& a, int rounds);
// z2 (z2.n > 0): a second, small sampling call in one more workgroup (the next step's shifted elites' noise)
void launch_merge_noise(const MergeSingleArgs& a, const FastSampleArgs& z, const FastSampleArgs& z2, hipStream_t st);
// icem_update_distribution for small f32 pools (topk_small_ok(n + n_keep, K)): top-K over [costs | keep_costs], gather from
// [pool | keep_actions], refit of mean / std in place -- one launch
struct UpdateSmallArgs {
    const float* costs;         // [n]
    const float* pool;          // [n, hd]
    const float* keep_costs;    // [n_keep] or nullptr
    const float* keep_actions;  // [n_keep, hd]
    int n, n_keep, K, hd;
    float alpha;
    float* mean;                // [hd], in place
    float* std;
    float* elites_out;          // [K, hd] (must not alias keep_actions)
    float* elite_costs_out;     // [K]
    int* idx_out;               // [K] indices into [pool | kept elites]
};
void launch_update_small(const UpdateSmallArgs& a, hipStream_t st);
// sorted top-K of a small f32 cost array in one launch (icem_topk_sorted's fast path)
bool topk_small_ok(int n, int K);
void launch_topk_small(const float* costs, int n, int K, float* out_c, int* out_i, hipStream_t st);
// sharded runs: the rank's K best of its candidate lists (part_k / actions / n_lists / n_keep = 0 of `a`) -> records [K, 2 + h*d]
// px.peers != nullptr (allowed when pack_can_push): the records also go into every rank's exchange block, followed by the
// flags -- pack and push in one launch
bool pack_can_push(int K, int h, int d);
void launch_pack_records(const MergeSingleArgs& a, int n_loc, int shard_lo, float* records, hipStream_t st,
                         const XchgPush& px = XchgPush());
// the same + the records merge `m` that waits for this pack's records (the step's last iteration), one launch
void launch_pack_merge(const MergeSingleArgs& pk, int n_loc, int shard_lo, float* records, const XchgPush& px, const MergeSingleArgs& m,
                       hipStream_t st);

// Sharded runs, "riding pack": the PREVIOUS iteration's record pack + push as one extra workgroup (index 0) of this
// iteration's launch, running while the other workgroups draw their noise; their merge prologue then waits for the
// exchange's flags (this rank's own among them) as always.  Reads the previous launch's lists and pool, which this
// launch overwrites only behind those flags.
struct PackPrev {
    const unsigned long long* part_k = nullptr;  // previous launch's candidate lists; nullptr: no pack in this launch
    const float* actions = nullptr;              // ... and pool
    int n_lists = 0, n_pool = 0, n_global = 0, K = 0;
    int n_loc = 0, shard_lo = 0;
    int n_keep = 0;                              // shifted-elite rows behind the lists (rank 0, iteration 0), scored through
    const float* keep_costs = nullptr;           // ... their costs (the cost array from row n_loc on)
    float* records = nullptr;                    // this rank's [K, 2 + h*d] records
    XchgPush px;
    // sample_folded_merge_kernel only ("published merge"): workgroup 0 also runs the launch's one records merge and
    // publishes mean | std here (written through), then stores pub_seq to pub_flag; the other workgroups wait for that
    float* pub = nullptr;                        // [2 * h*d]; nullptr: every workgroup merges for itself
    unsigned* pub_flag = nullptr;
    unsigned pub_seq = 0;
};
// may the merge-prologue launch of an iteration with n_rows local rows carry the previous iteration's pack? (the
// single-launch kernel or the sampler of the two-kernel path; the records must fit the workgroup's tile)
bool sample_rollout_pack_ok(int h, int d, int O, int rounds, int n_rows, int K);
bool sample_folded_pack_ok(int h, int d, int rounds, int K);

// ---- noise-ahead pipeline (large populations, world == 1; plan.hip::plan_step_ahead) -----------------------
// The colored noise of an iteration does not depend on the distribution (icem.py:73-79: powerlaw_psd_gaussian first,
// `* std + mean` after), so it is drawn AHEAD: every iteration is ONE launch (iter_ahead_kernel, k_rollout_ahead.hip)
// whose workgroups split into a rollout role (previous iteration's merge in the prologue, affine map + clip applied to
// every vector it loads from the raw-noise pool and written back in place), a noise role (the NEXT sampling call's raw
// y [n, h, d] into the next pool) and, at iteration 0, a shifted-elites role.  Same operations in the same order as the
// sampler + rollout pair: same bits in every buffer.
void launch_noise_rows(const FastSampleArgs& a, int rounds, hipStream_t st);  // uses n, first_index, W, seed / offset, out, white
struct IterAheadArgs {
    // rollout role
    FastRolloutArgs r;   // r.actions == pool
    MergeSingleArgs m;   // has_merge: the PREVIOUS iteration's merge (last == 0) -- 1: lists form, in every rollout
    int has_merge;       // workgroup's prologue (world 1); 2: records form, once, by the pack role (sharded runs)
    int n_xf;            // rows [0, n_xf) of the pool hold raw noise
    int row0_mean;       // icem.py:87-88
    int store_back;      // write the actions back over the noise (the last iteration: the caller's pool; the others leave
                         // the noise in place and the next prologue maps the K elite rows again, MergeSingleArgs::n_raw)
    float* pool;         // [n_rows, h, d]
    const float* mean;   // the distribution when has_merge == 0 (else the prologue computes it from m)
    const float* std;
    float lo, hi;        // the action bounds, equal in every action dimension (plan.hip checks before taking this path)
    // noise role: z.n rows of the sampling call (z.off_*) -> z.out (z.n == 0: no such workgroups)
    FastSampleArgs z;
    // shift role: s.n_shift shifted elites from s.elites_src (stream s.off2_*, distribution s.mean / s.std) -> rows
    // [s.n, s.n + s.n_shift) of s.out (== pool) and their costs -> r.costs[s.n ...] (s.n_shift == 0: no such workgroup)
    FastSampleArgs s;
    // sharded runs (world > 1; has_merge == 2): the PREVIOUS iteration's record pack + push rides as workgroup 0, which
    // then runs the launch's one records merge (m: records form) and publishes mean | std (p.pub / p.pub_flag / p.pub_seq,
    // as in sample_folded_merge_kernel); the rollout workgroups wait for that flag instead of merging themselves
    PackPrev p;
    int n_roll, n_noise;  // workgroups per role (filled by the launcher)
    int dbg_slot = 0;     // development: which 16-word block of r.dbg this launch stamps (the iteration)
};
bool rollout_ahead_ok(int h, int d, int O, int K, int n_rows);
int ahead_roll_workgroups(int n_rows);  // = candidate lists of that launch
void launch_iter_ahead(const IterAheadArgs& a, int h, int d, int O, int kind, hipStream_t st);

// K1 with the previous iteration's merge (last == 0) in its prologue, see sample_folded_merge_kernel
struct FastSampleMergeArgs {
    FastSampleArgs s;  // n_shift must be 0
    MergeSingleArgs m;
    PackPrev p;        // sharded runs: the previous iteration's pack rides as workgroup 0
};
bool sample_folded_merge_ok(int h, int d, int rounds, int K);
void launch_sample_folded_merge(const FastSampleMergeArgs& a, hipStream_t st);

// K1+K2+K3 in one launch (small populations): r.actions == s.out, r.n_rows == s.n + s.n_shift.
struct FastIterArgs {
    FastSampleArgs s;
    FastRolloutArgs r;
    MergeSingleArgs m;  // merge prologue only: the PREVIOUS iteration's merge (last == 0), see sample_rollout_kernel
    PackPrev p;         // ... and its pack (sharded runs)
};
// ---- batched planners (icem_plan_step_batch; plan.hip) ----------------------------------------------------------------
// B independent MPC problems of the same configuration advance together: every launch of the small-population path
// (sample_rollout_kernel x opt_iters, then the last merge) is ONE launch for all of them -- blockIdx.y = the problem, its
// argument block read from an array in device memory instead of the kernel-argument segment.  The host side of every
// handle runs as for a solo step, with the three launchers below RECORDING their arguments (g_batch.rec != nullptr) instead
// of launching; plan.hip then issues the batched launches.  Same device code (the kernels' bodies are shared), same bits
// per problem.
constexpr int ICEM_MAX_BATCH = 32;
struct BatchBases {          // by value in the kernel-argument segment: the noise stream offset of this MPC step per problem
    unsigned long long v[ICEM_MAX_BATCH];   // (the offsets in the argument blocks are stored RELATIVE to it: the blocks of a
};                                          //  steady-state step are the previous same-parity step's, and are not uploaded again)
struct BatchRecord {
    int kind = 0;            // 1 sample_rollout, 2 merge_single, 3 merge_noise, 4 iter_ahead (rw = waves per workgroup, grid = rollout workgroups)
    // kind 1
    FastIterArgs it;
    int h = 0, d = 0, O = 0, model_kind = 0, rw = 0, grid = 0;
    bool prologue = false;
    // kinds 2, 3
    MergeSingleArgs m;
    FastSampleArgs z1, z2;
    // kind 4
    IterAheadArgs ia;
};
struct BatchState {
    int mult = 1;                       // problems in the running batch: launch shapes are chosen for mult x the rows
    bool ahead = false;                 // ... and the batch may take the noise-ahead launches where all its rows together fill them
    std::vector<BatchRecord>* rec = nullptr;   // != nullptr: the launchers record here instead of launching
    bool unsupported = false;           // a launcher without a batched form was reached while recording
};
extern thread_local BatchState g_batch;     // (plan.hip)
struct MergeNoiseBatchArgs {
    MergeSingleArgs a;
    FastSampleArgs z1, z2;
};
// the batched launches: args[n] in DEVICE memory (offsets relative to bases.v[problem])
void launch_sample_rollout_batch(const BatchRecord& shape, const FastIterArgs* args_dev, const BatchBases& bases, int n, hipStream_t st);
void launch_merge_batch(const BatchRecord& shape, const MergeNoiseBatchArgs* args_dev, const BatchBases& bases, int n, hipStream_t st);
void launch_iter_ahead_batch(const BatchRecord& shape, const IterAheadArgs* args_dev, const BatchBases& bases, int n, hipStream_t st);

// ---- the whole MPC step of a small population as ONE launch inside one XCD (k_step_xcd.hip; plan.hip::plan_step_xcd) ------
constexpr int STEP_XCD_MAX_ITERS = 8, STEP_XCD_MAX_SEG = STEP_XCD_MAX_ITERS + 1;
struct StepXcdArgs {
    FastRolloutArgs r;            // model, cost, obs0, costs, K, arithmetic (actions / part_k / n_rows: set per iteration by the kernel)
    int iters, K, n_reuse, n_shift, keep, use_mean, white, g0;
    int pop[STEP_XCD_MAX_ITERS];  // rows of iteration it (<= 4096)
    float alpha, init_std;
    float* pool[2];               // pool of iteration it = pool[(iters - 1 - it) & 1] (the last iteration's: the caller's)
    unsigned long long* lists[2]; // [K][32] keys of iteration it = lists[it & 1]
    float* elites;                // [2][K][h*d]
    float* elites_cost;           // [2][K]
    const float* mean;            // the step's first distribution ...
    const float* std;
    float* mean_out;              // ... and what the epilogue leaves (the same buffers)
    float* std_out;
    const float* low;
    const float* high;
    float* executed;
    float* best_cost;
    // colored noise: segment s < iters = iteration s of this step (seg_n == 0: drawn by the previous step's launch), segment
    // iters = iteration 0 of the NEXT step
    const float* W;
    uint32_t seed_lo, seed_hi;
    int n_seg, n_jobs;
    int seg_n[STEP_XCD_MAX_SEG], seg_chunk0[STEP_XCD_MAX_SEG + 1];
    uint32_t seg_off_lo[STEP_XCD_MAX_SEG], seg_off_hi[STEP_XCD_MAX_SEG];
    float* seg_out[STEP_XCD_MAX_SEG];
    const float* raw[STEP_XCD_MAX_ITERS];   // where the members find iteration it's raw rows
    // the shifted elites (icem.py:91-104): previous elite set, noise stream of the call, rows / costs of their own
    const float* shift_src;
    uint32_t shift_off_lo, shift_off_hi;
    float* shift_rows;            // [16][h*d]
    float* shift_costs;           // [16]
    unsigned* state;              // step_xcd_state_bytes()
    unsigned bar_base;            // the members' barrier counter before this launch (cumulative over the handle's launches)
    unsigned max_polls;
};
size_t step_xcd_state_bytes();
int step_xcd_max_rows();
int step_xcd_rows_per_member();
bool step_xcd_supported(int h, int d, int O, int K);
void launch_step_xcd(const StepXcdArgs& a, int h, int d, int O, int kind, hipStream_t st);

// workgroups (= candidate lists) of that launch; 0 if this shape / generator / size has no single-launch kernel
// (n_tail trailing shifted-elite rows; *tail_out > 0: that many rows behind the lists are scored through the cost array)
int sample_rollout_lists(int h, int d, int O, int rounds, int n_rows, int n_tail = 0, int* tail_out = nullptr);
// may iteration `it >= 1` with n_rows rows fold the previous iteration's merge into its own launch?
bool sample_rollout_merge_ok(int h, int d, int O, int rounds, int n_rows, int K);
void launch_sample_rollout(const FastIterArgs& a, int h, int d, int O, int kind, bool merge_prologue, hipStream_t st);

}  // namespace icem
