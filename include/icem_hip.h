/*
 * icem_hip.h -- C ABI of libicem_hip.so: the MI355X-native iCEM inner planning loop.
 *
 * This is the drop-in boundary for ONE hot path of martius-lab/iCEM: everything inside the
 * `for i in range(opt_iter)` loop of MpcICem.get_action (reference icem/controllers/icem.py:106-189).
 * The reference is pure Python/NumPy and has no FFI; each entry point below names the reference
 * call site (file:line, relative to /root/reference) whose arithmetic it replaces.  The binding a
 * reference maintainer would add is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr())
 *     unless its name ends in `_host`.  `stream` is a hipStream_t passed as void*
 *     (torch.cuda.current_stream().cuda_stream); all work is enqueued on it and nothing
 *     synchronises unless stated.
 *   - `dtype` selects the arithmetic/storage type of every floating tensor of a call:
 *     ICEM_F32 (throughput) or ICEM_F64 (the reference's type; strict-parity mode).
 *   - Return value: 0 on success, <0 on error (ICEM_E_*); icem_last_error() returns a
 *     thread-local message.  No exceptions cross the boundary.
 *   - Tensor layouts are C-contiguous: actions [n, h, d]; mean/std [h, d]; low/high [d];
 *     white noise z_r/z_i [n, d, F] with F = h/2+1 (the layout of the reference's draws).
 *   - A handle is not thread-safe; different handles may be used concurrently.
 */
#ifndef ICEM_HIP_H
#define ICEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICEM_ABI_VERSION 5 /* 5: ICEM_TILE_AUTO serves the fp16 planes only where no state can leave their range (icem_tile_growth), icem_nonfinite_costs / ICEM_E_RANGE, icem_set_option (the library reads no environment variable), icem_plan_step_batch; 2: icem_build_hash, icem_allgather_elites / icem_rccl_*, noise-ahead planning; ICEM_MAX_OBS_DIM 384; 3: icem_set_tile_arith; 4: icem_set_wide_arith (default AUTO), ICEM_TILE_AUTO = planes at every population */

enum { ICEM_F32 = 0, ICEM_F64 = 1 };
enum { ICEM_COST_SUM = 0, ICEM_COST_BEST = 1, ICEM_COST_FINAL = 2 }; /* abstract_controller.py:82-87 */
enum { ICEM_MODEL_LINEAR = 0, ICEM_MODEL_TANH = 1 };

enum {
    ICEM_OK = 0,
    ICEM_E_INVALID = -1,      /* bad argument (the reference raises ValueError/AttributeError)   */
    ICEM_E_UNSUPPORTED = -2,  /* shape outside the compiled kernels (NotImplementedError)          */
    ICEM_E_HIP = -3,          /* a HIP runtime call failed                                         */
    ICEM_E_NO_DEVICE = -4,    /* no gfx950 device visible                                          */
    ICEM_E_STATE = -5,        /* call order violated (e.g. rollout before icem_set_model)          */
    ICEM_E_RANGE = -6         /* icem_get_action: the step's outputs are returned, but trajectories came back with a
                                 non-finite cost from a finite observation (a state left the arithmetic's range)       */
};

#define ICEM_MAX_HORIZON 64
#define ICEM_MAX_ACT_DIM 64
#define ICEM_MAX_OBS_DIM 384
#define ICEM_MAX_ELITES 64

/* Constructor kwargs of MpcICem (icem.py:213-233, mpc.py:22, abstract_controller.py:64-65). */
typedef struct icem_config {
    int32_t horizon;              /* h                                                   */
    int32_t act_dim;              /* d  = env.action_space.shape[0] (icem.py:245)        */
    int32_t num_traj;             /* N  = num_simulated_trajectories (GLOBAL, all ranks) */
    int32_t num_elites;           /* K  = min(elites_size, N/2), >= 2 (icem.py:235-240)  */
    int32_t elites_size;          /* raw elites_size (the N-decay floor is 2*elites_size, icem.py:127) */
    int32_t opt_iters;            /* opt_iterations                                      */
    int32_t cost_mode;            /* ICEM_COST_*  (cost_along_trajectory)                */
    int32_t use_mean_actions;     /* icem.py:87-88                                       */
    int32_t keep_previous_elites; /* icem.py:143-145                                     */
    int32_t shift_elites;         /* icem.py:131-137                                     */
    int32_t dtype;                /* ICEM_F32 / ICEM_F64                                 */
    int32_t rng_rounds;           /* Philox4x32 rounds: 10 (default) or 7                */
    int32_t rank;                 /* this process' shard of the N dimension              */
    int32_t world;                /* number of shards (GPUs)                             */
    double factor_decrease;       /* gamma = factor_decrease_num                         */
    double alpha;                 /* momentum                                            */
    double init_std;              /* relative to (high-low)/2 (icem.py:55-59)            */
    double fraction_reused;       /* xi = fraction_elites_reused                         */
    double noise_beta;            /* colored-noise exponent; <= 0: white noise (icem.py:77) */
    uint64_t seed;                /* Philox key                                          */
} icem_config;

/* Parametric form of the shipped cost functions (environments/mujoco.py:67-99 HalfCheetah,
 * :259-277 HumanoidStandup):
 *   cost_t = ctrl_weight*sum_j a_j^2 + lin_weight*obs[lin_idx]
 *            + flip_penalty*([obs[flip_idx] > flip_thresh] + [obs[flip_idx] < -flip_thresh])
 * with the flip term dropped when flip_idx < 0 and the linear term when lin_weight == 0 (lin_idx must still be a
 * valid index).  `obs` is the PRE-action observation. */
typedef struct icem_cost_spec {
    double ctrl_weight;
    double lin_weight;
    double flip_penalty;
    double flip_thresh;
    int32_t lin_idx;
    int32_t flip_idx;
} icem_cost_spec;

/* The remaining parametric cost functions of the reference's environments, as extra terms on top of
 * icem_cost_spec (a zero-initialised struct with diff_idx = health_idx = box_from = -1 and n_terms = 0 is "off"):
 *   + diff_weight * (next_obs[diff_idx] - obs[diff_idx])            Ant, Hopper: -x_velocity = -(x' - x)/dt
 *                                                                    (environments/mujoco.py:164, 218)
 *   + health_penalty * unhealthy(obs),  unhealthy = 1 - all_finite(obs) * [lo <(=) obs[health_idx] <(=) hi]
 *                                                   * [box_lo < obs[k] < box_hi for all k >= box_from]
 *       Ant :146-149,166 (closed range), Hopper :189-203,221 (open range + healthy_state_range over obs[2:];
 *       its healthy_angle is dropped by the reference itself, :199), Humanoid :302-315,338 (open range)
 *   + sum over terms[0 .. n_terms) of  weight * gate * f,  gate = [obs[gate_idx] > gate_thresh] (1 if gate_idx < 0;
 *     a product as in the reference, so a non-finite f is not masked by a closed gate),
 *     with r = || obs[a .. a+len) - obs[b .. b+len) ||  (b < 0: the norm of the slice itself) and f by kind:
 *       ICEM_TERM_NORM       r                    Reacher :366-368, FetchPickAndPlace / FetchReach
 *                                                 environments/robotics.py:150-164, 286-295, Door / Relocate
 *                                                 environments/mjenvs.py:64, 161, 166 (the latter gated on the lift)
 *       ICEM_TERM_NORM_GT    [r > thresh]         the sparse robotics costs robotics.py:159-161, 291-292
 *       ICEM_TERM_NORM_LT    [r < thresh]         Relocate's closeness bonuses mjenvs.py:170-172
 *       ICEM_TERM_SQ_OFFSET  (obs[a] - thresh)^2  Door's hinge target mjenvs.py:68
 *       ICEM_TERM_SUMSQ      sum_k obs[a+k]^2     Door's velocity cost mjenvs.py:70
 *       ICEM_TERM_STEP_GT    [obs[a] > thresh]    Door's opening bonuses :74-76, Relocate's lift bonus :162 */
enum { ICEM_TERM_NORM = 0, ICEM_TERM_NORM_GT = 1, ICEM_TERM_NORM_LT = 2, ICEM_TERM_SQ_OFFSET = 3, ICEM_TERM_SUMSQ = 4,
       ICEM_TERM_STEP_GT = 5 };
#define ICEM_MAX_COST_TERMS 8
#define ICEM_MAX_TERM_LEN 64

typedef struct icem_cost_term {
    double weight, thresh, gate_thresh;
    int32_t kind;
    int32_t a, b, len;      /* b = -1: no second slice; len <= ICEM_MAX_TERM_LEN (1 for the scalar kinds) */
    int32_t gate_idx;       /* -1: not gated */
    int32_t reserved;
} icem_cost_term;

typedef struct icem_cost_terms {
    double diff_weight;
    double health_penalty, health_lo, health_hi;
    double box_lo, box_hi;
    int32_t diff_idx;        /* -1: off */
    int32_t health_idx;      /* -1: off */
    int32_t health_closed;   /* 1: lo <= z <= hi, 0: lo < z < hi */
    int32_t box_from;        /* -1: off */
    int32_t n_terms;
    int32_t reserved;
    icem_cost_term terms[ICEM_MAX_COST_TERMS];
} icem_cost_terms;

typedef struct icem_handle icem_handle;

/* ---- library / handle ------------------------------------------------------------------ */

int icem_abi_version(void);
const char* icem_last_error(void);
/* Hash (16 hex digits) of the sources this binary was built from -- what `python -m icem_amd.build` compares with the
 * tree to tell a stale library from a current one (no reference counterpart: the reference is interpreted Python). */
const char* icem_build_hash(void);

/* Number of visible HIP devices (0 on a CPU-only host); never fails. */
int icem_device_count(void);

/* Creates a planner for one controller instance (replaces MpcICem.__init__ state, icem.py:22-29).
 * Builds the colored-noise synthesis table on the current device. */
int icem_create(const icem_config* cfg, icem_handle** out);
int icem_destroy(icem_handle* h);

/* Episode number of the rollout being planned, folded into the device noise streams (high word of the stream offset,
 * the sampling call number mpc_step*(opt_iters+1)+it is the low word): the reference draws from one np.random stream
 * that runs on across episodes (icem.py:73-77), so consecutive episodes must not replay the same exploration noise.
 * 0 after icem_create; the host mirror advances it in beginning_of_rollout (icem.py:31-43).  episode < 2^32. */
int icem_set_episode(icem_handle* h, uint64_t episode);

/* Sequence of population sizes N_i the loop will use (icem.py:123-127); out_host[opt_iters]. */
int icem_population_sizes(const icem_handle* h, int32_t* out_host);

/* Colored-noise synthesis matrices (float64, host): y[t] = sum_k z_r[k]*cr[k*h+t] + z_i[k]*ci[k*h+t]
 * restating colorednoise.powerlaw_psd_gaussian (third-party; call site icem.py:73-75). */
int icem_noise_tables_host(int32_t horizon, double beta, double* cr_host, double* ci_host);

/* Built-in batched forward model o' = act(o.A + a.B) (the `predict` contract of
 * models/abstract_models.py:17-26).  A [obs_dim, obs_dim], B [act_dim, obs_dim] are HOST float64
 * arrays; they are converted to the handle dtype and copied to the device.  `kind` = ICEM_MODEL_*.
 * obs_dim <= 32: every kernel family (f32 / f64).  32 < obs_dim <= 384 (HumanoidStandup's o = 378,
 * environments/mujoco.py:241-252): dtype f32 only, the model step runs as an exact-f32 GEMM on the matrix pipe
 * (icem_rollout_cost without `observations`, icem_plan_*); cost = icem_set_cost's form + icem_set_cost_terms
 * (Ant's o = 111/113, Humanoid's 376, Door / Relocate's 39: mujoco.py:151-171, 317-343, mjenvs.py:57-78, 155-174). */
int icem_set_model(icem_handle* h, int32_t kind, int32_t obs_dim, const double* A_host, const double* B_host);
int icem_set_cost(icem_handle* h, const icem_cost_spec* spec);
/* Extra cost terms (NULL switches them off again).  With any of them on, icem_rollout_cost / icem_plan_* evaluate
 * the cost in the general rollout kernel (one thread per trajectory) for obs_dim <= 32 and inside the matrix-pipe
 * rollout of k_rollout_wide.hip for 32 < obs_dim <= 384. */
int icem_set_cost_terms(icem_handle* h, const icem_cost_terms* terms);

/* ---- stateless operators (each replaces one NumPy call site) ------------------------------ */

/* K1  MpcICem.sample_action_sequences (icem.py:61-82) + colorednoise.powerlaw_psd_gaussian:
 *   actions[i, t, j] = clip(colored(i, j)[t] * std[t, j] + mean[t, j], low[j], high[j]),  i < n.
 * White noise: if z_r/z_i are non-NULL they are the [n, d, F] standard-normal draws (parity mode:
 * the same draws the reference takes from np.random); otherwise Philox4x32 keyed by
 * (cfg.seed, offset, first_index + i, j).  Only time steps t >= t_begin are written
 * (t_begin = h-1 restates the time_slice=slice(-1, None) call of icem.py:102).
 * cfg.noise_beta <= 0 is the reference's white branch (np.random.randn(num_traj, h, d), icem.py:77): no synthesis,
 * draw t of row (i, j) is its sample at step t; an external draw is then passed as z_r = randn [n, h, d], z_i NULL.
 * If row0_mean != 0 and first_index == 0, row 0 is overwritten with `mean` (icem.py:87-88). */
int icem_sample_clip(icem_handle* h, int32_t n, int64_t first_index, const void* mean, const void* std,
                     const void* low, const void* high, const void* z_r, const void* z_i,
                     uint64_t offset, int32_t t_begin, int32_t row0_mean, void* actions, void* stream);

/* MpcCemStd.sample_action_sequences (the CEM baseline, icem/controllers/mpc.py:188-198):
 *   actions[i, t, j] = mean[t, j] + std[t, j] * truncnorm.ppf(u[i, t, j]; lower[t, j], upper[t, j])
 * (scipy.stats.truncnorm.rvs draws ONE uniform(size=(n, h, d)) per call).  u: those uniforms [n, h, d] (parity mode),
 * or NULL: word t of the Philox / xoshiro stream of row (first_index + i, j), offset as in icem_sample_clip.
 * lower / upper are [h, d] in standard-normal units, as MpcCemStd keeps them. */
int icem_sample_truncnorm(icem_handle* h, int32_t n, int64_t first_index, const void* mean, const void* std,
                          const void* lower, const void* upper, const void* u, uint64_t offset, void* actions,
                          void* stream);
/* MpcCemStd._update_bounds (mpc.py:290-301): like_levine != 0 caps std at half the distance to the action bounds
 * (in place, floor 1e-8) and sets lower / upper = -2 / +2; otherwise lower / upper = (low|high - mean) / (std + 1e-8). */
int icem_cem_bounds(icem_handle* h, int32_t like_levine, const void* mean, void* std, const void* low, const void* high,
                    void* lower, void* upper, void* stream);

/* MpcRandom.sample_action_sequences (icem/controllers/mpc.py:96-109, random shooting): actions[i, t, :] = low +
 * (high - low) * U(block), uniform draws held for consecutive calls of MpcRandom.sample() (mpc.py:96-102) -- one call
 * per (trajectory, step) pair in row-major order, the call counter running on across MPC steps: call c =
 * call_offset + i*h + t uses block 0 (the action drawn at construction) while c < change_freq, then block
 * 1 + (c - change_freq) / (change_freq + 1).  u: uniforms [*, d] for blocks first_block, first_block + 1, ...
 * (parity mode: env.action_space.sample()'s draws), or NULL for the Philox path (keyed by the block index). */
int icem_sample_piecewise(icem_handle* h, int32_t n, int64_t call_offset, int32_t change_freq, int64_t first_block,
                          const void* low, const void* high, const void* u, void* actions, void* stream);

/* Raw white noise of the Philox path (for RNG known-answer tests): z_r, z_i [n, d, F]. */
int icem_philox_normals(icem_handle* h, int32_t n, int64_t first_index, uint64_t offset,
                        void* z_r, void* z_i, void* stream);

/* K2  MpcController.simulate_trajectories (mpc.py:56-67) through the batched model loop
 * (abstract_models.py:17-53) + trajectory_cost_fn (abstract_controller.py:74-91) with the built-in
 * model/cost:  costs[i] = reduce_t cost(o_t, a_t),  o_{t+1} = model(o_t, a_t),  o_0 = obs0[obs_dim].
 * If observations != NULL the pre-action observations [n, h, obs_dim] are also stored. */
int icem_rollout_cost(icem_handle* h, int32_t n, const void* obs0, const void* actions, void* costs,
                      void* observations, void* stream);

/* trajectory_cost_fn's reduction alone (abstract_controller.py:82-87) for step costs [n, h]
 * produced by an external (e.g. torch) model. */
int icem_cost_reduce(icem_handle* h, int32_t n, const void* step_costs, void* costs, void* stream);

/* trajectory_cost_fn as a whole (abstract_controller.py:74-91) for rollouts produced by an external model and
 * held as tensors: costs[i] = reduce_t cost(observations[i, t, :], actions[i, t, :], next_observations[i, t, :])
 * with the cost set by icem_set_cost (+ icem_set_cost_terms) and the handle's cost_mode.  Any obs_dim.
 * observations / next_observations: element (i, t, k) at ptr[i*traj_stride + t*step_stride + k] (so [n, h, o] and
 * step-major [h, n, o] both work); next_observations may be NULL unless diff_idx >= 0.  actions [n, h, d] dense. */
int icem_trajectory_cost(icem_handle* h, int32_t n, int32_t obs_dim, const void* observations,
                         const void* next_observations, int64_t traj_stride, int64_t step_stride,
                         const void* actions, void* costs, void* stream);

/* K3  costs.argsort()[:k] (icem.py:199) and argmin (icem.py:149): the k smallest (cost, index)
 * pairs in ascending order, ties broken by index, NaN treated as +inf.
 * out_cost [k] (handle dtype), out_idx [k] int32 (k > n: entries n.. are (+inf, INT_MAX)).  workspace:
 * icem_topk_workspace_bytes(n, k). */
size_t icem_topk_workspace_bytes(const icem_handle* h, int32_t n, int32_t k);
int icem_topk_sorted(icem_handle* h, int32_t n, const void* costs, int32_t k, void* out_cost,
                     int32_t* out_idx, void* workspace, void* stream);

/* K3 + K4 in one launch for small pools (the stage-wise controller loop: learned dynamics, host models, the CEM
 * baselines): update_distributions (icem.py:194-211) with the previous elites appended behind the pool
 * (icem.py:143-145) -- the k best of [costs (n) | keep_costs (n_keep)] in (cost, index) order, index n + e = kept elite e;
 * their rows from [pool | keep_actions] into elites_out [k, h, d] (must not alias keep_actions), elite_costs_out [k],
 * idx_out [k]; mean / std refitted in place as icem_gather_refit does (same arithmetic, same bits as icem_topk_sorted
 * over the concatenation + icem_gather_refit).  f32 handles, n + n_keep <= 16384, k <= 32; otherwise
 * ICEM_E_UNSUPPORTED: use the two operators. */
int icem_update_distribution(icem_handle* h, int32_t n, const void* costs, const void* pool, int32_t n_keep,
                             const void* keep_costs, const void* keep_actions, int32_t k, void* mean, void* std,
                             void* elites_out, void* elite_costs_out, int32_t* idx_out, void* stream);
/* 1 if THIS handle serves icem_update_distribution for a pool of n_all = n + n_keep candidates and k elites (the handle's
 * dtype and the fast-path switch it latched at icem_create decide, not the caller's environment), else 0. */
int icem_update_distribution_ok(const icem_handle* h, int32_t n_all, int32_t k);

/* K4  update_distributions (icem.py:201-211): gather the k elite rows of `actions` [*, h, d] in
 * `idx` order into elites_out [k, h, d]; mean <- (1-alpha)*mean_k + alpha*mean,
 * std <- (1-alpha)*std_k(ddof=0) + alpha*std (in place).  An entry INT_MAX (padding of icem_topk_sorted) repeats row idx[0]. */
int icem_gather_refit(icem_handle* h, const void* actions, const int32_t* idx, int32_t k, void* mean,
                      void* std, void* elites_out, void* stream);

/* Epilogue of get_action (icem.py:167-175): mean[:-1] = mean[1:] (last row kept);
 * std = (high-low)/2*init_std. */
int icem_shift(icem_handle* h, void* mean, void* std, const void* low, const void* high, void* stream);

/* beginning_of_rollout (icem.py:31-59): mean = (high+low)/2, std = (high-low)/2*init_std.  The action bounds are
 * read when this is called and, on the f32 large-population path, once more per (low, high) buffer pair at the next
 * step: a caller that rewrites the bounds IN PLACE does so between episodes, in front of this call. */
int icem_reset_distribution(icem_handle* h, void* mean, void* std, const void* low, const void* high,
                            void* stream);

/* ---- fused MPC step ------------------------------------------------------------------------ */

/* Device buffers of one planner; all owned by the caller, sized by icem_plan_buffer_bytes(). */
typedef struct icem_plan_buffers {
    void* mean;        /* [h, d]              in/out: persistent across MPC steps               */
    void* std;         /* [h, d]              in/out                                            */
    void* low;         /* [d]                                                                   */
    void* high;        /* [d]                                                                   */
    void* obs0;        /* [obs_dim]           current observation                               */
    void* actions;     /* [n_local_max + r, h, d]  sampled pool of this rank (r = reused elites)*/
    void* costs;       /* [n_local_max + r]                                                     */
    void* elites;      /* 2 x [K, h, d] + 2 x [K] costs: persistent elite set (double buffered) */
    void* records;     /* [world*K + r] candidate records {cost, gidx, actions[h*d]}            */
    void* workspace;   /* scratch for the block-level top-k                                     */
    void* executed;    /* [d]   out: first action of the best trajectory (icem.py:163)          */
    void* best_cost;   /* [1]   out: min(costs) of the last iteration (icem.py:177)             */
    /* Optional external white noise for the CURRENT iteration (parity mode, NULL = Philox):
     * z_r/z_i [n_local, d, F] for the main batch; z_r_shift/z_i_shift [r, d, F] for the
     * shifted elites' last action (iteration 0 only).  noise_beta <= 0: z_r = randn [n_local, h, d] and
     * z_r_shift = randn [r, h, d] (icem.py:77), z_i / z_i_shift NULL.                           */
    const void* z_r;
    const void* z_i;
    const void* z_r_shift;
    const void* z_i_shift;
} icem_plan_buffers;

enum {
    ICEM_BUF_MEAN = 0, ICEM_BUF_STD, ICEM_BUF_LOW, ICEM_BUF_HIGH, ICEM_BUF_OBS0, ICEM_BUF_ACTIONS,
    ICEM_BUF_COSTS, ICEM_BUF_ELITES, ICEM_BUF_RECORDS, ICEM_BUF_WORKSPACE, ICEM_BUF_EXECUTED,
    ICEM_BUF_BEST_COST, ICEM_BUF_COUNT
};
/* Bytes required for buffer `which` (ICEM_BUF_*) of icem_plan_buffers. */
size_t icem_plan_buffer_bytes(const icem_handle* h, int32_t which);

/* One CEM iteration `it` (0..opt_iters-1) of MPC step number `mpc_step` (0 = first after
 * beginning_of_rollout), local part: sample this rank's shard of N_it (+ shifted / kept elites),
 * roll out, cost, block-level top-k, and pack this rank's sorted local top-K candidate records
 * into records[rank*K .. rank*K+K).  icem.py:124-147. */
int icem_plan_iter_local(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it,
                         void* stream);
/* After the ranks' records have been all-gathered in place (world > 1; nothing to do for world == 1):
 * merge world*K candidates (+ kept elites), take the global sorted top-K, refit mean/std, store the
 * new elite set; on the last iteration also write executed/best_cost and shift mean/std
 * (icem.py:149-175, 194-211). */
int icem_plan_iter_merge(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it,
                         void* stream);
/* Sharded runs (world > 1): allow icem_plan_iter_merge of a non-last iteration to postpone its work into the next
 * icem_plan_iter_local launch of the same MPC step (the merge of the gathered records then runs in that launch's
 * prologue, next to the sampling; one kernel launch less per iteration).  While on, mean / std / elites in the plan
 * buffers are current only after the LAST icem_plan_iter_merge of an MPC step.  Off by default. */
int icem_set_merge_deferral(icem_handle* h, int32_t on);

/* ---- in-library elite exchange (world > 1) ------------------------------------------------------------------
 * The all-gather of every rank's K candidate records {cost, gidx, actions[h*d]} before the replicated refit (the
 * reference's only data parallelism gathers its workers' results over pipes, icem/models/gt_par_model.py:77-94).
 * Each rank owns an exchange block in its HBM and maps the blocks of all others (HIP IPC); a rank's records reach its
 * peers as peer-to-peer stores over xGMI from ONE launch behind its local kernels, and every merge waits on flags in
 * its own block -- no host call and no collective between icem_plan_iter_local and icem_plan_iter_merge.
 *   1. icem_exchange_create on every rank: allocates the block, returns its IPC handle (ICEM_IPC_HANDLE_BYTES bytes);
 *   2. the caller moves the handles between the ranks' processes once (any channel: torch.distributed, MPI, a file);
 *   3. icem_exchange_connect with all `world` handles in rank order.  Ranks living in the CALLER'S process are passed
 *      as device pointers instead (local_blocks[r] = icem_exchange_block(peer handle); NULL entries use the IPC handle;
 *      handles_host may be NULL when every peer is local).
 * From then on icem_plan_iter_local ends with the push and icem_plan_iter_merge reads the gathered records from the
 * block (b->records then only receives this rank's own K records).  Every device-side wait is bounded (~seconds); a
 * timeout sets the block's status word, which icem_exchange_status returns and clears (0 = fine). */
#define ICEM_IPC_HANDLE_BYTES 64
int icem_exchange_create(icem_handle* h, void* ipc_handle_out_host);
int icem_exchange_connect(icem_handle* h, const void* handles_host, void* const* local_blocks);
void* icem_exchange_block(icem_handle* h);
/* Back to caller-driven exchanges (frees the block, unmaps the peers'): every rank must do the same. */
int icem_exchange_disable(icem_handle* h);
/* finegrained_host (may be NULL): 1 if the block is fine-grained device memory (what multi-GPU runs want). */
int icem_exchange_status(icem_handle* h, int32_t* status_host, int32_t* finegrained_host);
/* Measurement only (collective: every rank calls it at the same time): average latency [us] of one exchange -- this
 * rank's K records to every rank's block, then the wait for all ranks' -- over `rounds` back-to-back exchanges inside
 * one launch.  Synchronises the stream. */
int icem_exchange_probe(icem_handle* h, int32_t rounds, void* stream, double* us_out);
/* ---- the same all-gather on RCCL (world > 1) -------------------------------------------------------------------
 * SURVEY 8(b)'s icem_allgather_elites: ncclAllGather of the ranks' K records, in place in `records`
 * [world*K, 2 + h*d] (this rank's slot written by icem_plan_iter_local), enqueued on the launch stream between the
 * record pack and the merge.  It is the fallback of the exchange above -- for ranks whose peer blocks do not map or
 * whose self-test fails -- and keeps an MPC step ONE C call with zero host-side collectives (icem_plan_step_sharded).
 * Same reference analogue: the pipe gather of icem/models/gt_par_model.py:77-94.  RCCL is bound at run time (dlopen:
 * the copy already loaded in the process, else icem_rccl_load's path / $ICEM_RCCL_LIB / librccl.so.1); nothing here is
 * needed, or loaded, on one GPU.
 *   icem_rccl_unique_id   rank 0: a fresh ncclUniqueId (ICEM_RCCL_ID_BYTES bytes) for the caller to hand to every rank
 *   icem_rccl_connect     every rank, together: ncclCommInitRank(world, id, rank) -- the handle owns the communicator
 *   icem_rccl_adopt       or: use a communicator the caller already has (an ncclComm_t; size / rank must match)
 *   icem_allgather_elites the collective itself (every rank, same stream order)
 *   icem_rccl_disconnect  destroys an owned communicator / forgets an adopted one */
#define ICEM_RCCL_ID_BYTES 128
int icem_rccl_load(const char* path_or_null);
const char* icem_rccl_library(void); /* which library was bound ("" = none yet) */
int icem_rccl_unique_id(void* id_out_host);
int icem_rccl_connect(icem_handle* h, const void* id_host);
int icem_rccl_adopt(icem_handle* h, void* nccl_comm);
int icem_rccl_disconnect(icem_handle* h);
int icem_allgather_elites(icem_handle* h, void* records, void* stream);

/* Whole MPC step of one rank of a sharded run (world > 1, device noise) as ONE call, no host synchronisation and no
 * host-side collective: opt_iters x (local launch, pack, exchange, merge -- every merge but the last in the next local
 * launch's prologue).  The exchange is the in-library one when it is connected (pack + push in one launch), else
 * icem_allgather_elites when a communicator is connected; neither: ICEM_E_STATE.  world == 1: icem_plan_step. */
int icem_plan_step_sharded(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, void* stream);

/* Whole MPC step for world == 1: opt_iters x (local + merge), no host synchronisation
 * (the body of MpcICem.get_action, icem.py:123-175). */
int icem_plan_step(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, void* stream);

/* B independent planners of ONE configuration -- the reference's parallel episodes, each its own get_action
 * (icem/misc/rollout_utils.py:46-58, 129-152; icem.py:106-189) -- advanced by one MPC step together: every stage of the
 * small-population path is one launch for all of them (grid.y = the problem), so the step's chain of launch latencies is paid
 * once per batch instead of once per problem.  handles[i] with buffers[i]: exactly the arguments icem_plan_step takes; every
 * problem's outputs (executed action, best cost, mean, std, elites, pool, costs) are bit for bit those of its own
 * icem_plan_step.  The handles must share horizon, act_dim, populations, elites, flags, model width / kind and tile
 * arithmetic; models, costs, seeds, bounds and observations are per problem.  world == 1, f32, device noise, o <= 20 tile
 * shapes, populations that take the single-launch kernel alone (<= 8192 rows per iteration): otherwise ICEM_E_UNSUPPORTED and
 * nothing is launched.  (From 49 152 rows of all problems together the batch takes the noise-ahead launches of large
 * populations -- k_rollout_ahead.hip -- instead of the single-launch kernels: measured faster from there on.)  n in [1, 32]; n == 1 is icem_plan_step.  No host synchronisation; one small host-to-device copy
 * on `stream` when the argument blocks changed (the first steps). */
int icem_plan_step_batch(icem_handle* const* handles, int32_t n, const icem_plan_buffers* buffers, int32_t mpc_step, void* stream);
int64_t icem_batch_uploads(const icem_handle* h); /* how often handles[0]'s argument array was (re)written (measurement) */

/* Under option step_xcd = 1 (icem_set_option; OFF by default: built, bit for bit the default path's results, and measured
 * slower -- 87 against 61 us per MPC step at N = 4096, EXPERIMENTS R6.4) icem_plan_step / icem_get_action serve a small
 * population (every iteration <= 4096 rows, the o <= 20 tile shapes, f32, device noise, a 256-CU device) with ONE launch per
 * MPC step whose rollout workgroups all sit on one XCD and meet at the iteration boundaries inside its L2 (k_step_xcd.hip);
 * every wait of that launch is bounded.  icem_step_status: how many such launches the handle has issued and whether a wait
 * ever ran out (then that step's outputs are not valid, and the handle keeps to one launch per iteration from here on).
 * Copies one word back and synchronises `stream`. */
int icem_step_status(icem_handle* h, int64_t* xcd_launches_out, int32_t* timed_out_out, void* stream);

/* Wide observations (32 < obs_dim <= 384; HumanoidStandup's o = 378, environments/mujoco.py:241-277): which matrix-pipe
 * arithmetic the rollout's model step (the GEMM of abstract_models.py:31-53's predict at this width) runs in.  The modes
 * are NAMES, not an accuracy order:
 *   ICEM_WIDE_F32 (1): v_mfma_f32_16x16x4_f32 -- bitwise an f32 fmaf chain, a quarter of the speed.  What strict-parity
 *     callers pass; the only arithmetic at widths the split kernel's LDS does not hold (o + d > 416).
 *   ICEM_WIDE_F16X2 (0): every f32 operand x 2^k (k per trajectory row / per model, so that nothing exceeds 2^15) as the
 *     sum of two fp16 numbers, three fp16 products per multiply-add with f32 accumulation on v_mfma_f32_16x16x32_f16 --
 *     f32-class rounding (operands to 2^-24 relative, the dropped lo x lo product below 2^-24 of a product), NOT the bits
 *     of an f32 fmaf chain; a tanh model's state is bounded by the equilibrated scales instead of scanned.  The model is
 *     carried equilibrated -- rows and columns scaled by powers of two -- so observations in mixed units keep their
 *     accuracy; contributions more than 2^13 apart inside the balanced model keep an absolute, not a relative, accuracy:
 *     2^-40 of a row's largest.
 *   ICEM_WIDE_BF16X3 (2): three bf16 numbers and six products (operands exact whatever their magnitudes, dropped products
 *     below 2^-31), two thirds of F16X2's speed.
 *   ICEM_WIDE_AUTO (-1, THE DEFAULT since ABI 4; ABI 3 defaulted to F16X2 unconditionally, ABI 2 to F32): F16X2 where one
 *     sweep of balancing leaves every row and column of the model with its largest weight within 2^13 of the model's
 *     largest AND its median nonzero weight within 2^13 of its own largest (icem_wide_imbalance_log2 <= 13: the larger of
 *     the two measures over all lines), BF16X3 otherwise -- decided per model at icem_set_model.  A caller that needs the
 *     fmaf chain's bits (golden vectors recorded at this width in f32) must ask for ICEM_WIDE_F32; the parity bar of
 *     north_star (1e-5 relative, identical elite sets) holds in all three (tests/test_gpu_parity_sizes.py).
 * icem_wide_arith: the arithmetic in effect for the handle's current model (never AUTO).  Takes effect at the next
 * rollout; no effect at obs_dim <= 32 (narrow models on the GEMM kernel always compute in exact f32: icem_wide_arith = 1).
 * Other values: ICEM_E_INVALID.  icem_set_wide_exact(h, on) is the ABI <= 3 spelling of icem_set_wide_arith(h, on), on in {0, 1, 2}. */
enum { ICEM_WIDE_AUTO = -1, ICEM_WIDE_F16X2 = 0, ICEM_WIDE_F32 = 1, ICEM_WIDE_BF16X3 = 2 };
int icem_set_wide_arith(icem_handle* h, int32_t mode);
int icem_wide_arith(const icem_handle* h);
int icem_wide_imbalance_log2(const icem_handle* h); /* of the balanced model: max over rows / columns of log2(model max / line max) and log2(line max / line median) */
/* the same measure for any model (host arrays as icem_set_model takes them; no handle, no device): -1 on bad arguments */
int icem_wide_model_imbalance_log2(int32_t obs_dim, int32_t act_dim, const double* A_host, const double* B_host);
int icem_set_wide_exact(icem_handle* h, int32_t on);

/* Narrow observations (the 16-trajectory tile kernels, 16 <= padded obs_dim <= 20: HalfCheetah's o = 17 / 18): which
 * arithmetic the model step (abstract_models.py:17-26's predict) runs in.
 *   ICEM_TILE_F32 (0): v_mfma_f32_16x16x4_f32 -- bitwise an f32 fmaf chain; the single-launch kernel of small populations
 *     runs the same chain on the VALU, so every launch shape gives the same bits.
 *   ICEM_TILE_F16X2 (1): every f32 operand x S (S one power of two per launch, from max(|obs0|, action bound)) as the sum
 *     of two fp16 numbers, three fp16 products per multiply-add with f32 accumulation on v_mfma_f32_16x16x32_f16 -- f32-class
 *     rounding (operands to 2^-24 relative down to 2^-7 of the largest, 2^-29 of the largest below), a quarter of the
 *     matrix-pipe time of the exact form; NOT the bits of an fmaf chain.  fp16 ends at 65 504 = 2^11 x the scaled magnitude,
 *     so the planes are SERVED ONLY WHERE NO STATE CAN GET THERE: for a linear model the reachable maximum of
 *     |state entry| / max(|obs0|, action bound) over the horizon -- over every start observation and every action sequence
 *     inside the bounds; icem_tile_growth returns it -- must not exceed 2^10 (a tanh model's state is bounded by 1), and the
 *     largest |entry| of B must not exceed 2^10 x A's (the action operand's scale).  Every other model keeps the
 *     exact form whatever is asked -- with A = 1.5 I over 30 steps the reference ranks finite costs of 1e5 (icem.py:147-159,
 *     199) and so does this library -- and icem_tile_arith tells which arithmetic a handle's launches use.
 *   ICEM_TILE_AUTO (-1, the default): F16X2 wherever it is served (the widths, models and thresholds above), at every
 *     population -- since ABI 4; ABI 3 kept configurations with an iteration of at most 8192 rows on F32, which measured
 *     SLOWER there too (N = 4096 x 5 iterations: 66.7 -> 61.9 us per MPC step).  Decided from the configuration alone -- the
 *     handle's, never the process environment's: every rank of a sharded run and every iteration of a decaying population
 *     computes in the same arithmetic.  Strict-parity callers pass ICEM_TILE_F32.
 * The same setting governs the Door / Relocate / FetchPickAndPlace shapes' TileHN rollout (o = 28 / 39; k_rollout_hn.hip --
 * fp16 planes by default under the same range rules; ICEM_TILE_F32, and at o = 39 also icem_set_wide_arith(ICEM_WIDE_F32),
 * puts them on the exact-f32 GEMM kernel).
 * Takes effect at the next launch (not between icem_plan_iter_local and its merge).  No reference counterpart.
 * The scale S takes the action bound from the (low, high) buffers of the plan calls and of icem_reset_distribution (fetched
 * once per buffer pair and again at every reset -- the first such call on a stream must not be inside a stream capture);
 * icem_rollout_cost on actions OUTSIDE those bounds can still leave the planes' range: see icem_nonfinite_costs. */
enum { ICEM_TILE_AUTO = -1, ICEM_TILE_F32 = 0, ICEM_TILE_F16X2 = 1 };
int icem_set_tile_arith(icem_handle* h, int32_t mode);
int icem_tile_arith(const icem_handle* h); /* the arithmetic in effect: ICEM_TILE_F32 or ICEM_TILE_F16X2 */
double icem_tile_growth(const icem_handle* h); /* reachable max of |state| / max(|obs0|, action bound) over the horizon (1: tanh) */

/* Status word of the tile rollouts (every launch of the o <= 48 kernels counts into it): the number of trajectories since
 * icem_create whose cost came out NaN -- a state that left the arithmetic's range (f32's; the fp16 planes' for actions
 * outside the bounds their scale was taken from) or non-finite inputs.  Such a trajectory ranks last, as a NaN does under
 * icem.py:199's argsort; the reference's float64 would have ranked a finite cost, so a caller that fed finite inputs must
 * not trust the step: icem_get_action returns ICEM_E_RANGE (outputs filled) when its own step raised the count from a finite
 * observation; icem_plan_step callers read it here.  Copies one word back and synchronises `stream`. */
int icem_nonfinite_costs(icem_handle* h, int64_t* count_out, void* stream);

/* Development options: which of several BIT-IDENTICAL launch arrangements serves a call, tuning fractions, the bound of the
 * exchange's device-side waits (names and defaults: icem_amd/csrc/options.h; icem_option_name(i) enumerates them, NULL
 * behind the last).  One process-wide table; the library reads NO environment variable -- tools map ICEM_<NAME> variables
 * onto it explicitly (icem_amd._lib.apply_env_options).  No option selects an arithmetic. */
int icem_set_option(const char* name, double value);
int icem_get_option(const char* name, double* value_out);
int icem_reset_options(void);
const char* icem_option_name(int32_t index);

/* ---- per-kernel timing (measurement only) ------------------------------------------------- */
enum {
    ICEM_K_SAMPLE = 0, ICEM_K_ROLLOUT, ICEM_K_TOPK_PARTIAL, ICEM_K_LOCAL_PACK, ICEM_K_MERGE_REFIT,
    ICEM_K_SAMPLE_ROLLOUT, ICEM_K_COUNT
};
/* While enabled, every kernel launch of this handle is bracketed by hipEventRecord on the
 * caller's stream.  icem_profile_read synchronises those events, returns per kernel class the
 * summed duration [ms], the number of launches and the units processed (traj-steps for
 * SAMPLE/ROLLOUT/SAMPLE_ROLLOUT, keys for the top-k kernels), and clears the log. */
int icem_profile_enable(icem_handle* h, int32_t on);
/* Development aid: [grid, 8] int64 device buffer a kernel under study may fill with per-workgroup phase
 * cycle stamps (NULL disables; no production kernel writes it).  h = NULL addresses the stateless
 * icem_rssm_rollout_cost instead: 16 int64 of wall_clock64 stamps of tile 0 (icem_rssm_split.hip). */
int icem_debug_stamps(icem_handle* h, void* dev_ptr);
int icem_profile_read(icem_handle* h, double* total_ms, int64_t* launches, int64_t* units);
/* What an event pair adds to the kernel it brackets (measurement only; stateless).  Times `reps` event pairs on `stream`
 * around a one-wave kernel that spins for spin_us of the 100 MHz wall clock and reports how long it really ran:
 * *pair_us = median event-pair time, *kernel_us = median in-kernel duration.  pair_us - kernel_us is the part of every
 * icem_profile_read span that is the bracket (dispatch in front, the closing event behind), not the kernel; bench.py
 * subtracts it so that the per-kernel times sum to the step they were taken from. */
int icem_profile_overhead(void* stream, int32_t reps, double spin_us, double* pair_us, double* kernel_us);

/* Byte size / layout of one candidate record: {cost (T), gidx (int32, padded to sizeof(T)),
 * actions[h*d] (T)}; all-gather moves K records per rank. */
size_t icem_record_bytes(const icem_handle* h);

/* Learned-dynamics rollout (BASELINE configs[4]; the reference has no model code for it, README.md:21-29): the
 * recurrent state-space model declared in icem_amd/models.py::declared_rssm, rolled out on its prior mean from
 * obs0 = [h (200) | z (30)] (f32) along actions [n, horizon, 6] (f32) in ONE launch on the bf16 matrix cores
 * (f32 accumulation, f32 recurrent state); costs[i] (f32) = reduce_t -reward(state_t) with cost_mode = ICEM_COST_*.
 * params: the packed bf16 parameter buffer of icem_rssm_param_elems() elements (layout: icem_amd/csrc/icem_rssm.h;
 * packer: icem_amd.models.pack_rssm).  No handle.  Populations of up to 65 536 rows take a launch in which the
 * recurrence and the reward head are separate workgroups exchanging the states through a staging area in device
 * memory (8 KB per 16 trajectories and step: 25 MB up to 4096 rows at h = 12, grown on demand to 403 MB at 65 536); the
 * library keeps one such area per (device, stream), allocated at the
 * first call on that stream -- which therefore must not be inside a stream capture (later ones may be: the launch
 * leaves its flags as it found them).  A reward workgroup whose bounded wait for its recurrence times out reports NaN
 * costs and raises a host-visible word: every later launch on that stream reports NaN too until the NEXT call of this
 * function, which resets the flags, launches nothing and returns ICEM_E_STATE once.  icem_rssm_trim() frees the staging
 * areas (no launch of this path may be in flight). */
size_t icem_rssm_param_elems(void);
int icem_rssm_trim(void);
int icem_rssm_rollout_cost(int32_t n, int32_t horizon, int32_t cost_mode, const void* params, const void* obs0,
                           const void* actions, void* costs, void* stream);

/* MpcICem.get_action (icem/controllers/icem.py:106-189) as ONE call for a host caller: obs_host [obs_dim] float64
 * goes to b->obs0 through a pinned staging buffer of the handle, icem_plan_step runs, and the executed action
 * [act_dim] (+ the best cost of the last pool, if best_cost_host != NULL) comes back as float64 after ONE stream
 * synchronisation.  world == 1, device noise (b->z_* must be NULL). */
int icem_get_action(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, const double* obs_host,
                    double* action_host, double* best_cost_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICEM_HIP_H */
