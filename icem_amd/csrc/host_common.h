// host_common.h -- what the host-side translation units of libicem_hip.so share: the handle, error plumbing,
// per-kernel timing scopes, and the launcher interfaces between them.
//   generic_kernels.hip  the generic (f32 / f64, any shape) kernels + their launchers (gk_*)
//   plan.hip             the fused MPC step: icem_plan_* / icem_get_action and the f32 throughput-path orchestration
//   exchange.hip         the in-library elite exchange between GPUs (icem_exchange_*)
//   abi.hip              handle life cycle and the stateless operators of include/icem_hip.h
// Internal; not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/icem_hip.h"
#include "icem_fused.h"
#include "options.h"

namespace icem {

inline thread_local std::string g_err;

inline int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define ICEM_HIP_TRY(expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return ::icem::fail(ICEM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int WG = 256;          // 4 wavefronts of 64
constexpr int TOPK_CHUNK = 1024; // costs per workgroup in the block-level top-k

struct Exchange;  // exchange.hip

}  // namespace icem

struct icem_handle {
    icem_config cfg;
    int F = 0, HMAX = 0, hd = 0;
    size_t tsize = 4;
    void* W_dev = nullptr;
    int model_kind = 0, obs_dim = 0, O = 0;
    void* A_dev = nullptr;
    void* B_dev = nullptr;
    bool has_model = false, has_cost = false;
    icem_cost_spec cost;
    icem_cost_terms terms;
    bool has_terms = false;  // any term of icem_cost_terms switched on
    void* host_stage = nullptr;  // pinned, device-mapped block of icem_get_action [obs | action, best cost | flag]
    void* host_stage_dev = nullptr;
    unsigned io_seq = 0;
    std::vector<int> pop;
    int n_reuse = 0;
    int n_local_max = 0;
    // optional per-kernel timing with HIP events on the caller's stream (bench.py roofline leg)
    bool profiling = false;
    struct Span {
        int kind;
        long long units;
        hipEvent_t a, b;
    };
    bool use_fast = true;
    long long* dbg = nullptr;
    int fast_lists = 0;  // candidate lists written by the last matrix-pipe rollout (0 = generic path ran)
    int fast_tail_rows = 0;  // ... and shifted-elite rows behind them that the merge scores through the cost array
    // icem_plan_step (world == 1): an iteration's merge can ride in the prologue of the next iteration's launch,
    // which then reads the previous pool / lists / distribution while writing new ones -> ping-pong partners of
    // the caller's actions / workspace buffers and of mean | std, owned by the handle
    void* actions_alt = nullptr;
    void* ws_alt = nullptr;
    float* pp_stats = nullptr;          // [2][2 * hd]
    bool defer_merge = false;           // plan_iter_merge: stash the merge instead of launching it
    bool pm_pending = false;            // a stashed merge waits for the next local launch
    icem::MergeSingleArgs pm_args;
    float* merge_mean_out = nullptr;    // where the next merge writes mean / std (nullptr: in place)
    float* merge_std_out = nullptr;
    // world > 1 with icem_set_merge_deferral(on): the same folding across the split calls; the distribution of the
    // running MPC step lives at cur_mean / cur_std (the caller's buffers or pp_stats)
    bool deferral = false;
    float* cur_mean = nullptr;
    float* cur_std = nullptr;
    // permuted, padded model of the matrix-pipe rollout: column 0 = obs[lin_idx], column 1 = obs[flip_idx]
    bool wide = false;           // obs_dim > 32: the rollout is k_rollout_wide.hip's GEMM kernel (f32 only)
    int Of = 0;                  // O of the 16-trajectory tile kernels (Tile16 / Tile4), or 0 where they do not serve this handle's
                                 // model + cost (wide observations; icem_cost_terms; a cost without its linear term; a shape
                                 // (h, d, O) that is not compiled): the f32 rollout is then the GEMM kernel at ANY width.
                                 // Kept current by update_paths() (abi.hip) behind every model / cost setter.
    // arithmetic of the tile kernels' model step (icem_set_tile_arith): mode -1 = by configuration, 0 = exact f32, 1 = fp16 planes;
    // tile_arith = what the handle's launches use (update_paths: mode + whether Tile16H serves this model)
    int tile_arith_mode = -1;
    int tile_arith = 0;
    int hn_prog[3] = {0, 0, 0};  // ... its term program (N32, N4, NP) and the device copy of the terms sorted into it
    void* hn_cs_dev = nullptr;
    bool hn_tile = false;        // the f32 rollout is k_rollout_hn.hip's TileHN kernel (Door / Relocate / FetchPickAndPlace shapes; fp16 planes)
    float tile_m_scale = 1.f, tile_b_scale = 1.f;   // FastRolloutArgs::m_scale / b_scale (update_paths)
    float act_mag = 1.f;         // max(|low|, |high|) of the action bounds last seen (FastRolloutArgs::act_mag), from ...
    const void* am_lo = nullptr; // ... this (low, high) buffer pair (refreshed by icem_reset_distribution and the plan calls)
    const void* am_hi = nullptr;
    // fp16-plane tiles (Tile16H / TileHN): served only where no state can leave fp16's range inside the horizon
    double tile_growth = 1.0;    // worst case of |state entry| / max(|obs0|, action bound) over the horizon (update_paths; 1 for tanh)
    int tile_ratio_log2 = 0;     // log2(largest |B entry| / largest |A entry|) (the action operand's scale relative to the state's)
    // trajectories whose cost left a tile kernel non-finite (FastRolloutArgs::nonfinite): a device word, and the value
    // icem_get_action saw at its last call
    unsigned* nonfinite_dev = nullptr;
    unsigned nonfinite_seen = 0;
    void* Mw_dev = nullptr;      // its packed model
    void* Mws_dev = nullptr;     // ... as three bf16 planes (k_rollout_wide_split.hip) ...
    void* Mwh_dev = nullptr;     // ... and as two fp16 planes of the model x 2^k, Mwh_inv = 2^-k: the default wide rollout
    float Mwh_inv = 1.f;
    void* Mwh_ksc_dev = nullptr; // ... and the contraction entries' and output columns' powers of two (the model equilibrated): [Mwh_nk | columns]
    int Mwh_nk = 0;
    float Mwh_sbound = 0.f;
    int wide_mode = -1;          // icem_set_wide_arith: ICEM_WIDE_AUTO (-1) / F16X2 (0) / F32 (1) / BF16X3 (2) as asked for ...
    int wide_eff = 0;            // ... and the one in effect (update_paths: AUTO = fp16 planes unless the balanced model is not)
    int wide_imbalance = 0;      // wide_model_imbalance_log2 of the current model
    // generic path, world == 1: the iteration's selection waits for the merge call (gk_select_refit: one launch for top-K +
    // gather + refit); gen_sel_cand > 0 = rows with a cost in b->costs, gen_sel_loc of them sampled rows
    int gen_sel_cand = 0, gen_sel_loc = 0;
    int wide_packed = -2;        // which arithmetic Mw_dev / Mwh_dev / Mws_dev currently hold the model for (-2: none)
                                 // (k_rollout_wide.hip + its row kernel), 2 = bf16 planes (6 products)
    void* wide_cs_dev = nullptr; // CostArgs<float> (cost spec + terms) for k_rollout_wide, refreshed by the cost setters
    void* Mp_dev = nullptr;
    void* perm_dev = nullptr;
    int flip_col = -1;
    bool fast_model_ready = false;
    std::vector<double> A_host, B_host;
    std::vector<Span> spans;
    std::vector<hipEvent_t> free_events;
    icem::Exchange* xchg = nullptr;  // in-library elite exchange (icem_exchange_*), world > 1
    icem::XchgWait xw_last;          // ... and what the merge of the running iteration waits for (set by the push)
    uint64_t episode = 0;            // folded into the noise stream offset (icem_set_episode)
    // sharded runs with merge deferral: an iteration's record pack + push can ride in the NEXT local launch too
    // (workgroup 0 of sample_rollout_kernel) instead of being a launch of its own
    bool pk_pending = false;
    icem::PackPrev pk_args;
    float* pub_dev = nullptr;        // published merge (PackPrev::pub): [2 * hd] floats, then the flag word
    unsigned pub_seq = 0;
    // noise-ahead pipeline (plan.hip::plan_step_ahead; world == 1, large populations): every iteration's launch
    // also draws the raw noise of the NEXT sampling call into the next pool.  Non-last iterations rotate through three
    // pools owned by the handle (a pool is rewritten by the noise role two launches after the merge that read it).
    struct Ahead {
        void* pool[3] = {nullptr, nullptr, nullptr};
        unsigned long long ctr = 0;          // non-last iterations so far: pool of (step, it) = pool[(ctr + it) % 3]
        // the noise drawn ahead for iteration 0 of the next MPC step
        bool next_valid = false;
        uint64_t next_episode = 0;
        int next_step = -1;
        void* next_pool = nullptr;
        hipStream_t next_stream = nullptr;   // the stream that noise was enqueued on: consumed only by a step on the SAME stream
        // host copy of the action bounds (the transform takes them as scalars: equal in every dimension or no pipeline)
        const void* lo_ptr = nullptr;
        const void* hi_ptr = nullptr;
        bool uniform = false;
        float lo = 0.f, hi = 0.f;
        // part of that noise rides beside the step's LAST merge (the one launch that leaves the chip idle)
        bool tail_pending = false;
        icem::FastSampleArgs tail_args, tail2_args;   // tail2 (n > 0): the next step's shifted elites' noise as well
        // ... and the head of the next step's iteration-1 noise (rows [0, next1_rows) of next1_pool): iteration 0's launch has
        // no merge prologue for its noise role to hide behind, its noise role outlasts its rollout (EXPERIMENTS R5.2)
        int next1_rows = 0;
        void* next1_pool = nullptr;
        // small populations (single-launch kernel): the whole first noise of the next MPC step is drawn beside the last
        // merge into `pre_raw` [pop[0] + n_reuse, h, d]; iteration 0 of that step only maps it (FastSampleArgs::raw_src)
        void* pre_raw = nullptr;
        bool pre_valid = false;
        uint64_t pre_episode = 0;
        int pre_step = -1;
        hipStream_t pre_stream = nullptr;    // (as next_stream)
        int disabled = -1;                   // ICEM_NOISE_AHEAD (latched at first use)
        int min_rows = 0;                    // ICEM_NOISE_AHEAD_MIN_ROWS
    } ahead;
    // the step of a small population as one launch inside one XCD (plan.hip::plan_step_xcd; k_step_xcd.hip)
    struct StepXcd {
        void* state = nullptr;         // counters (zero between launches)
        void* raw = nullptr;           // raw noise of iterations 0 .. iters - 1 of the running step: [sum pop][h*d]
        void* pre[2] = {nullptr, nullptr};   // iteration 0's noise of the NEXT step, by the parity of that step
        void* shift = nullptr;         // [16][h*d] rows + [16] costs of the shifted elites
        unsigned long long launches = 0;
        bool pre_valid = false;        // pre[pre_step & 1] holds the noise of (pre_episode, pre_step), enqueued on pre_stream
        uint64_t pre_episode = 0;
        int pre_step = -1;
        hipStream_t pre_stream = nullptr;
        bool disabled = false;         // a bounded wait ran out once (icem_step_status): the handle keeps to the launches per iteration
        int cus = 0;                   // compute units of the device (the kernel is built around 8 XCDs x 32)
    } sx;
    // icem_plan_step_batch (plan.hip): the device array of the batch's argument blocks lives with the batch's FIRST handle
    void* batch_ctx = nullptr;
    void (*batch_ctx_free)(void*) = nullptr;
    unsigned long long batch_uploads = 0;   // how often that array was (re)written (steady state: never)
    void* rccl_comm = nullptr;       // collective.hip: the RCCL communicator of icem_allgather_elites (world > 1)
    bool rccl_owned = false;         // ... created by icem_rccl_connect (destroyed with the handle) or adopted
};

namespace icem {

struct ProfScope {
    icem_handle* h;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    int kind;
    long long units;
    static hipEvent_t get(icem_handle* h) {
        if (!h->free_events.empty()) {
            hipEvent_t e = h->free_events.back();
            h->free_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    ProfScope(const icem_handle* hc, int kind_, long long units_, hipStream_t st_)
        : h(const_cast<icem_handle*>(hc)), st(st_), kind(kind_), units(units_) {
        if (!h->profiling || g_batch.rec) return;
        a = get(h);
        b = get(h);
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        h->spans.push_back({kind, units, a, b});
    }
};

inline int shard_chunk(int n_global, int world) { return (n_global + world - 1) / world; }
inline int topk_blocks(int n) { return (n + TOPK_CHUNK - 1) / TOPK_CHUNK; }

template <typename T>
void split_partial_ws(void* ws, int nblk, int K, T** pc, int** pi) {
    *pc = (T*)ws;
    *pi = (int*)((unsigned char*)ws + (size_t)nblk * K * sizeof(T));
}

inline int check_handle(const icem_handle* h) {
    if (!h) return fail(ICEM_E_INVALID, "null handle");
    return ICEM_OK;
}

// ---- generic_kernels.hip: launchers of the generic kernels; pointers are of the handle's dtype ------------------
int gk_sample(const icem_handle* h, int n, long long first_index, const void* mean, const void* std, const void* low,
              const void* high, const void* zr, const void* zi, uint64_t offset, int t_begin, int row0_mean, void* out,
              hipStream_t st);
int gk_sample_truncnorm(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                        const void* lower, const void* upper, const void* u, uint64_t offset, void* actions, hipStream_t st);
int gk_sample_piecewise(const icem_handle* h, int n, long long call_offset, int change_freq, long long first_block,
                        const void* low, const void* high, const void* u, void* actions, hipStream_t st);
int gk_cem_bounds(const icem_handle* h, int like_levine, const void* mean, void* std, const void* low, const void* high,
                  void* lower, void* upper, hipStream_t st);
int gk_philox_normals(const icem_handle* h, int n, long long first_index, uint64_t offset, void* z_r, void* z_i,
                      hipStream_t st);
int gk_rollout(const icem_handle* h, int n, const void* obs0, const void* actions, void* costs, void* observations,
               hipStream_t st);
int gk_trajectory_cost(const icem_handle* h, int n, int o, const void* obs, const void* nxt, long long ts, long long ss,
                       const void* actions, void* costs, hipStream_t st);
int gk_cost_reduce(const icem_handle* h, int n, const void* step_costs, void* costs, hipStream_t st);
int gk_topk(const icem_handle* h, int n, int K, const void* costs, void* out_c, int* out_i, void* ws, hipStream_t st);
int gk_topk_partial(const icem_handle* h, int n_cand, int K, const void* costs, void* ws, int nblk, hipStream_t st);
int gk_local_pack(const icem_handle* h, int nblk, int K, int n_loc, int shard_lo, int n_global, const void* ws,
                  const void* actions, void* records, hipStream_t st);
int gk_gather_refit(const icem_handle* h, const void* actions, const int32_t* idx, int k, void* mean, void* std,
                    void* elites_out, hipStream_t st);
int gk_shift(const icem_handle* h, void* mean, void* std, const void* low, const void* high, hipStream_t st);
int gk_reset(const icem_handle* h, void* mean, void* std, const void* low, const void* high, hipStream_t st);
int gk_shift_elites(const icem_handle* h, int n_extra, const void* elites, void* dst, hipStream_t st);
// merge of the all-gathered records (+ kept elites): arguments of merge_refit_kernel, dtype-erased
struct MergeArgsV {
    int n_rec, n_keep, K, h, d, n_global, last;
    double alpha, init_std;
    const void* records;
    const void* elites_cur;
    const void* elites_cost_cur;
    void* elites_next;
    void* elites_cost_next;
    const void* mean_in;
    const void* std_in;
    void* mean;
    void* std;
    const void* low;
    const void* high;
    void* executed;
    void* best_cost;
    XchgWait xw;
};
int gk_merge_refit(const icem_handle* h, const MergeArgsV& a, hipStream_t st);
// world == 1: top-K over the pool's costs (+ kept elites) + gather + refit in ONE launch (a.records / a.n_rec unused)
bool gk_select_ok(const icem_handle* h, int n_cand, int n_keep, int K);
int gk_select_refit(const icem_handle* h, int n_cand, int n_loc, const void* costs, const void* actions, const MergeArgsV& a,
                    hipStream_t st);
// every index a cost term reads lies inside an observation of width o (nullptr = fine)
const char* cost_indices_error(const icem_handle* h, int o);

// ---- exchange.hip: the in-library elite exchange --------------------------------------------------------------------
bool xchg_connected(const icem_handle* h);
unsigned xchg_status_peek(const icem_handle* h);  // != 0: a device-side wait for a peer has timed out (not cleared)
// this rank's K records of the running iteration -> every rank's block (one launch); *wait_out: what the merge polls
// every peer runs in a process (stream) of its own: a launch of ours may wait for something a peer launches later
bool xchg_concurrent_peers(const icem_handle* h);
int xchg_push(icem_handle* h, const void* my_records, hipStream_t st, XchgWait* wait_out);
// ... or, where the caller's own kernel does the push (pack_records_kernel): the arguments for it
int xchg_begin(icem_handle* h, XchgPush* push_out, XchgWait* wait_out);
void xchg_destroy(icem_handle* h);

// Observation widths in (32, 384] exist only in k_rollout_wide.hip (f32, built-in HalfCheetah / HumanoidStandup cost form,
// device noise, K <= 32, costs only).  What a wide handle cannot do, spelled out (nullptr = fine) -- asked by
// icem_set_model / icem_set_cost_terms up front and by icem_rollout_cost / check_plan at use.
inline const char* wide_unsupported(const icem_handle* h, int K, bool external_noise, bool want_observations) {
    if (!h->wide) return nullptr;
    if (K > 32) return "obs_dim > 32 needs num_elites <= 32 (candidate lists of k_rollout_wide)";
    if (external_noise) return "obs_dim > 32 has no external-noise (z_r / z_i) path: the wide rollout is f32 with device noise only";
    if (want_observations) return "obs_dim > 32: icem_rollout_cost returns costs only at this width (observations == NULL)";
    return nullptr;
}

// ---- collective.hip: the RCCL form of the elite all-gather ----------------------------------------------------------
bool rccl_connected(const icem_handle* h);
int rccl_allgather_records(icem_handle* h, void* records, hipStream_t st);
void rccl_release(icem_handle* h);

// ---- plan.hip: the f32 throughput path ---------------------------------------------------------------------------
bool fast_rollout_ok(const icem_handle* h, int K);
// the f32 rollout of this handle is k_rollout_wide*.hip's GEMM kernel (obs_dim > 32, or a narrow model the tile kernels do not serve)
inline bool gemm_rollout(const icem_handle* h) { return h->wide || h->Of == 0; }
bool fast_sample_ok(const icem_handle* h);
int launch_fast_rollout(icem_handle* h, int n_rows, int n_cand, int K, const void* obs0, const void* actions,
                        void* costs, float* part_c, int* part_i, hipStream_t st, int* lists_out,
                        unsigned long long* part_k = nullptr, int n_tail = 0, int* tail_out = nullptr);
int refresh_act_mag(icem_handle* h, const void* low, const void* high, hipStream_t st, bool force);   // abi.hip
void ahead_destroy(icem_handle* h);
void predraw_next_step(icem_handle* h, const icem_plan_buffers* b, int mpc_step, hipStream_t st);
int launch_fast_sample(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                       const void* low, const void* high, uint64_t offset, int row0_mean, void* out, hipStream_t st,
                       int n_shift = 0, const void* elites_src = nullptr, uint64_t offset2 = 0);

}  // namespace icem

#define ICEM_DISPATCH(h, expr_f32, expr_f64) ((h)->cfg.dtype == ICEM_F64 ? (expr_f64) : (expr_f32))
