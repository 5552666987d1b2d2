// plan.hip -- the fused MPC step behind icem_plan_iter_local / icem_plan_iter_merge / icem_plan_step / icem_get_action
// (the body of MpcICem.get_action, icem/controllers/icem.py:106-189): which kernels an iteration launches (generic
// path, two-kernel f32 path, single-launch f32 paths), where a merge rides (own launch or the next launch's prologue),
// where a sharded rank's record pack rides (own launch or workgroup 0 of the next local launch), which trailing
// shifted-elite rows go around the candidate lists, and which of the ping-pong buffers each launch reads and writes.  No device code here except the result publisher.
#include "host_common.h"
#include "cost_terms_dev.h"

using namespace icem;

// icem_get_action: executed action + best cost -> the host-mapped block, then the sequence flag (system scope)
__global__ void publish_result_kernel(const float* executed, const float* best_cost, const unsigned* nonfinite, int d, float* host_out,
                                      unsigned* flag, unsigned seq) {
    const int j = threadIdx.x;
    if (j < d) host_out[j] = executed[j];
    if (j == d) host_out[d] = best_cost[0];
    if (j == d + 1) reinterpret_cast<unsigned*>(host_out)[d + 1] = nonfinite[0];   // the handle's status word (icem_nonfinite_costs)
    __threadfence_system();
    __syncthreads();
    if (j == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

namespace icem {

thread_local BatchState g_batch;   // icem_plan_step_batch: launch-shape hint + the recording launchers (icem_fused.h)

// An argument block with EVERY byte defined (padding included): icem_plan_step_batch keeps the blocks of the previous
// same-parity step and compares bytes to decide whether the device copy is still good.  (Members with a non-zero default:
// MergeSingleArgs::keep_base, FastRolloutArgs::act_mag / m_scale / b_scale.)
template <class T>
static T zeroed_args() {
    T t;
    std::memset((void*)&t, 0, sizeof(T));
    return t;
}

// Build the permuted, zero-padded [A ; B] operand of the matrix-pipe rollout (lazily: it depends on
// both icem_set_model and icem_set_cost).  Observation entries are reordered so that the linear cost
// term reads column 0 and the flip term column 0 or 1 -- static registers in the kernel.
int ensure_fast_model(icem_handle* h) {
    if (h->fast_model_ready) return ICEM_OK;
    if (gemm_rollout(h)) {  // the GEMM kernels' model, packed in MFMA operand order (k_rollout_wide.hip, k_rollout_wide_split.hip)
        // ONE packing, the one of the arithmetic in effect (icem_handle::wide_eff): exact f32, two fp16 planes with the
        // equilibrated model's scales, or three bf16 planes.  Everything is uploaded into fresh allocations first and the
        // handle's pointers and scales change together, after the last upload succeeded.
        struct Fresh {
            void* p[4] = {nullptr, nullptr, nullptr, nullptr};
            ~Fresh() { for (void* q : p) if (q) (void)hipFree(q); }
        } fr;
        auto up = [](void** dev, const void* src, size_t bytes) -> int {
            ICEM_HIP_TRY(hipMalloc(dev, bytes));
            ICEM_HIP_TRY(hipMemcpy(*dev, src, bytes, hipMemcpyHostToDevice));
            return ICEM_OK;
        };
        const int eff = h->wide_eff;
        float minv = 1.f, sb = 0.f;
        int nk = 0;
        int rc = ICEM_OK;
        if (eff == 1) {
            std::vector<float> Mw;
            pack_wide_model(h->obs_dim, h->cfg.act_dim, h->A_host.data(), h->B_host.data(), Mw);
            rc = up(&fr.p[0], Mw.data(), Mw.size() * sizeof(float));
        } else {
            std::vector<unsigned short> Mb;
            std::vector<float> ksc, csc;
            pack_wide_model_split(h->obs_dim, h->cfg.act_dim, h->A_host.data(), h->B_host.data(), eff == 2 ? 3 : 2, Mb, &minv, &ksc, &csc, &sb);
            rc = up(&fr.p[0], Mb.data(), Mb.size() * sizeof(unsigned short));
            if (!rc && eff == 0) {   // the scales of the equilibrated model: [ksc | csc] in one allocation
                nk = (int)ksc.size();
                ksc.insert(ksc.end(), csc.begin(), csc.end());
                rc = up(&fr.p[1], ksc.data(), ksc.size() * sizeof(float));
            }
        }
        if (rc) return rc;
        if (h->wide) {   // ... and row-major in f32 for the rows rolled out one by one (rollout_rows_wide_kernel) and TileHN
            std::vector<float> tmp(h->A_host.begin(), h->A_host.end());
            rc = up(&fr.p[2], tmp.data(), tmp.size() * sizeof(float));
            if (rc) return rc;
            tmp.assign(h->B_host.begin(), h->B_host.end());
            rc = up(&fr.p[3], tmp.data(), tmp.size() * sizeof(float));
            if (rc) return rc;
        }
        // commit
        void*& slot = eff == 1 ? h->Mw_dev : (eff == 0 ? h->Mwh_dev : h->Mws_dev);
        for (void** old : {&h->Mw_dev, &h->Mwh_dev, &h->Mws_dev, &h->Mwh_ksc_dev}) {
            if (*old) (void)hipFree(*old);
            *old = nullptr;
        }
        slot = fr.p[0];
        h->Mwh_ksc_dev = fr.p[1];
        h->Mwh_inv = minv;
        h->Mwh_sbound = sb;
        h->Mwh_nk = nk;
        fr.p[0] = fr.p[1] = nullptr;
        if (h->wide) {   // (a narrow model on the GEMM kernel: A_dev / B_dev keep the generic kernels' padded layout)
            if (h->A_dev) (void)hipFree(h->A_dev);
            if (h->B_dev) (void)hipFree(h->B_dev);
            h->A_dev = fr.p[2];
            h->B_dev = fr.p[3];
            fr.p[2] = fr.p[3] = nullptr;
        }
        h->wide_packed = eff;
        h->fast_model_ready = true;
        return ICEM_OK;
    }
    const int O = h->O, o = h->obs_dim, d = h->cfg.act_dim;
    // layout of Tile16 (fused_dev.h): O <= 20 -> one 16-column matrix-pipe tile + extra columns, Mp [O + d + 1, ceil4(O)];
    // O > 20 -> two tiles, observation block padded to 32 rows / columns, Mp [32 + d + 1, 32]
    const bool two = O > 20;
    const int OP = two ? 32 : O;
    const int CT4 = two ? 32 : ((O + 3) / 4) * 4;
    std::vector<int> perm;
    perm.push_back(h->cost.lin_idx);
    h->flip_col = -1;
    if (h->cost.flip_idx >= 0) {
        if (h->cost.flip_idx == h->cost.lin_idx) {
            h->flip_col = 0;
        } else {
            perm.push_back(h->cost.flip_idx);
            h->flip_col = 1;
        }
    }
    for (int k = 0; k < O; ++k)
        if (std::find(perm.begin(), perm.end(), k) == perm.end()) perm.push_back(k);
    std::vector<float> Mp((size_t)(OP + d + 1) * CT4, 0.f);  // + one zero row (contraction slots without an entry)
    auto Aat = [&](int r, int c) { return (r < o && c < o) ? h->A_host[(size_t)r * o + c] : 0.0; };
    auto Bat = [&](int j, int c) { return c < o ? h->B_host[(size_t)j * o + c] : 0.0; };
    for (int k = 0; k < O; ++k)
        for (int c = 0; c < O; ++c) Mp[(size_t)k * CT4 + c] = (float)Aat(perm[k], perm[c]);
    for (int j = 0; j < d; ++j)
        for (int c = 0; c < O; ++c) Mp[(size_t)(OP + j) * CT4 + c] = (float)Bat(j, perm[c]);
    for (int k = 0; k < (int)perm.size(); ++k)
        if (k >= o || perm[k] >= o) perm[k] = 31;  // padding columns start from the zero slot of the staged observation
    perm.resize(32, 31);
    if (h->Mp_dev) (void)hipFree(h->Mp_dev);
    if (h->perm_dev) (void)hipFree(h->perm_dev);
    ICEM_HIP_TRY(hipMalloc(&h->Mp_dev, Mp.size() * sizeof(float)));
    ICEM_HIP_TRY(hipMalloc(&h->perm_dev, perm.size() * sizeof(int)));
    ICEM_HIP_TRY(hipMemcpy(h->Mp_dev, Mp.data(), Mp.size() * sizeof(float), hipMemcpyHostToDevice));
    ICEM_HIP_TRY(hipMemcpy(h->perm_dev, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice));
    h->fast_model_ready = true;
    return ICEM_OK;
}

bool fast_rollout_ok(const icem_handle* h, int K) {
    if (h->cfg.dtype != ICEM_F32 || !h->has_model || !h->has_cost) return false;
    if (h->wide)  // (the only rollout there is at this width: ICEM_DISABLE_FAST does not apply)
        return wide_rollout_supported(h->obs_dim, h->cfg.act_dim, K);
    if (!h->use_fast) return false;
    // o <= 32: the tile kernels where they serve the model + cost (no icem_cost_terms, a linear term, a compiled shape);
    // otherwise the exact-f32 GEMM kernel with its smallest column count -- FetchPickAndPlace (settings/fpp: o = 28, d = 4,
    // a norm cost), Hopper, Reacher, FetchReach ... run at matrix-pipe speed instead of one thread per trajectory
    if (h->Of == 0) return gemm_rollout_supported(h->obs_dim, h->cfg.act_dim, K);
    return fast_rollout_supported(h->cfg.horizon, h->cfg.act_dim, h->Of, K);
}

FastRolloutArgs fast_rollout_args(const icem_handle* h, int n_rows, int n_cand, int K, const void* obs0,
                                  const void* actions, void* costs, float* part_c, int* part_i) {
    FastRolloutArgs a = zeroed_args<FastRolloutArgs>();
    a.n_rows = n_rows;
    a.n_cand = n_cand;
    a.K = K;
    a.o = h->obs_dim;
    a.cost_mode = h->cfg.cost_mode;
    a.Mp = (const float*)h->Mp_dev;
    a.perm = (const int*)h->perm_dev;
    a.obs0 = (const float*)obs0;
    a.ctrl_w = (float)h->cost.ctrl_weight;
    a.lin_w = (float)h->cost.lin_weight;
    a.flip_pen = (float)h->cost.flip_penalty;
    a.flip_th = (float)h->cost.flip_thresh;
    a.flip_col = h->flip_col;
    a.actions = (const float*)actions;
    a.costs = (float*)costs;
    a.part_c = part_c;
    a.part_i = part_i;
    a.part_k = nullptr;
    a.dbg = h->dbg;
    a.arith = h->tile_arith;
    a.act_mag = h->act_mag;
    a.m_scale = h->tile_m_scale;
    a.b_scale = h->tile_b_scale;
    a.nonfinite = h->nonfinite_dev;
    return a;
}

// workgroups (= candidate lists) of the single-launch kernel for this handle, 0 where it has none.  (Its small slabs roll out on
// Tile4, the VALU twin of the EXACT tile; a handle whose tile arithmetic is the fp16 planes gets the kernel's Tile16H
// instantiation instead -- one arithmetic per handle, whatever a launch's row count.)
static int one_launch_lists(const icem_handle* h, int n_rows, int n_tail = 0, int* tail_out = nullptr) {
    if (tail_out) *tail_out = 0;
    const icem_config& c = h->cfg;
    return sample_rollout_lists(c.horizon, c.act_dim, h->Of, c.rng_rounds, n_rows, n_tail, tail_out);
}

// rows -> costs (+ one sorted candidate list per workgroup when K > 0); returns the number of candidate lists
int launch_fast_rollout(icem_handle* h, int n_rows, int n_cand, int K, const void* obs0, const void* actions,
                        void* costs, float* part_c, int* part_i, hipStream_t st, int* lists_out,
                        unsigned long long* part_k, int n_tail, int* tail_out) {
    if (g_batch.rec) {   // (no batched form: icem_plan_step_batch checks its configurations up front; belt and braces)
        g_batch.unsupported = true;
        return fail(ICEM_E_UNSUPPORTED, "icem_plan_step_batch: this configuration's rollout launch has no batched form");
    }
    int rc = ensure_fast_model(h);
    if (rc) return rc;
    if (tail_out) *tail_out = 0;
    if (h->hn_tile) {   // Door / Relocate / FetchPickAndPlace shapes: TileHN (k_rollout_hn.hip)
        // trailing shifted elites that would open a second round of tiles: workgroups of their own, no list (tail_out rows: the
        // caller's merge takes them as extra candidates through the cost array)
        const int tail = (tail_out && n_cand == n_rows && K > 0) ? hn_tail_rows(n_rows, n_tail) : 0;
        FastRolloutArgs a = fast_rollout_args(h, n_rows, tail ? n_rows - tail : n_cand, K, obs0, actions, costs, part_c, part_i);
        a.part_k = part_k;
        a.arith = 1;
        a.list_wgs = tail ? (n_rows - tail) / 16 : 0;
        if (tail) *tail_out = tail;
        const int ld = h->wide ? h->obs_dim : h->O;   // A_dev / B_dev: row-major f32, unpadded at o > 32, padded to O below
        {
            ProfScope prof(h, ICEM_K_ROLLOUT, (long long)n_rows * h->cfg.horizon, st);
            launch_rollout_hn(a, h->cfg.horizon, h->cfg.act_dim, h->obs_dim, h->model_kind, (const float*)h->A_dev, ld, (const float*)h->B_dev, ld,
                              h->cost.lin_idx, h->cost.flip_idx, h->has_terms ? (const CostArgs<float>*)h->hn_cs_dev : nullptr, h->hn_prog, st);
        }
        ICEM_HIP_TRY(hipGetLastError());
        if (lists_out) *lists_out = tail ? a.list_wgs : hn_rollout_lists(n_rows);
        return ICEM_OK;
    }
    if (gemm_rollout(h)) {
        // narrow observations always take the exact-f32 kernel (two workgroup barriers per step buy nothing at o <= 32)
        const bool exact = h->wide_eff == 1;   // (update_paths: asked for, a narrow model, or a width the split kernel does not hold)
        // trailing shifted elites that would open a tile of their own: rolled out row by row (rollout_rows_wide_kernel),
        // scored by the merge through the cost array (tail_out rows; the caller's merge takes them as extra candidates)
        // (exact-f32 tile kernel only: the bf16-split kernel's workgroups take a fifth tile instead)
        const bool split_tail = h->wide && exact && tail_out && n_tail > 0 && n_tail <= 64 && n_cand == n_rows && (n_rows - n_tail) % 16 == 0 &&
                                n_rows - n_tail > 0 && h->cfg.dtype == ICEM_F32;
        if (split_tail) {
            n_rows -= n_tail;
            n_cand = n_rows;
            *tail_out = n_tail;
        }
        WideRolloutArgs w{};
        w.n_rows = n_rows;
        w.n_cand = n_cand;
        w.K = K;
        w.o = h->obs_dim;
        w.d = h->cfg.act_dim;
        w.h = h->cfg.horizon;
        w.kb = exact ? wide_kb(w.o, w.d) : wide_split_kb(w.o, w.d);
        w.xs = exact ? wide_xs(w.o, w.d) : wide_split_xs(w.o, w.d);
        w.cost_mode = h->cfg.cost_mode;
        w.lin_idx = h->cost.lin_idx;
        w.flip_idx = h->cost.flip_idx;
        w.ctrl_w = (float)h->cost.ctrl_weight;
        w.lin_w = (float)h->cost.lin_weight;
        w.flip_pen = (float)h->cost.flip_penalty;
        w.flip_th = (float)h->cost.flip_thresh;
        w.cs = h->has_terms ? (const CostArgs<float>*)h->wide_cs_dev : nullptr;
        w.planes = h->wide_eff == 2 ? 3 : 2;    // (split kernel) fp16 planes, or the bf16 ones (asked for, or AUTO on an unbalanced model)
        w.minv = w.planes == 2 ? h->Mwh_inv : 1.f;
        w.ksc = (const float*)h->Mwh_ksc_dev;
        w.csc = w.ksc ? w.ksc + h->Mwh_nk : nullptr;
        w.sbound = h->Mwh_sbound;
        w.Mp = exact ? (const float*)h->Mw_dev : (const float*)(w.planes == 2 ? h->Mwh_dev : h->Mws_dev);
        w.dbg = h->dbg;
        w.obs0 = (const float*)obs0;
        w.actions = (const float*)actions;
        w.costs = (float*)costs;
        w.part_c = part_c;
        w.part_i = part_i;
        w.part_k = part_k;
        {
            ProfScope prof(h, ICEM_K_ROLLOUT, (long long)n_rows * h->cfg.horizon, st);
            if (exact) launch_rollout_wide(w, h->model_kind, st);
            else launch_rollout_wide_split(w, h->model_kind, st);
            // (the row-wise kernel BESIDE the tile kernel on a second stream instead of behind it was measured: 4.70
            //  instead of 4.15 ms per MPC step -- the three CUs that host a row workgroup finish their tile workgroup
            //  late, and the launch waits for its slowest workgroup; EXPERIMENTS.md R3.7)
            if (split_tail) launch_rollout_rows_wide(w, n_rows, n_tail, (const float*)h->A_dev, (const float*)h->B_dev, h->model_kind, st);
        }
        ICEM_HIP_TRY(hipGetLastError());
        if (lists_out) *lists_out = exact ? wide_rollout_lists(n_rows) : wide_split_lists(n_rows);
        return ICEM_OK;
    }
    FastRolloutArgs a = fast_rollout_args(h, n_rows, n_cand, K, obs0, actions, costs, part_c, part_i);
    a.part_k = part_k;
    const int grid = rollout_lists(h->cfg.horizon, h->cfg.act_dim, h->Of, n_rows);
    {
        ProfScope prof(h, ICEM_K_ROLLOUT, (long long)n_rows * h->cfg.horizon, st);
        launch_rollout16(a, h->cfg.horizon, h->cfg.act_dim, h->Of, h->model_kind, st);
    }
    ICEM_HIP_TRY(hipGetLastError());
    if (lists_out) *lists_out = grid;
    return ICEM_OK;
}

bool fast_sample_ok(const icem_handle* h) {
    return h->use_fast && h->cfg.dtype == ICEM_F32 && fast_sample_supported(h->cfg.horizon, h->cfg.act_dim);
}

FastSampleArgs fast_sample_args(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                                const void* low, const void* high, uint64_t offset, int row0_mean, void* out,
                                int n_shift, const void* elites_src, uint64_t offset2) {
    FastSampleArgs a = zeroed_args<FastSampleArgs>();
    a.n = n;
    a.h = h->cfg.horizon;
    a.d = h->cfg.act_dim;
    a.first_index = first_index;
    a.W = (const float*)h->W_dev;
    a.mean = (const float*)mean;
    a.std = (const float*)std;
    a.low = (const float*)low;
    a.high = (const float*)high;
    a.seed_lo = (uint32_t)h->cfg.seed;
    a.seed_hi = (uint32_t)(h->cfg.seed >> 32);
    a.off_lo = (uint32_t)offset;
    a.off_hi = (uint32_t)(offset >> 32);
    a.row0_mean = row0_mean;
    a.out = (float*)out;
    a.n_shift = n_shift;
    a.elites_src = (const float*)elites_src;
    a.off2_lo = (uint32_t)offset2;
    a.off2_hi = (uint32_t)(offset2 >> 32);
    a.white = h->cfg.noise_beta <= 0 ? 1 : 0;
    a.raw_src = nullptr;
    return a;
}

int launch_fast_sample(const icem_handle* h, int n, long long first_index, const void* mean, const void* std,
                       const void* low, const void* high, uint64_t offset, int row0_mean, void* out, hipStream_t st,
                       int n_shift, const void* elites_src, uint64_t offset2) {
    if (n <= 0 && n_shift <= 0) return ICEM_OK;
    if (g_batch.rec) {
        g_batch.unsupported = true;
        return fail(ICEM_E_UNSUPPORTED, "icem_plan_step_batch: this configuration's sampling launch has no batched form");
    }
    const FastSampleArgs a = fast_sample_args(h, n, first_index, mean, std, low, high, offset, row0_mean, out, n_shift,
                                              elites_src, offset2);
    {
        ProfScope prof(h, ICEM_K_SAMPLE, (long long)n * a.h, st);
        launch_sample_folded(a, h->cfg.rng_rounds, st);
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

// Can the f32 launch of an iteration with n_rows local rows (no shifted elites) carry a merge in its prologue?
// (single-launch kernel with <= 4 rollout waves, or the sampler of the two-kernel path)
bool prologue_possible(const icem_handle* h, int n_rows) {
    const icem_config& c = h->cfg;
    const int K = c.num_elites;
    if (c.dtype != ICEM_F32 || !fast_rollout_ok(h, K) || !fast_sample_ok(h) || n_rows <= 0) return false;
    if (one_launch_lists(h, n_rows) > 0)
        return sample_rollout_merge_ok(c.horizon, c.act_dim, h->Of, c.rng_rounds, n_rows, K);
    return sample_folded_merge_ok(c.horizon, c.act_dim, c.rng_rounds, K);
}

// a stashed merge that found no launch to ride in
int launch_pending_merge(icem_handle* h, hipStream_t st) {
    if (g_batch.rec) {
        g_batch.unsupported = true;
        return fail(ICEM_E_UNSUPPORTED, "icem_plan_step_batch: a merge found no launch to ride in");
    }
    const MergeSingleArgs& m = h->pm_args;
    if (m.records == nullptr) {
        ProfScope prof(h, ICEM_K_MERGE_REFIT, m.n_lists * m.K + m.n_keep, st);
        launch_merge_single(m, st);
    } else {
        MergeArgsV a{};
        a.n_rec = m.n_rec;
        a.n_keep = m.n_keep;
        a.K = m.K;
        a.h = m.h;
        a.d = m.d;
        a.n_global = m.n_global;
        a.last = 0;
        a.alpha = m.alpha;
        a.init_std = m.init_std;
        a.records = m.records;
        a.elites_cur = m.elites_cur;
        a.elites_cost_cur = m.elites_cost_cur;
        a.elites_next = m.elites_next;
        a.elites_cost_next = m.elites_cost_next;
        a.mean_in = m.mean;
        a.std_in = m.std;
        a.mean = m.mean_out;
        a.std = m.std_out;
        a.low = m.low;
        a.high = m.high;
        a.executed = m.executed;
        a.best_cost = m.best_cost;
        a.xw = m.xw;
        return gk_merge_refit(h, a, st);
    }
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

// will the merge-prologue launch of an iteration with n_rows local rows take the previous iteration's pack along?
static bool next_launch_takes_pack(const icem_handle* h, int n_rows) {
    const icem_config& c = h->cfg;
    if (one_launch_lists(h, n_rows) > 0)
        return sample_rollout_pack_ok(c.horizon, c.act_dim, h->Of, c.rng_rounds, n_rows, c.num_elites);
    return sample_folded_pack_ok(c.horizon, c.act_dim, c.rng_rounds, c.num_elites);
}

// a stashed pack that found no launch to ride in
int launch_pending_pack(icem_handle* h, hipStream_t st) {
    const PackPrev& pp = h->pk_args;
    MergeSingleArgs pk{};
    pk.n_lists = pp.n_lists;
    pk.n_pool = pp.n_pool;
    pk.n_global = pp.n_global;
    pk.K = pp.K;
    pk.h = h->cfg.horizon;
    pk.d = h->cfg.act_dim;
    pk.part_k = pp.part_k;
    pk.actions = pp.actions;
    pk.n_keep = pp.n_keep;
    pk.elites_cost_cur = pp.keep_costs;
    pk.keep_base = pp.n_loc;
    {
        ProfScope prof(h, ICEM_K_LOCAL_PACK, pp.n_lists * pp.K, st);
        launch_pack_records(pk, pp.n_loc, pp.shard_lo, pp.records, st, pp.px);
    }
    ICEM_HIP_TRY(hipGetLastError());
    h->pk_pending = false;
    return ICEM_OK;
}

// rows of this rank's shard at iteration `it`
static int local_rows(const icem_handle* h, int it) {
    const int n_global = h->pop[it];
    const int chunk = shard_chunk(n_global, h->cfg.world);
    const int lo = std::min(n_global, h->cfg.rank * chunk);
    return std::max(0, std::min(n_global - lo, chunk));
}

template <typename T>
int plan_iter_local_t(icem_handle* h, const icem_plan_buffers* b, int mpc_step, int it, hipStream_t st) {
    const icem_config& c = h->cfg;
    const int hd = h->hd, K = c.num_elites;
    const int n_global = h->pop[it];
    const int chunk = shard_chunk(n_global, c.world);
    const int lo = std::min(n_global, c.rank * chunk);
    const int n_loc = std::max(0, std::min(n_global - lo, chunk));
    // noise stream offset of this call: episode in the high word (icem_set_episode; the reference's np.random stream
    // runs on across episodes, icem.py:73), sampling call number of the episode in the low one
    const uint64_t call_base = (h->episode << 32) + (uint64_t)mpc_step * (uint64_t)(c.opt_iters + 1);
    const bool last = it == c.opt_iters - 1;
    T* actions = (T*)b->actions;
    // shifted elites, simulated at iteration 0 of every MPC step but the first (icem.py:131-137)
    int n_extra = 0;
    const T* shift_src = nullptr;
    bool shift_in_sampler = false;
    if (it == 0 && c.shift_elites && mpc_step > 0 && h->n_reuse > 0) {
        n_extra = h->n_reuse;
        const int g = (int)(((long long)mpc_step * c.opt_iters) & 1);  // elite buffer holding the previous step's set
        shift_src = (const T*)b->elites + (size_t)g * K * hd;
        // the fast sampler prepares them in an extra workgroup of its own launch
        shift_in_sampler = std::is_same<T, float>::value && b->z_r == nullptr && b->z_r_shift == nullptr &&
                           fast_rollout_ok(h, K) && fast_sample_ok(h) &&
                           n_extra * c.act_dim <= 256;
        if (!shift_in_sampler) {
            T* dst = actions + (size_t)n_loc * hd;
            int rc = gk_shift_elites(h, n_extra, shift_src, dst, st);
            if (rc) return rc;
            rc = gk_sample(h, n_extra, 0, b->mean, b->std, b->low, b->high, b->z_r_shift, b->z_i_shift,
                           call_base + (uint64_t)c.opt_iters, c.horizon - 1, 0, dst, st);
            if (rc) return rc;
        }
    }
    // candidates: the shard, plus the shifted elites on rank 0 only (they are replicated)
    const int n_cand = n_loc + (c.rank == 0 ? n_extra : 0);
    const int row0 = (last && c.use_mean_actions) ? 1 : 0;
    T* rec = (T*)b->records + (size_t)c.rank * K * (hd + 2);
    h->fast_lists = 0;
    h->gen_sel_cand = 0;
    if constexpr (std::is_same<T, float>::value) {
        if (fast_rollout_ok(h, K)) {
            // f32 throughput path.  External white noise (b->z_r: the reference's own draws, tests/golden) takes it too: the
            // generic sampler turns the caller's z into the pool (it is the only kernel that reads z), and from there on the
            // tile rollout, the lists and the threshold merges are the ones device noise gets -- the reference's draws reach
            // Tile16 / Tile16H and merge_select*, not only the generic kernels.
            const bool ext_z = b->z_r != nullptr;
            const uint64_t off = call_base + (uint64_t)it;
            int rc = ensure_fast_model(h);
            if (rc) return rc;
            int lists = 0;
            float* pc;
            int* pi;
            const int n_rows = n_loc + n_extra;
            int tail_rows = 0;  // shifted-elite rows scored through the cost array instead of a list (world 1 only)
            const int one = (!ext_z && fast_sample_ok(h) && (n_extra == 0 || shift_in_sampler))
                                ? one_launch_lists(h, n_rows, shift_in_sampler ? n_extra : 0, &tail_rows) : 0;
            h->fast_tail_rows = one > 0 ? tail_rows : 0;
            // the merge finds the lists' indices behind `lists * K` costs
            split_partial_ws<float>(b->workspace, one > 0 ? one : rollout_lists(c.horizon, c.act_dim, h->Of, n_rows), K, &pc, &pi);
            bool prologue = false, ride = false;
            if (h->pm_pending) {
                prologue = !ext_z && n_extra == 0 && prologue_possible(h, n_rows);
                // a stashed pack rides with the merge whose records it produces, or runs now -- in front of that merge
                ride = prologue && h->pk_pending && next_launch_takes_pack(h, n_rows);
                if (h->pk_pending && !ride) {
                    rc = launch_pending_pack(h, st);
                    if (rc) return rc;
                }
                if (!prologue) {  // cannot ride along after all: run it now
                    rc = launch_pending_merge(h, st);
                    if (rc) return rc;
                }
                h->pm_pending = false;
            } else if (h->pk_pending) {
                rc = launch_pending_pack(h, st);
                if (rc) return rc;
            }
            h->pk_pending = false;
            if (one > 0) {
                // small populations: sample + rollout + top-K in one launch
                FastIterArgs fa = zeroed_args<FastIterArgs>();
                fa.m.keep_base = -1;
                if (prologue) fa.m = h->pm_args;
                if (ride) fa.p = h->pk_args;
                fa.s = fast_sample_args(h, n_loc, lo, b->mean, b->std, b->low, b->high, off, row0, actions,
                                        shift_in_sampler ? n_extra : 0, shift_src, call_base + (uint64_t)c.opt_iters);
                if (it == 0 && c.world == 1 && h->ahead.pre_valid) {
                    // this step's first noise (and the shifted elites') was drawn beside the previous step's last merge -- on
                    // pre_stream: a step enqueued on another stream is not ordered behind that launch and redraws instead
                    if (h->ahead.pre_episode == h->episode && h->ahead.pre_step == mpc_step && h->ahead.pre_stream == st) fa.s.raw_src = (const float*)h->ahead.pre_raw;
                    h->ahead.pre_valid = false;
                }
                fa.r = fast_rollout_args(h, n_rows, n_cand, K, b->obs0, actions, b->costs, pc, pi);
                fa.r.part_k = (unsigned long long*)b->workspace;  // read by merge_single_kernel / pack_records_kernel
                fa.r.list_wgs = tail_rows > 0 ? one : 0;
                {
                    ProfScope prof(h, ICEM_K_SAMPLE_ROLLOUT, (long long)n_rows * c.horizon, st);
                    launch_sample_rollout(fa, c.horizon, c.act_dim, h->Of, h->model_kind, prologue, st);
                }
                ICEM_HIP_TRY(hipGetLastError());
                lists = one;
            }
            if (one == 0) {
                if (prologue) {
                    FastSampleMergeArgs sm;
                    sm.s = fast_sample_args(h, n_loc, lo, b->mean, b->std, b->low, b->high, off, row0, actions, 0, nullptr, 0);
                    sm.m = h->pm_args;
                    if (ride) {
                        sm.p = h->pk_args;
                        // published merge: workgroup 0 merges the records once and publishes mean | std to the rest
                        const int pub_on = opt_i(OPT_PUBLISHED_MERGE);
                        if (pub_on && sm.m.records) {
                            if (!h->pub_dev) {
                                ICEM_HIP_TRY(hipMalloc((void**)&h->pub_dev, ((size_t)2 * hd + 16) * sizeof(float)));
                                ICEM_HIP_TRY(hipMemsetAsync(h->pub_dev, 0, ((size_t)2 * hd + 16) * sizeof(float), st));
                            }
                            sm.p.pub = h->pub_dev;
                            sm.p.pub_flag = reinterpret_cast<unsigned*>(h->pub_dev + 2 * hd);
                            sm.p.pub_seq = ++h->pub_seq;
                        }
                    }
                    {
                        ProfScope prof(h, ICEM_K_SAMPLE, (long long)n_loc * c.horizon, st);
                        launch_sample_folded_merge(sm, st);
                    }
                    ICEM_HIP_TRY(hipGetLastError());
                    rc = ICEM_OK;
                } else if (ext_z) {
                    rc = gk_sample(h, n_loc, lo, b->mean, b->std, b->low, b->high, b->z_r, b->z_i, off, 0, row0, actions, st);
                } else if (fast_sample_ok(h)) {
                    rc = launch_fast_sample(h, n_loc, lo, b->mean, b->std, b->low, b->high, off, row0, actions, st,
                                            shift_in_sampler ? n_extra : 0, shift_src, call_base + (uint64_t)c.opt_iters);
                } else {
                    rc = gk_sample(h, n_loc, lo, b->mean, b->std, b->low, b->high, nullptr, nullptr, off, 0, row0, actions, st);
                }
                if (rc) return rc;
                int tail2 = 0;
                rc = launch_fast_rollout(h, n_rows, n_cand, K, b->obs0, actions, b->costs, pc, pi, st, &lists,
                                         (unsigned long long*)b->workspace, (c.world == 1 && it == 0) ? n_extra : 0, &tail2);
                if (rc) return rc;
                h->fast_tail_rows = tail2;
            }
            h->fast_lists = lists;
            if (c.world > 1) {
                // this rank's K best -> records for the all-gather (same selection code as the merge)
                MergeSingleArgs pk{};
                pk.n_lists = lists;
                pk.n_keep = 0;
                pk.n_pool = n_rows;
                pk.n_global = n_global;
                pk.K = K;
                pk.h = c.horizon;
                pk.d = c.act_dim;
                pk.part_k = (const unsigned long long*)b->workspace;
                pk.actions = (const float*)actions;
                if (one > 0 && tail_rows > 0 && c.rank == 0) {
                    // rank 0's shifted elites sit behind the list-writing workgroups: extra candidates of the pack, costs
                    // from the cost array, key index = their local pool row
                    pk.n_keep = tail_rows;
                    pk.elites_cost_cur = (const float*)b->costs + n_loc;
                    pk.keep_base = n_loc;
                }
                XchgPush px;  // in-library exchange: the pack kernel pushes the records itself where they fit its LDS
                const bool fold_push = xchg_connected(h) && pack_can_push(K, c.horizon, c.act_dim);
                if (fold_push) {
                    rc = xchg_begin(h, &px, &h->xw_last);
                    if (rc) return rc;
                }
                // Riding pack: where this iteration's merge will ride in the next local launch (h->deferral, same
                // conditions as icem_plan_iter_merge's fold) and that launch is a single-launch kernel, the pack rides
                // there too, as its workgroup 0.  Stashed; the next icem_plan_iter_local consumes it (or launches it).
                // (the step's LAST pack: stashed for the merge that waits for its records -- pack_merge_kernel, one launch)
                const int fuse_last = opt_i(OPT_PACK_MERGE);
                const bool rides_next = !last && it + 1 < c.opt_iters;
                if (fold_push && xchg_concurrent_peers(h) && h->deferral && (rides_next || (last && fuse_last)) && lists > 0 && c.world * K <= 128 && K <= 32) {
                    const int n_next = rides_next ? local_rows(h, it + 1) : 0;
                    if (!rides_next || (prologue_possible(h, n_next) && next_launch_takes_pack(h, n_next))) {
                        PackPrev& pp = h->pk_args;
                        pp.part_k = pk.part_k;
                        pp.actions = pk.actions;
                        pp.n_lists = pk.n_lists;
                        pp.n_pool = pk.n_pool;
                        pp.n_global = pk.n_global;
                        pp.K = K;
                        pp.n_loc = n_loc;
                        pp.shard_lo = lo;
                        pp.n_keep = pk.n_keep;
                        pp.keep_costs = pk.elites_cost_cur;
                        pp.records = (float*)rec;
                        pp.px = px;
                        h->pk_pending = true;
                        return ICEM_OK;
                    }
                }
                {
                    ProfScope prof(h, ICEM_K_LOCAL_PACK, lists * K, st);
                    launch_pack_records(pk, n_loc, lo, (float*)rec, st, px);
                }
                ICEM_HIP_TRY(hipGetLastError());
                if (xchg_connected(h) && !fold_push) return xchg_push(h, rec, st, &h->xw_last);
            }
            ICEM_HIP_TRY(hipGetLastError());
            return ICEM_OK;
        }
    }
    // generic path (f64, external noise, shapes outside the fast list): one kernel per stage
    int rc = gk_sample(h, n_loc, lo, b->mean, b->std, b->low, b->high, b->z_r, b->z_i, call_base + (uint64_t)it, 0, row0,
                       actions, st);
    if (rc) return rc;
    rc = gk_rollout(h, n_loc + n_extra, b->obs0, actions, b->costs, nullptr, st);
    if (rc) return rc;
    h->gen_sel_cand = 0;
    if (c.world == 1 && gk_select_ok(h, n_cand, h->n_reuse, K)) {
        // one GPU: nothing to exchange -- the merge call selects straight from the cost array (select_refit_kernel)
        h->gen_sel_cand = n_cand;
        h->gen_sel_loc = n_loc;
        return ICEM_OK;
    }
    const int nblk = std::max(1, topk_blocks(n_cand));
    rc = gk_topk_partial(h, n_cand, K, b->costs, b->workspace, nblk, st);
    if (rc) return rc;
    rc = gk_local_pack(h, nblk, K, n_loc, lo, n_global, b->workspace, actions, rec, st);
    if (rc) return rc;
    if (c.world > 1 && xchg_connected(h)) return xchg_push(h, rec, st, &h->xw_last);
    return ICEM_OK;
}

template <typename T>
int plan_iter_merge_t(icem_handle* h, const icem_plan_buffers* b, int mpc_step, int it, hipStream_t st) {
    const icem_config& c = h->cfg;
    const int hd = h->hd, K = c.num_elites;
    const long long g = (long long)mpc_step * c.opt_iters + it;  // global iteration number
    const int cur = (int)(g & 1), nxt = cur ^ 1;
    T* el = (T*)b->elites;
    T* elc = el + (size_t)2 * K * hd;
    if constexpr (std::is_same<T, float>::value) {
        if (c.world == 1 && h->fast_lists > 0) {
            MergeSingleArgs m = zeroed_args<MergeSingleArgs>();
            m.keep_base = -1;
            m.n_lists = h->fast_lists;
            m.n_keep = (it > 0 && c.keep_previous_elites) ? h->n_reuse : 0;
            const int n_extra = (it == 0 && c.shift_elites && mpc_step > 0) ? h->n_reuse : 0;
            // iteration 0: shifted elites that sit behind the list-writing workgroups (sample_rollout_lists) come in
            // through the kept-elite slot: costs from the cost array, rows from the pool (index n_global + e < n_pool)
            const bool tail = it == 0 && h->fast_tail_rows > 0;
            if (tail) m.n_keep = h->fast_tail_rows;
            m.n_pool = h->pop[it] + n_extra;
            m.n_global = h->pop[it];
            m.K = K;
            m.h = c.horizon;
            m.d = c.act_dim;
            m.last = it == c.opt_iters - 1;
            m.alpha = (float)c.alpha;
            m.init_std = (float)c.init_std;
            m.part_k = (const unsigned long long*)b->workspace;
            m.records = nullptr;
            m.n_rec = 0;
            m.actions = (const float*)b->actions;
            m.elites_cur = (const float*)el + (size_t)cur * K * hd;
            m.elites_cost_cur = tail ? (const float*)b->costs + h->pop[it] : (const float*)elc + (size_t)cur * K;
            m.elites_next = (float*)el + (size_t)nxt * K * hd;
            m.elites_cost_next = (float*)elc + (size_t)nxt * K;
            m.mean = (const float*)b->mean;
            m.std = (const float*)b->std;
            m.mean_out = h->merge_mean_out ? h->merge_mean_out : (float*)b->mean;
            m.std_out = h->merge_std_out ? h->merge_std_out : (float*)b->std;
            m.low = (const float*)b->low;
            m.high = (const float*)b->high;
            m.executed = (float*)b->executed;
            m.best_cost = (float*)b->best_cost;
            m.dbg = h->dbg;
            if (h->defer_merge && !m.last) {  // rides in the next iteration's launch (icem_plan_step)
                h->pm_args = m;
                h->pm_pending = true;
                return ICEM_OK;
            }
            if (h->pk_pending) {  // (cannot happen at world == 1; kept symmetrical)
                const int rc = launch_pending_pack(h, st);
                if (rc) return rc;
            }
            ProfScope prof(h, ICEM_K_MERGE_REFIT, h->fast_lists * K + m.n_keep, st);
            if (h->ahead.tail_pending && merge_noise_ok(m, c.rng_rounds)) {
                launch_merge_noise(m, h->ahead.tail_args, h->ahead.tail2_args, st);  // + (the rest of) the next step's first noise
                h->ahead.tail_pending = false;
            } else {
                launch_merge_single(m, st);
            }
            ICEM_HIP_TRY(hipGetLastError());
            return ICEM_OK;
        }
    }
    MergeArgsV a{};
    a.n_rec = c.world * K;
    a.n_keep = (it > 0 && c.keep_previous_elites) ? h->n_reuse : 0;
    a.K = K;
    a.h = c.horizon;
    a.d = c.act_dim;
    a.n_global = h->pop[it];
    a.last = it == c.opt_iters - 1;
    a.alpha = c.alpha;
    a.init_std = c.init_std;
    a.records = b->records;
    if (c.world > 1 && xchg_connected(h)) {  // the records of this iteration arrive in the exchange block
        a.xw = h->xw_last;
        a.records = h->xw_last.records;
    }
    a.elites_cur = el + (size_t)cur * K * hd;
    a.elites_cost_cur = elc + (size_t)cur * K;
    a.elites_next = el + (size_t)nxt * K * hd;
    a.elites_cost_next = elc + (size_t)nxt * K;
    a.mean_in = b->mean;
    a.std_in = b->std;
    a.mean = h->merge_mean_out ? (void*)h->merge_mean_out : b->mean;
    a.std = h->merge_std_out ? (void*)h->merge_std_out : b->std;
    a.low = b->low;
    a.high = b->high;
    a.executed = b->executed;
    a.best_cost = b->best_cost;
    if constexpr (std::is_same<T, float>::value) {
        const bool fast_records = h->fast_lists > 0 && a.n_rec <= 128 && K <= 32;
        if (fast_records) {  // f32 throughput path: the selection / refit code of the single-GPU merge on the records
            MergeSingleArgs m{};
            m.n_lists = 0;
            m.n_keep = a.n_keep;
            m.n_pool = 0;
            m.n_global = a.n_global;
            m.K = K;
            m.h = a.h;
            m.d = a.d;
            m.last = 0;
            m.alpha = (float)a.alpha;
            m.init_std = (float)a.init_std;
            m.part_k = nullptr;
            m.records = (const float*)a.records;
            m.xw = a.xw;
            m.n_rec = a.n_rec;
            m.actions = nullptr;
            m.elites_cur = (const float*)a.elites_cur;
            m.elites_cost_cur = (const float*)a.elites_cost_cur;
            m.elites_next = (float*)a.elites_next;
            m.elites_cost_next = (float*)a.elites_cost_next;
            m.mean = (const float*)a.mean_in;
            m.std = (const float*)a.std_in;
            m.mean_out = (float*)a.mean;
            m.std_out = (float*)a.std;
            m.low = (const float*)a.low;
            m.high = (const float*)a.high;
            m.executed = (float*)a.executed;
            m.best_cost = (float*)a.best_cost;
            m.dbg = nullptr;
            if (h->defer_merge && !a.last) {  // rides in the next iteration's launch
                h->pm_args = m;
                h->pm_pending = true;
                return ICEM_OK;
            }
            m.last = a.last;
            const int fuse_on = opt_i(OPT_PACK_MERGE);
            if (h->pk_pending && fuse_on && m.records != nullptr && xchg_connected(h)) {
                // the merge runs now, and so must the pack whose records it waits for: ONE launch for the two
                const PackPrev& pp = h->pk_args;
                MergeSingleArgs pk{};
                pk.n_lists = pp.n_lists;
                pk.n_pool = pp.n_pool;
                pk.n_global = pp.n_global;
                pk.K = pp.K;
                pk.h = h->cfg.horizon;
                pk.d = h->cfg.act_dim;
                pk.part_k = pp.part_k;
                pk.actions = pp.actions;
                pk.n_keep = pp.n_keep;
                pk.elites_cost_cur = pp.keep_costs;
                pk.keep_base = pp.n_loc;
                ProfScope prof(h, ICEM_K_MERGE_REFIT, a.n_rec + a.n_keep, st);
                launch_pack_merge(pk, pp.n_loc, pp.shard_lo, pp.records, pp.px, m, st);
                ICEM_HIP_TRY(hipGetLastError());
                h->pk_pending = false;
                return ICEM_OK;
            }
            if (h->pk_pending) {  // (records gathered by a collective between the two: separate launches)
                const int rc = launch_pending_pack(h, st);
                if (rc) return rc;
            }
            ProfScope prof(h, ICEM_K_MERGE_REFIT, a.n_rec + a.n_keep, st);
            launch_merge_single(m, st);
            ICEM_HIP_TRY(hipGetLastError());
            return ICEM_OK;
        }
    }
    if (h->pk_pending) {
        const int rc = launch_pending_pack(h, st);
        if (rc) return rc;
    }
    if (c.world == 1 && h->gen_sel_cand > 0 && h->fast_lists == 0) {
        const int n_cand = h->gen_sel_cand;
        h->gen_sel_cand = 0;
        return gk_select_refit(h, n_cand, h->gen_sel_loc, b->costs, b->actions, a, st);
    }
    return gk_merge_refit(h, a, st);
}


// ---------------------------------------------------------------------------------------------------------------------
// Noise-ahead pipeline (world == 1, f32, device noise, every iteration's rollout launch >= 4 waves per workgroup)
// ---------------------------------------------------------------------------------------------------------------------
// icem.py:73-79 draws the colored noise first and applies `* std + mean` afterwards: only the affine map depends on the
// previous iteration.  So an iteration is ONE launch (iter_ahead_kernel, k_rollout_ahead.hip): its rollout workgroups run
// the previous iteration's merge in their prologue, map the raw noise of the pool to actions as they load it, roll out
// and emit candidate lists; its noise workgroups draw the NEXT sampling call's noise (iteration i + 1, or iteration 0 of
// the next MPC step) into the next pool beside them; at iteration 0 one more workgroup builds and rolls out the shifted
// elites (icem.py:131-137), which reach the merge through the cost array as in the single-launch kernel's tail shape.
// One stream, six launches per MPC step at five iterations (ten on the default path).  Buffers: the last iteration's
// pool is the caller's `actions`; the others rotate through three pools of the handle.
// The first form of this pipeline drew the noise on a second stream: every cross-stream event wait cost 10-15 us on the
// critical path and a low-priority side stream starved the rollouts (profiles/r03_noise_ahead_*; EXPERIMENTS.md).

void ahead_destroy(icem_handle* h) {
    icem_handle::Ahead& A = h->ahead;
    for (void*& p : A.pool) {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    if (A.pre_raw) (void)hipFree(A.pre_raw);
    A = icem_handle::Ahead();
}

static bool ahead_eligible(icem_handle* h, const icem_plan_buffers* b, bool sharded = false) {
    icem_handle::Ahead& A = h->ahead;
    const icem_config& c = h->cfg;
    if (A.disabled < 0) {
        // on by default where it applies (measured 1.09-1.21x the sampler + rollout pair from N = 32 768 to 262 144;
        // option noise_ahead = 0 switches it off -- latched per handle at its first step: the equivalence test and
        // tools/ahead_bench.py flip it between planners)
        A.disabled = opt_i(OPT_NOISE_AHEAD) == 0 ? 1 : 0;
        A.min_rows = opt_i(OPT_NOISE_AHEAD_MIN_ROWS);
    }
    if (A.disabled || (c.world != 1) != sharded || c.dtype != ICEM_F32 || b->z_r != nullptr || !h->use_fast || gemm_rollout(h) ||
        c.opt_iters < 2 || c.rng_rounds != 10)
        return false;
    const int stamps_on = opt_i(OPT_AHEAD_STAMPS);
    if (h->dbg != nullptr && !stamps_on) return false;   // (another kernel is under study; option ahead_stamps = 1: this path's phase stamps)
    if (!fast_rollout_ok(h, c.num_elites) || !fast_sample_ok(h)) return false;
    if (sharded) {
        // peers in processes of their own (the pack rides in the next launch), records that fit the pack's LDS stage and
        // the records merge's two-per-lane layout, the published merge switched on
        const int on = opt_i(OPT_NOISE_AHEAD_SHARDED);
        if (!on || !xchg_connected(h) || !xchg_concurrent_peers(h) || !pack_can_push(c.num_elites, c.horizon, c.act_dim) ||
            c.world * c.num_elites > 128 || c.num_elites > 32)
            return false;
    }
    for (size_t it = 0; it < h->pop.size(); ++it) {
        const int n = sharded ? local_rows(h, (int)it) : h->pop[it];
        if (n < A.min_rows || !rollout_ahead_ok(c.horizon, c.act_dim, h->Of, c.num_elites, n)) return false;
    }
    if (c.shift_elites && h->n_reuse > 16) return false;  // (the shift role rolls its rows out as one 16-row tile)
    // the transform takes the bounds as two scalars: fetch them once per (low, high) buffer pair
    if (A.lo_ptr != b->low || A.hi_ptr != b->high) {
        std::vector<float> lo(c.act_dim), hi(c.act_dim);
        if (hipMemcpy(lo.data(), b->low, lo.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hi.data(), b->high, hi.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        A.lo_ptr = b->low;
        A.hi_ptr = b->high;
        A.uniform = true;
        for (int j = 1; j < c.act_dim; ++j) A.uniform = A.uniform && lo[j] == lo[0] && hi[j] == hi[0];
        A.lo = lo[0];
        A.hi = hi[0];
    }
    return A.uniform;
}

static int ahead_setup(icem_handle* h) {
    icem_handle::Ahead& A = h->ahead;
    if (A.pool[0]) return ICEM_OK;
    for (void*& p : A.pool) ICEM_HIP_TRY(hipMalloc(&p, icem_plan_buffer_bytes(h, ICEM_BUF_ACTIONS)));
    if (!h->ws_alt) ICEM_HIP_TRY(hipMalloc(&h->ws_alt, icem_plan_buffer_bytes(h, ICEM_BUF_WORKSPACE)));
    if (!h->pp_stats) ICEM_HIP_TRY(hipMalloc((void**)&h->pp_stats, (size_t)4 * h->hd * sizeof(float)));
    return ICEM_OK;
}

static int plan_step_ahead(icem_handle* h, const icem_plan_buffers* b, int mpc_step, hipStream_t st) {
    icem_handle::Ahead& A = h->ahead;
    const icem_config& c = h->cfg;
    const int iters = c.opt_iters, K = c.num_elites, hd = h->hd;
    int rc = ahead_setup(h);
    if (rc) return rc;
    rc = ensure_fast_model(h);
    if (rc) return rc;
    const uint64_t call_base = (h->episode << 32) + (uint64_t)mpc_step * (uint64_t)(iters + 1);
    auto pool_of = [&](int it) -> float* { return it == iters - 1 ? (float*)b->actions : (float*)A.pool[(A.ctr + (unsigned)it) % 3]; };
    auto noise_args = [&](int n, uint64_t off, void* out) {
        return fast_sample_args(h, n, 0, nullptr, nullptr, nullptr, nullptr, off, 0, out, 0, nullptr, 0);
    };
    float* cur_mean = (float*)b->mean;
    float* cur_std = (float*)b->std;
    const int n_extra = (c.shift_elites && mpc_step > 0 && h->n_reuse > 0) ? h->n_reuse : 0;
    int n1_done = 0;   // rows of iteration 1's noise that are there already
    for (int it = 0; it < iters; ++it) {
        const bool last = it == iters - 1;
        const int n = h->pop[it];
        float* pool = pool_of(it);
        icem_plan_buffers bb = *b;
        bb.actions = pool;
        if (it & 1) bb.workspace = h->ws_alt;
        bb.mean = cur_mean;
        bb.std = cur_std;
        if (it == 0) {
            // this step's first noise: drawn by the previous step's last launch -- or, for a step nobody predicted, here
            const bool hit = A.next_valid && A.next_episode == h->episode && A.next_step == mpc_step && A.next_pool == pool &&
                             A.next_stream == st;   // (another stream is not ordered behind the launch that drew it: a miss)
            A.next_valid = false;
            if (!hit) {
                const FastSampleArgs za = noise_args(n, call_base, pool);
                ProfScope prof(h, ICEM_K_SAMPLE, (long long)n * c.horizon, st);
                launch_noise_rows(za, c.rng_rounds, st);
            }
            ICEM_HIP_TRY(hipGetLastError());
            // the head of iteration 1's noise may have been drawn beside the previous step's last merge as well
            n1_done = (hit && iters > 1 && A.next1_pool == pool_of(1)) ? std::min(A.next1_rows, h->pop[1]) : 0;
            A.next1_rows = 0;
        }
        IterAheadArgs ia = zeroed_args<IterAheadArgs>();
        ia.m.keep_base = -1;
        ia.r = fast_rollout_args(h, n, n, K, b->obs0, pool, b->costs, nullptr, nullptr);
        ia.r.part_k = (unsigned long long*)bb.workspace;
        ia.dbg_slot = it;
        ia.has_merge = h->pm_pending ? 1 : 0;
        if (h->pm_pending) ia.m = h->pm_args;
        h->pm_pending = false;
        ia.n_xf = n;
        ia.row0_mean = (last && c.use_mean_actions) ? 1 : 0;
        ia.store_back = last ? 1 : 0;
        ia.pool = pool;
        ia.mean = cur_mean;
        ia.std = cur_std;
        ia.lo = A.lo;
        ia.hi = A.hi;
        // the noise role: the next sampling call
        if (!last) {
            const int skip = it == 0 ? n1_done : 0;
            ia.z = noise_args(h->pop[it + 1] - skip, call_base + (uint64_t)(it + 1), pool_of(it + 1) + (size_t)skip * hd);
            ia.z.first_index = skip;
        } else {
            // iteration 0 of the NEXT MPC step (same episode, step + 1 -- checked when it comes)
            // -- split: what fits beside this launch's rollout here, the rest beside the step's last merge, a launch that
            // leaves 255 of the 256 CUs idle (ICEM_AHEAD_TAIL_FRAC: share of the rows that goes there)
            void* np = A.pool[(A.ctr + (unsigned)(iters - 1)) % 3];
            const uint64_t off0 = (h->episode << 32) + (uint64_t)(mpc_step + 1) * (uint64_t)(iters + 1);
            const double tail_frac = opt(OPT_AHEAD_TAIL_FRAC);
            int n_tail = (int)(tail_frac * h->pop[0]);
            n_tail = std::max(0, std::min(h->pop[0], n_tail));
            const int n_here = h->pop[0] - n_tail;
            ia.z = noise_args(n_here, off0, np);
            A.tail_pending = n_tail > 0;
            A.tail2_args = zeroed_args<FastSampleArgs>();
            A.tail2_args.n = 0;
            if (n_tail > 0) {
                A.tail_args = noise_args(n_tail, off0, (float*)np + (size_t)n_here * hd);
                A.tail_args.first_index = n_here;
                // ... and the head of the next step's iteration-1 noise, into the pool that step will find at pool_of(1)
                // (this step's iteration-2 pool: its last reader was iteration 3's prologue) -- ICEM_AHEAD_NEXT1_FRAC of it
                const double frac1 = opt(OPT_AHEAD_NEXT1_FRAC);
                const int n1 = iters > 2 ? std::max(0, std::min(h->pop[1], (int)(frac1 * h->pop[1]))) : 0;
                if (n1 > 0) {
                    void* np1 = A.pool[(A.ctr + (unsigned)(iters - 1) + 1u) % 3];
                    A.tail2_args = noise_args(n1, off0 + 1u, np1);
                    A.next1_rows = n1;
                    A.next1_pool = np1;
                }
            }
            A.next_valid = true;
            A.next_episode = h->episode;
            A.next_step = mpc_step + 1;
            A.next_pool = np;
            A.next_stream = st;
        }
        // the shift role (icem.py:91-104, 131-137): rows [n, n + n_extra) of this pool, costs behind costs[n]
        if (it == 0 && n_extra > 0) {
            const int g = (int)(((long long)mpc_step * iters) & 1);  // elite buffer holding the previous step's set
            ia.s = fast_sample_args(h, n, 0, cur_mean, cur_std, b->low, b->high, call_base, 0, pool, n_extra,
                                    (const float*)b->elites + (size_t)g * K * hd, call_base + (uint64_t)iters);
        }
        {
            ProfScope prof(h, ICEM_K_SAMPLE_ROLLOUT, (long long)n * c.horizon, st);
            launch_iter_ahead(ia, c.horizon, c.act_dim, h->Of, h->model_kind, st);
        }
        ICEM_HIP_TRY(hipGetLastError());
        h->fast_lists = ahead_roll_workgroups(n);
        h->fast_tail_rows = it == 0 ? n_extra : 0;
        // ---- its merge: stashed for the next launch's prologue, or (last) a launch of its own ----
        float* pp = h->pp_stats + (size_t)(it & 1) * 2 * hd;
        h->defer_merge = !last;
        h->merge_mean_out = last ? (float*)b->mean : pp;
        h->merge_std_out = last ? (float*)b->std : pp + hd;
        rc = plan_iter_merge_t<float>(h, &bb, mpc_step, it, st);
        h->defer_merge = false;
        h->merge_mean_out = h->merge_std_out = nullptr;
        if (rc) return rc;
        if (last && A.tail_pending) {  // (the merge could not take it along: launches of their own)
            if (g_batch.rec) g_batch.unsupported = true;   // (icem_plan_step_batch: these would run AHEAD of the recorded launches)
            launch_noise_rows(A.tail_args, c.rng_rounds, st);
            if (A.tail2_args.n > 0) launch_noise_rows(A.tail2_args, c.rng_rounds, st);
            ICEM_HIP_TRY(hipGetLastError());
            A.tail_pending = false;
        }
        if (!last) {
            if (!h->pm_pending) return fail(ICEM_E_STATE, "noise-ahead: the merge did not defer");
            h->pm_args.n_raw = n;  // this pool keeps its noise: the next prologue maps the elite rows among rows [0, n)
            h->pm_args.xf_lo = A.lo;
            h->pm_args.xf_hi = A.hi;
            cur_mean = pp;
            cur_std = pp + hd;
        }
    }
    A.ctr += (unsigned long long)(iters - 1);
    return ICEM_OK;
}

// The same pipeline for one rank of a sharded run (world > 1, in-library exchange connected, peers in processes of their
// own).  Per iteration ONE local launch: workgroup 0 packs + pushes the PREVIOUS iteration's K records and runs the
// launch's one records merge for everybody (published mean | std), the rollout workgroups wait for that flag, map and
// roll out this rank's shard, the noise workgroups draw the shard's next noise; rank 0 alone builds and rolls out the
// (replicated) shifted elites.  Every pool is written back (the record pack gathers actions).  The last iteration's
// pack and merge share one launch of their own, as on the sampler + rollout path.
static int plan_step_sharded_ahead(icem_handle* h, const icem_plan_buffers* b, int mpc_step, hipStream_t st) {
    icem_handle::Ahead& A = h->ahead;
    const icem_config& c = h->cfg;
    const int iters = c.opt_iters, K = c.num_elites, hd = h->hd;
    int rc = ahead_setup(h);
    if (rc) return rc;
    rc = ensure_fast_model(h);
    if (rc) return rc;
    if (!h->pub_dev) {
        ICEM_HIP_TRY(hipMalloc((void**)&h->pub_dev, ((size_t)2 * hd + 16) * sizeof(float)));
        ICEM_HIP_TRY(hipMemsetAsync(h->pub_dev, 0, ((size_t)2 * hd + 16) * sizeof(float), st));
    }
    const uint64_t call_base = (h->episode << 32) + (uint64_t)mpc_step * (uint64_t)(iters + 1);
    auto pool_of = [&](int it) -> float* { return it == iters - 1 ? (float*)b->actions : (float*)A.pool[(A.ctr + (unsigned)it) % 3]; };
    auto shard = [&](int it, int* lo_out) {
        const int n_global = h->pop[it], chunk = shard_chunk(n_global, c.world);
        const int lo = std::min(n_global, c.rank * chunk);
        *lo_out = lo;
        return std::max(0, std::min(n_global - lo, chunk));
    };
    auto noise_args = [&](int n, int lo, uint64_t off, void* out) {
        return fast_sample_args(h, n, lo, nullptr, nullptr, nullptr, nullptr, off, 0, out, 0, nullptr, 0);
    };
    float* cur_mean = (float*)b->mean;
    float* cur_std = (float*)b->std;
    const int n_extra = (c.shift_elites && mpc_step > 0 && h->n_reuse > 0) ? h->n_reuse : 0;
    float* rec = (float*)b->records + (size_t)c.rank * K * (hd + 2);
    PackPrev pack{};        // the previous iteration's pack, riding in this iteration's launch
    bool pack_pending = false;
    for (int it = 0; it < iters; ++it) {
        const bool last = it == iters - 1;
        int lo = 0;
        const int n_loc = shard(it, &lo);
        const int n_global = h->pop[it];
        float* pool = pool_of(it);
        icem_plan_buffers bb = *b;
        bb.actions = pool;
        bb.mean = cur_mean;
        bb.std = cur_std;
        if (it == 0) {
            const bool hit = A.next_valid && A.next_episode == h->episode && A.next_step == mpc_step && A.next_pool == pool &&
                             A.next_stream == st;   // (another stream is not ordered behind the launch that drew it: a miss)
            A.next_valid = false;
            if (!hit) {
                const FastSampleArgs za = noise_args(n_loc, lo, call_base, pool);
                ProfScope prof(h, ICEM_K_SAMPLE, (long long)n_loc * c.horizon, st);
                launch_noise_rows(za, c.rng_rounds, st);
            }
            ICEM_HIP_TRY(hipGetLastError());
        }
        const int tail = (it == 0 && c.rank == 0) ? n_extra : 0;  // shifted elites: rank 0 submits them (they are replicated)
        IterAheadArgs ia{};
        ia.r = fast_rollout_args(h, n_loc, n_loc, K, b->obs0, pool, b->costs, nullptr, nullptr);
        ia.r.part_k = (unsigned long long*)b->workspace;
        ia.n_xf = n_loc;
        ia.row0_mean = (last && c.use_mean_actions && lo == 0) ? 1 : 0;
        ia.store_back = 1;
        ia.pool = pool;
        ia.mean = cur_mean;
        ia.std = cur_std;
        ia.lo = A.lo;
        ia.hi = A.hi;
        if (it > 0) {
            if (!h->pm_pending || !pack_pending) return fail(ICEM_E_STATE, "noise-ahead (sharded): merge / pack not stashed");
            ia.has_merge = 2;
            ia.m = h->pm_args;
            ia.p = pack;
            ia.p.pub = h->pub_dev;
            ia.p.pub_flag = reinterpret_cast<unsigned*>(h->pub_dev + 2 * hd);
            ia.p.pub_seq = ++h->pub_seq;
            h->pm_pending = false;
            pack_pending = false;
        }
        if (!last) {
            int lo1 = 0;
            const int n1 = shard(it + 1, &lo1);
            ia.z = noise_args(n1, lo1, call_base + (uint64_t)(it + 1), pool_of(it + 1));
        } else {
            int lo0 = 0;
            const int n0 = shard(0, &lo0);
            void* np = A.pool[(A.ctr + (unsigned)(iters - 1)) % 3];
            ia.z = noise_args(n0, lo0, (h->episode << 32) + (uint64_t)(mpc_step + 1) * (uint64_t)(iters + 1), np);
            A.next_valid = true;
            A.next_episode = h->episode;
            A.next_step = mpc_step + 1;
            A.next_pool = np;
            A.next_stream = st;
        }
        if (tail > 0) {
            const int g = (int)(((long long)mpc_step * iters) & 1);
            ia.s = fast_sample_args(h, n_loc, 0, cur_mean, cur_std, b->low, b->high, call_base, 0, pool, tail,
                                    (const float*)b->elites + (size_t)g * K * hd, call_base + (uint64_t)iters);
        }
        {
            ProfScope prof(h, ICEM_K_SAMPLE_ROLLOUT, (long long)n_loc * c.horizon, st);
            launch_iter_ahead(ia, c.horizon, c.act_dim, h->Of, h->model_kind, st);
        }
        ICEM_HIP_TRY(hipGetLastError());
        const int lists = ahead_roll_workgroups(n_loc);
        h->fast_lists = lists;
        h->fast_tail_rows = 0;
        // ---- this iteration's pack: stashed for the next launch's workgroup 0, or (last) a launch of its own ----
        MergeSingleArgs pk{};
        pk.n_lists = lists;
        pk.n_pool = n_loc + tail;
        pk.n_global = n_global;
        pk.K = K;
        pk.h = c.horizon;
        pk.d = c.act_dim;
        pk.part_k = (const unsigned long long*)b->workspace;
        pk.actions = pool;
        if (tail > 0) {  // rank 0's shifted elites: extra candidates of the pack, costs from the cost array
            pk.n_keep = tail;
            pk.elites_cost_cur = (const float*)b->costs + n_loc;
            pk.keep_base = n_loc;
        }
        XchgPush px;
        rc = xchg_begin(h, &px, &h->xw_last);
        if (rc) return rc;
        if (!last) {
            pack.part_k = pk.part_k;
            pack.actions = pk.actions;
            pack.n_lists = pk.n_lists;
            pack.n_pool = pk.n_pool;
            pack.n_global = pk.n_global;
            pack.K = K;
            pack.n_loc = n_loc;
            pack.shard_lo = lo;
            pack.n_keep = pk.n_keep;
            pack.keep_costs = pk.elites_cost_cur;
            pack.records = rec;
            pack.px = px;
            pack_pending = true;
        } else {
            // the step's last pack: stashed for the merge below, which takes it along in its own launch (pack_merge_kernel)
            PackPrev& pp = h->pk_args;
            pp.part_k = pk.part_k;
            pp.actions = pk.actions;
            pp.n_lists = pk.n_lists;
            pp.n_pool = pk.n_pool;
            pp.n_global = pk.n_global;
            pp.K = K;
            pp.n_loc = n_loc;
            pp.shard_lo = lo;
            pp.n_keep = pk.n_keep;
            pp.keep_costs = pk.elites_cost_cur;
            pp.records = rec;
            pp.px = px;
            h->pk_pending = true;
        }
        // ---- its merge (records form): stashed for the next launch's pack role, or (last) a launch with the pack ----
        float* pp = h->pp_stats + (size_t)(it & 1) * 2 * hd;
        h->defer_merge = !last;
        h->merge_mean_out = last ? (float*)b->mean : pp;
        h->merge_std_out = last ? (float*)b->std : pp + hd;
        rc = plan_iter_merge_t<float>(h, &bb, mpc_step, it, st);
        h->defer_merge = false;
        h->merge_mean_out = h->merge_std_out = nullptr;
        if (rc) return rc;
        if (!last) {
            if (!h->pm_pending || h->pm_args.records == nullptr) return fail(ICEM_E_STATE, "noise-ahead (sharded): the merge did not defer");
            cur_mean = pp;
            cur_std = pp + hd;
        }
    }
    A.ctr += (unsigned long long)(iters - 1);
    return ICEM_OK;
}

// Small populations (the single-launch kernel serves iteration 0): the raw colored noise of the NEXT MPC step's first
// sampling call -- and of its shifted elites' call -- needs neither the observation nor the distribution (icem.py:73-79),
// so it is drawn beside THIS step's last merge, the one launch that leaves 255 CUs idle (merge_noise_kernel), into a
// buffer of the handle; iteration 0 of the next step then only maps it (FastSampleArgs::raw_src).  Same sample_row, same
// map: same bits as sampling in place.  Arms h->ahead.tail_args for plan_iter_merge_t's last launch.
void predraw_next_step(icem_handle* h, const icem_plan_buffers* b, int mpc_step, hipStream_t st) {
    icem_handle::Ahead& A = h->ahead;
    const icem_config& c = h->cfg;
    const int on = opt_i(OPT_PREDRAW);
    A.pre_valid = false;
    if (!on || c.world != 1 || c.dtype != ICEM_F32 || b->z_r != nullptr || !h->use_fast || gemm_rollout(h) || c.rng_rounds != 10 ||
        h->dbg != nullptr || h->fast_lists <= 0 || c.num_elites + 1 > 12 || !fast_rollout_ok(h, c.num_elites) || !fast_sample_ok(h))
        return;
    const int n0 = h->pop[0];
    // (measured: N = 1000 66.7 -> 64.9, N = 4096 67.9 -> 65.9 us per MPC step; 8192 unchanged; at 16 384 the noise outlasts
    //  the merge it rides with, 89.4 -> 91.1: up to 8192 rows)
    if (n0 > 8192) return;
    const int n_shift = (c.shift_elites && h->n_reuse > 0) ? h->n_reuse : 0;   // (mpc_step + 1 > 0: the next step shifts)
    if (n_shift * c.act_dim > 256) return;
    int tail_rows = 0;
    if (one_launch_lists(h, n0 + n_shift, n_shift, &tail_rows) <= 0) return;
    if (!A.pre_raw && hipMalloc(&A.pre_raw, (size_t)(n0 + h->n_reuse + 16) * h->hd * sizeof(float)) != hipSuccess) {
        (void)hipGetLastError();
        A.pre_raw = nullptr;
        return;
    }
    const uint64_t base_next = (h->episode << 32) + (uint64_t)(mpc_step + 1) * (uint64_t)(c.opt_iters + 1);
    A.tail_args = fast_sample_args(h, n0, 0, nullptr, nullptr, nullptr, nullptr, base_next, 0, A.pre_raw, 0, nullptr, 0);
    A.tail2_args = fast_sample_args(h, n_shift, 0, nullptr, nullptr, nullptr, nullptr, base_next + (uint64_t)c.opt_iters, 0,
                                    (float*)A.pre_raw + (size_t)n0 * h->hd, 0, nullptr, 0);
    A.tail_pending = true;
    A.pre_valid = true;
    A.pre_episode = h->episode;
    A.pre_step = mpc_step + 1;
    A.pre_stream = st;
}

}  // namespace icem

extern "C" {

size_t icem_plan_buffer_bytes(const icem_handle* h, int32_t which) {
    if (!h) return 0;
    const size_t ts = h->tsize, hd = (size_t)h->hd, K = (size_t)h->cfg.num_elites;
    const size_t rows = (size_t)h->n_local_max + (size_t)h->n_reuse;
    switch (which) {
        case ICEM_BUF_MEAN:
        case ICEM_BUF_STD:
            return hd * ts;
        case ICEM_BUF_LOW:
        case ICEM_BUF_HIGH:
        case ICEM_BUF_EXECUTED:
            return (size_t)h->cfg.act_dim * ts;
        case ICEM_BUF_OBS0:
            return (size_t)std::max(1, h->obs_dim) * ts;
        case ICEM_BUF_ACTIONS:
            return rows * hd * ts;
        case ICEM_BUF_COSTS:
            return rows * ts;
        case ICEM_BUF_ELITES:
            return 2 * K * hd * ts + 2 * K * ts;
        case ICEM_BUF_RECORDS:
            return (size_t)h->cfg.world * K * (hd + 2) * ts;
        case ICEM_BUF_WORKSPACE:
            return (size_t)std::max(topk_blocks((int)rows), 1024) * K * (ts + sizeof(int));
        case ICEM_BUF_BEST_COST:
            return ts;
        default:
            return 0;
    }
}

static int check_plan(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it, hipStream_t st = nullptr) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!b) return fail(ICEM_E_INVALID, "null buffers");
    if (!h->has_model || !h->has_cost) return fail(ICEM_E_STATE, "icem_set_model / icem_set_cost must be called first");
    if (mpc_step < 0 || it < 0 || it >= h->cfg.opt_iters) return fail(ICEM_E_INVALID, "mpc_step / iteration out of range");
    if (const char* e = cost_indices_error(h, h->obs_dim)) return fail(ICEM_E_INVALID, e);
    if (const char* e = wide_unsupported(h, h->cfg.num_elites, b->z_r != nullptr, false)) return fail(ICEM_E_UNSUPPORTED, e);
    if (!b->mean || !b->std || !b->low || !b->high || !b->obs0 || !b->actions || !b->costs || !b->elites || !b->records ||
        !b->workspace || !b->executed || !b->best_cost)
        return fail(ICEM_E_INVALID, "null plan buffer");
    if (h->cfg.noise_beta > 0 &&
        ((b->z_r == nullptr) != (b->z_i == nullptr) || (b->z_r_shift == nullptr) != (b->z_i_shift == nullptr)))
        return fail(ICEM_E_INVALID, "z_r/z_i must be given in pairs");
    // fp16-plane tile arithmetic: the launch's scale needs the action bounds' magnitude -- fetched once per (low, high) pair
    // (and at every icem_reset_distribution: abi.hip::refresh_act_mag)
    if (h->tile_arith || h->hn_tile) return refresh_act_mag(h, b->low, b->high, st, false);
    return ICEM_OK;
}

static int ensure_pp_stats(icem_handle* h) {
    if (!h->pp_stats) ICEM_HIP_TRY(hipMalloc((void**)&h->pp_stats, (size_t)4 * h->hd * sizeof(float)));
    return ICEM_OK;
}

// world > 1 with merge deferral: the running step's distribution is at cur_mean / cur_std
static bool deferral_active(const icem_handle* h, const icem_plan_buffers* b) {
    return h->deferral && h->cfg.world > 1 && h->cfg.dtype == ICEM_F32 && b->z_r == nullptr && h->use_fast;
}

int icem_set_merge_deferral(icem_handle* h, int32_t on) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (h->pm_pending) return fail(ICEM_E_STATE, "a deferred merge is pending: finish the MPC step first");
    h->deferral = on != 0;
    return ICEM_OK;
}

int icem_plan_iter_local(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it, void* stream) {
    int rc = check_plan(h, b, mpc_step, it, (hipStream_t)stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    icem_plan_buffers bb = *b;
    if (deferral_active(h, b)) {
        if (it == 0 || !h->cur_mean) {
            h->cur_mean = (float*)b->mean;
            h->cur_std = (float*)b->std;
        }
        bb.mean = h->cur_mean;
        bb.std = h->cur_std;
    }
    return ICEM_DISPATCH(h, plan_iter_local_t<float>(h, &bb, mpc_step, it, st), plan_iter_local_t<double>(h, &bb, mpc_step, it, st));
}

int icem_plan_iter_merge(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, int32_t it, void* stream) {
    int rc = check_plan(h, b, mpc_step, it, (hipStream_t)stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (!deferral_active(h, b) || !h->cur_mean)
        return ICEM_DISPATCH(h, plan_iter_merge_t<float>(h, b, mpc_step, it, st), plan_iter_merge_t<double>(h, b, mpc_step, it, st));
    // sharded, deferral on: a non-last merge may ride in the next icem_plan_iter_local launch (mean / std / elites in
    // the caller's buffers are then current again only after that launch; the last merge always runs here)
    rc = ensure_pp_stats(h);
    if (rc) return rc;
    const bool last = it == h->cfg.opt_iters - 1;
    // (the prologue's record selection holds two records per lane: world * K <= 128)
    const bool fold = !last && h->fast_lists > 0 && h->cfg.world * h->cfg.num_elites <= 128 && h->cfg.num_elites <= 32 &&
                      prologue_possible(h, local_rows(h, it + 1));
    icem_plan_buffers bb = *b;
    bb.mean = h->cur_mean;
    bb.std = h->cur_std;
    float* pp = h->pp_stats + (size_t)(it & 1) * 2 * h->hd;
    h->defer_merge = fold;
    if (last) {
        h->merge_mean_out = (float*)b->mean;
        h->merge_std_out = (float*)b->std;
    } else if (fold) {
        h->merge_mean_out = pp;
        h->merge_std_out = pp + h->hd;
    } else {
        h->merge_mean_out = h->merge_std_out = nullptr;
    }
    rc = plan_iter_merge_t<float>(h, &bb, mpc_step, it, st);
    h->defer_merge = false;
    h->merge_mean_out = h->merge_std_out = nullptr;
    if (rc) return rc;
    if (fold && h->pm_pending) {
        h->cur_mean = pp;
        h->cur_std = pp + h->hd;
    }
    if (last) h->cur_mean = h->cur_std = nullptr;
    return ICEM_OK;
}

// A step that failed half way must not leave arguments armed for a later step's launches (stale merge / pack / noise
// arguments would reach kernels with buffers of a step that never completed).
static int disarm_on_error(icem_handle* h, int rc) {
    if (rc != ICEM_OK) {
        h->ahead.tail_pending = h->ahead.next_valid = h->ahead.pre_valid = false;
        h->ahead.next1_rows = 0;
        h->pm_pending = h->pk_pending = false;
        h->defer_merge = false;
        h->merge_mean_out = h->merge_std_out = nullptr;
    }
    return rc;
}

static int plan_step_sharded_body(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, void* stream) {
    if (h->cfg.world < 2) return icem_plan_step(h, b, mpc_step, stream);
    if (!xchg_connected(h) && !rccl_connected(h))
        return fail(ICEM_E_STATE, "neither the in-library exchange (icem_exchange_create / _connect) nor an RCCL communicator "
                                  "(icem_rccl_connect / _adopt) is connected");
    if (b && b->z_r != nullptr) return fail(ICEM_E_INVALID, "external noise goes through icem_plan_iter_local / _merge");
    // A merge of an earlier step gave up waiting for a peer's records (bounded waits): that step's elites, mean and
    // action are garbage and differ between ranks.  The status word is host memory -- this check costs one load.
    if (xchg_status_peek(h) & 1u)
        return fail(ICEM_E_STATE, "in-library exchange: a wait for a peer's elite records timed out in an earlier MPC step "
                                  "(icem_exchange_status reads and clears the word); the plans since then are not valid");
    if (b && check_plan(h, b, mpc_step, 0, (hipStream_t)stream) == ICEM_OK && !h->pm_pending && !h->pk_pending && ahead_eligible(h, b, true))
        return plan_step_sharded_ahead(h, b, mpc_step, (hipStream_t)stream);
    const bool was = h->deferral;
    h->deferral = true;  // non-last merges ride in the next local launch
    int rc = ICEM_OK;
    const bool by_rccl = !xchg_connected(h);  // the pack left this rank's K records in b->records: gather them in place
    for (int it = 0; it < h->cfg.opt_iters && rc == ICEM_OK; ++it) {
        rc = icem_plan_iter_local(h, b, mpc_step, it, stream);
        if (rc == ICEM_OK && by_rccl) rc = rccl_allgather_records(h, b->records, (hipStream_t)stream);
        if (rc == ICEM_OK) rc = icem_plan_iter_merge(h, b, mpc_step, it, stream);
    }
    h->deferral = was;
    return rc;
}

int icem_plan_step_sharded(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    return disarm_on_error(h, plan_step_sharded_body(h, b, mpc_step, stream));
}


// ---------------------------------------------------------------------------------------------------------------------
// The step of a small population as ONE launch inside one XCD (k_step_xcd.hip)
// ---------------------------------------------------------------------------------------------------------------------
static bool step_xcd_eligible(icem_handle* h, const icem_plan_buffers* b) {
    const icem_config& c = h->cfg;
    if (!opt_i(OPT_STEP_XCD) || h->sx.disabled || g_batch.rec) return false;
    if (c.world != 1 || c.dtype != ICEM_F32 || !h->use_fast || b->z_r != nullptr || c.rng_rounds != 10) return false;
    if (h->dbg != nullptr && !opt_i(OPT_AHEAD_STAMPS)) return false;   // (another kernel is under study; option ahead_stamps = 1: this one's stamps)
    if (gemm_rollout(h) || h->hn_tile || h->Of == 0 || !fast_rollout_ok(h, c.num_elites) || !fast_sample_ok(h)) return false;
    if (!step_xcd_supported(c.horizon, c.act_dim, h->Of, c.num_elites) || c.opt_iters > STEP_XCD_MAX_ITERS) return false;
    for (int n : h->pop)
        if (n > step_xcd_max_rows()) return false;
    if (h->n_reuse > 16 || h->n_reuse > c.num_elites) return false;
    if (h->sx.cus == 0) {
        hipDeviceProp_t p;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
            (void)hipGetLastError();
            h->sx.cus = -1;
        } else {
            h->sx.cus = p.multiProcessorCount;
        }
    }
    return h->sx.cus == 256;   // 8 XCDs x 32 CUs, one workgroup each (a partitioned or masked device keeps the launches per iteration)
}

static int plan_step_xcd(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, hipStream_t st) {
    const icem_config& c = h->cfg;
    icem_handle::StepXcd& X = h->sx;
    const int iters = c.opt_iters, K = c.num_elites, hd = h->hd;
    int rc = ensure_fast_model(h);
    if (rc) return rc;
    // buffers of the handle: the partner pool and list array of the launches-per-iteration path, raw noise, shift rows, state
    if (iters > 1 && !h->actions_alt) ICEM_HIP_TRY(hipMalloc(&h->actions_alt, icem_plan_buffer_bytes(h, ICEM_BUF_ACTIONS)));
    if (iters > 1 && !h->ws_alt) ICEM_HIP_TRY(hipMalloc(&h->ws_alt, icem_plan_buffer_bytes(h, ICEM_BUF_WORKSPACE)));
    size_t rows_all = 0;
    for (int n : h->pop) rows_all += (size_t)n;
    if (!X.raw) ICEM_HIP_TRY(hipMalloc(&X.raw, (rows_all + 16) * hd * sizeof(float)));
    for (void*& p : X.pre)
        if (!p) ICEM_HIP_TRY(hipMalloc(&p, ((size_t)h->pop[0] + 16) * hd * sizeof(float)));
    if (!X.shift) ICEM_HIP_TRY(hipMalloc(&X.shift, ((size_t)16 * hd + 64) * sizeof(float)));
    if (!X.state) {
        ICEM_HIP_TRY(hipMalloc(&X.state, step_xcd_state_bytes()));
        ICEM_HIP_TRY(hipMemset(X.state, 0, step_xcd_state_bytes()));
    }
    StepXcdArgs a;
    std::memset((void*)&a, 0, sizeof(a));
    a.r = fast_rollout_args(h, 0, 0, K, b->obs0, nullptr, b->costs, nullptr, nullptr);
    a.iters = iters;
    a.K = K;
    a.n_reuse = h->n_reuse;
    a.n_shift = (c.shift_elites && mpc_step > 0 && h->n_reuse > 0) ? h->n_reuse : 0;
    a.keep = c.keep_previous_elites ? 1 : 0;
    a.use_mean = c.use_mean_actions ? 1 : 0;
    a.white = c.noise_beta <= 0 ? 1 : 0;
    a.g0 = (int)(((long long)mpc_step * iters) & 1);
    for (int it = 0; it < iters; ++it) a.pop[it] = h->pop[it];
    a.alpha = (float)c.alpha;
    a.init_std = (float)c.init_std;
    a.pool[0] = (float*)b->actions;
    a.pool[1] = (float*)(iters > 1 ? h->actions_alt : b->actions);
    a.lists[0] = (unsigned long long*)b->workspace;
    a.lists[1] = (unsigned long long*)(iters > 1 ? h->ws_alt : b->workspace);
    a.elites = (float*)b->elites;
    a.elites_cost = a.elites + (size_t)2 * K * hd;
    a.mean = (const float*)b->mean;
    a.std = (const float*)b->std;
    a.mean_out = (float*)b->mean;
    a.std_out = (float*)b->std;
    a.low = (const float*)b->low;
    a.high = (const float*)b->high;
    a.executed = (float*)b->executed;
    a.best_cost = (float*)b->best_cost;
    a.W = (const float*)h->W_dev;
    a.seed_lo = (uint32_t)c.seed;
    a.seed_hi = (uint32_t)(c.seed >> 32);
    const uint64_t call_base = (h->episode << 32) + (uint64_t)mpc_step * (uint64_t)(iters + 1);
    // iteration 0's noise: drawn by the previous step's launch on this stream, or here
    const bool pre0 = X.pre_valid && X.pre_episode == h->episode && X.pre_step == mpc_step && X.pre_stream == st;
    const int rpm = step_xcd_rows_per_member();
    a.n_seg = iters + 1;
    size_t at = 0;
    int chunks = 0;
    for (int s = 0; s <= iters; ++s) {
        const bool next0 = s == iters;
        const int n = next0 ? h->pop[0] : ((s == 0 && pre0) ? 0 : h->pop[s]);
        const uint64_t off = next0 ? call_base + (uint64_t)(iters + 1) : call_base + (uint64_t)s;
        a.seg_n[s] = n;
        a.seg_off_lo[s] = (uint32_t)off;
        a.seg_off_hi[s] = (uint32_t)(off >> 32);
        a.seg_chunk0[s] = chunks;
        chunks += (n + rpm - 1) / rpm;
        if (next0) {
            a.seg_out[s] = (float*)X.pre[(mpc_step + 1) & 1];
        } else {
            a.seg_out[s] = (float*)X.raw + at * hd;
            a.raw[s] = (s == 0 && pre0) ? (const float*)X.pre[mpc_step & 1] : a.seg_out[s];
            at += (size_t)h->pop[s];
        }
    }
    a.seg_chunk0[iters + 1] = chunks;
    a.n_jobs = chunks + (a.n_shift > 0 ? 1 : 0);
    a.shift_src = a.elites + (size_t)a.g0 * K * hd;   // the previous step's elite set
    const uint64_t soff = call_base + (uint64_t)iters;
    a.shift_off_lo = (uint32_t)soff;
    a.shift_off_hi = (uint32_t)(soff >> 32);
    a.shift_rows = (float*)X.shift;
    a.shift_costs = (float*)X.shift + (size_t)16 * hd + 32;   // (a cache line of its own behind the rows)
    a.state = (unsigned*)X.state;
    a.bar_base = (unsigned)(X.launches * (unsigned long long)iters * 32ull);
    a.max_polls = 1u << 22;
    long long units = 0;
    for (int n : h->pop) units += (long long)n * c.horizon;
    {
        ProfScope prof(h, ICEM_K_SAMPLE_ROLLOUT, units, st);
        launch_step_xcd(a, c.horizon, c.act_dim, h->Of, h->model_kind, st);
    }
    ICEM_HIP_TRY(hipGetLastError());
    ++X.launches;
    X.pre_valid = true;
    X.pre_episode = h->episode;
    X.pre_step = mpc_step + 1;
    X.pre_stream = st;
    // what the launches-per-iteration path keeps across steps does not describe this step
    h->ahead.pre_valid = h->ahead.tail_pending = false;
    h->fast_lists = 0;
    return ICEM_OK;
}

extern "C" int icem_step_status(icem_handle* h, int64_t* xcd_launches_out, int32_t* timed_out_out, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (xcd_launches_out) *xcd_launches_out = (int64_t)h->sx.launches;
    unsigned v = 0;
    if (h->sx.state) {
        hipStream_t st = (hipStream_t)stream;
        ICEM_HIP_TRY(hipMemcpyAsync(&v, (const unsigned*)h->sx.state + 28, sizeof(v), hipMemcpyDeviceToHost, st));
        ICEM_HIP_TRY(hipStreamSynchronize(st));
    }
    if (v) h->sx.disabled = true;
    if (timed_out_out) *timed_out_out = v ? 1 : 0;
    return ICEM_OK;
}

static int plan_step_body(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, void* stream) {
    if (h->cfg.world != 1) return fail(ICEM_E_INVALID, "icem_plan_step is the world == 1 path; use iter_local/iter_merge");
    int rc = check_plan(h, b, mpc_step, 0, (hipStream_t)stream);
    if (rc) return rc;
    const icem_config& c = h->cfg;
    const int iters = c.opt_iters;
    if (step_xcd_eligible(h, b)) return plan_step_xcd(h, b, mpc_step, (hipStream_t)stream);
    h->sx.pre_valid = false;   // (a step on the launches-per-iteration path: the noise a one-launch step drew ahead is not its)
    if (ahead_eligible(h, b)) return plan_step_ahead(h, b, mpc_step, (hipStream_t)stream);
    // f32, device noise: iteration it's merge may ride in the prologue of iteration it+1's launch.  That launch
    // reads pool / lists / distribution of iteration it while writing its own, so consecutive iterations alternate
    // between the caller's buffers and the handle's partners (the last iteration always uses the caller's).
    const bool pingpong = c.dtype == ICEM_F32 && b->z_r == nullptr && h->use_fast && iters > 1;
    if (pingpong && !h->actions_alt) ICEM_HIP_TRY(hipMalloc(&h->actions_alt, icem_plan_buffer_bytes(h, ICEM_BUF_ACTIONS)));
    if (pingpong && !h->ws_alt) ICEM_HIP_TRY(hipMalloc(&h->ws_alt, icem_plan_buffer_bytes(h, ICEM_BUF_WORKSPACE)));   // (ahead_setup may own one already)
    if (pingpong) {
        rc = ensure_pp_stats(h);
        if (rc) return rc;
    }
    float* cur_mean = (float*)b->mean;  // where the current distribution lives
    float* cur_std = (float*)b->std;
    for (int it = 0; it < iters; ++it) {
        icem_plan_buffers bb = *b;
        if (pingpong) {
            if ((iters - 1 - it) & 1) bb.actions = h->actions_alt;
            if (it & 1) bb.workspace = h->ws_alt;
            bb.mean = cur_mean;
            bb.std = cur_std;
        }
        rc = icem_plan_iter_local(h, &bb, mpc_step, it, stream);
        if (rc) return rc;
        const bool last = it == iters - 1;
        if (last) predraw_next_step(h, b, mpc_step, (hipStream_t)stream);  // small populations: the next step's first noise rides with the last merge
        bool fold = false;
        if (pingpong && !last && h->fast_lists > 0) fold = prologue_possible(h, h->pop[it + 1]);
        h->defer_merge = fold;
        float* pp = pingpong ? h->pp_stats + (size_t)(it & 1) * 2 * h->hd : nullptr;
        if (last) {  // the final distribution goes to the caller's buffers
            h->merge_mean_out = (float*)b->mean;
            h->merge_std_out = (float*)b->std;
        } else if (fold) {
            h->merge_mean_out = pp;
            h->merge_std_out = pp + h->hd;
        } else {
            h->merge_mean_out = h->merge_std_out = nullptr;  // in place
        }
        rc = icem_plan_iter_merge(h, &bb, mpc_step, it, stream);
        h->defer_merge = false;
        h->merge_mean_out = h->merge_std_out = nullptr;
        if (rc) return rc;
        if (last && h->ahead.tail_pending) {  // the merge that ran was not one that takes noise along: no noise was drawn
            h->ahead.tail_pending = false;
            h->ahead.pre_valid = false;
        }
        if (fold && h->pm_pending) {
            cur_mean = pp;
            cur_std = pp + h->hd;
        }
    }
    return ICEM_OK;
}

int icem_plan_step(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    return disarm_on_error(h, plan_step_body(h, b, mpc_step, stream));
}


// ---------------------------------------------------------------------------------------------------------------------
// icem_plan_step_batch: B independent planners, one launch per stage for all of them
// ---------------------------------------------------------------------------------------------------------------------
// The reference runs several controllers side by side (parallel episodes: icem/misc/rollout_utils.py:46-58, 129-152), each
// its own get_action (icem.py:106-189).  At the metric's population (N = 4096) one problem leaves the chip mostly empty and a
// step is a chain of six launch latencies; B problems of one configuration share those six launches: the host side of
// every handle runs exactly as for icem_plan_step -- same buffers, same ping-pong, same noise offsets -- with the launchers
// recording their argument blocks (g_batch.rec), then every stage is ONE launch with blockIdx.y = the problem and the
// blocks in a device array (one upload per step, skipped when nothing but the step number changed: the offsets are stored
// relative to the step's base, which travels in the kernel arguments).  Slab sizes are chosen for all rows together.
struct BatchCtx {
    // six arrays, by the MPC step modulo 6: the elite buffers ping-pong per ITERATION (with an odd iteration count every second
    // step's blocks are the same again), the noise-ahead launches rotate three pools per step: 2 x 3 steps close every cycle
    static constexpr int SLOTS = 6;
    void* dev[SLOTS] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[SLOTS] = {0, 0, 0, 0, 0, 0};
    std::vector<unsigned char> shadow[SLOTS];   // what each device array holds
};
static void batch_ctx_free(void* p) {
    BatchCtx* c = (BatchCtx*)p;
    if (!c) return;
    for (void* d : c->dev)
        if (d) (void)hipFree(d);
    delete c;
}

static void sub_base(uint32_t& lo, uint32_t& hi, unsigned long long base) {
    const unsigned long long v = (((unsigned long long)hi << 32) | lo) - base;
    lo = (uint32_t)v;
    hi = (uint32_t)(v >> 32);
}

// does this handle's step consist of single-launch iterations with merge prologues and one last merge? (the path
// decisions of plan_step_body / plan_iter_local_t / plan_iter_merge_t, evaluated without launching anything)
static const char* batch_ineligible(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step) {
    const icem_config& c = h->cfg;
    const int K = c.num_elites;
    if (c.world != 1) return "world must be 1";
    if (c.dtype != ICEM_F32 || !h->use_fast) return "dtype f32 on the throughput kernels only";
    if (b->z_r || b->z_i || b->z_r_shift || b->z_i_shift) return "external noise is not batched";
    if (h->profiling || h->dbg) return "per-kernel profiling / debug stamps are per handle: switch them off";
    if (gemm_rollout(h) || h->hn_tile || h->Of == 0) return "only the 16-trajectory tile kernels (o <= 20 shapes) are batched";
    if (!fast_rollout_ok(h, K) || !fast_sample_ok(h) || K + 1 > 12 || c.rng_rounds != 10) return "shape outside the single-launch kernels";
    if (h->pm_pending || h->pk_pending) return "a deferred merge is pending: finish the MPC step first";
    if (c.opt_iters < 1) return "opt_iters";
    // (g_batch.mult is set: the shapes below are the batch's.)  Where every iteration's rows of ALL problems together fill the
    // noise-ahead launch (>= 4 waves per rollout workgroup), the batch takes that path -- rollout, next noise and shifted elites
    // as roles of one launch (k_rollout_ahead.hip), 107 against 130 us per step at eight problems of 4096 rows -- provided a
    // problem ALONE does not take it (a population above 8192 rows fills the chip by itself: not batched)
    {
        const bool ah = g_batch.ahead;
        g_batch.ahead = false;
        const bool alone = ahead_eligible(h, b);
        g_batch.ahead = ah;
        if (alone) return "populations that take the noise-ahead launches alone (> 8192 rows per iteration) are not batched: they fill the chip by themselves";
    }
    if (ahead_eligible(h, b)) return nullptr;   // (false without g_batch.ahead: option batch_ahead = 0)
    for (int it = 0; it < c.opt_iters; ++it) {
        const int n_extra = (it == 0 && c.shift_elites && mpc_step > 0) ? h->n_reuse : 0;
        if (n_extra * c.act_dim > 256) return "too many shifted elites for the sampling launch";
        if (one_launch_lists(h, h->pop[it] + n_extra, n_extra) <= 0) return "an iteration's population has no single-launch kernel";
        if (it > 0 && !prologue_possible(h, h->pop[it])) return "an iteration cannot carry the previous merge in its prologue";
    }
    return nullptr;
}

extern "C" int icem_plan_step_batch(icem_handle* const* handles, int32_t n, const icem_plan_buffers* buffers, int32_t mpc_step, void* stream) {
    if (!handles || !buffers || n < 1 || n > ICEM_MAX_BATCH) return fail(ICEM_E_INVALID, "null argument / n outside [1, 32]");
    for (int i = 0; i < n; ++i)
        if (!handles[i]) return fail(ICEM_E_INVALID, "null handle");
    if (n == 1) return icem_plan_step(handles[0], &buffers[0], mpc_step, stream);
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (handles[i] == handles[j]) return fail(ICEM_E_INVALID, "the same handle twice in one batch");
    // one configuration (every launch shape and template instantiation is shared); models, costs, seeds, bounds, observations differ
    const icem_handle* h0 = handles[0];
    for (int i = 0; i < n; ++i) {
        icem_handle* h = handles[i];
        const icem_config &a = h->cfg, &r = h0->cfg;
        if (a.horizon != r.horizon || a.act_dim != r.act_dim || a.num_traj != r.num_traj || a.num_elites != r.num_elites ||
            a.elites_size != r.elites_size || a.opt_iters != r.opt_iters || a.use_mean_actions != r.use_mean_actions ||
            a.keep_previous_elites != r.keep_previous_elites || a.shift_elites != r.shift_elites || a.factor_decrease != r.factor_decrease ||
            a.fraction_reused != r.fraction_reused || a.rng_rounds != r.rng_rounds || a.dtype != r.dtype || a.world != r.world ||
            (a.noise_beta > 0) != (r.noise_beta > 0))
            return fail(ICEM_E_INVALID, "icem_plan_step_batch: the handles must share one configuration (horizon, act_dim, populations, elites, flags)");
        int rc = check_plan(h, &buffers[i], mpc_step, 0, st);
        if (rc) return rc;
        if (h->Of != h0->Of || h->model_kind != h0->model_kind || h->tile_arith != h0->tile_arith)
            return fail(ICEM_E_INVALID, "icem_plan_step_batch: the handles must share the model's width and kind and the tile arithmetic");
    }
    struct MultGuard {
        MultGuard(int m, long long rows0) {
            g_batch.mult = m;
            g_batch.unsupported = false;
            // the noise-ahead launches from where they measure faster than the single-launch kernels: 12 problems of 4096 rows
            // (160 against 161 us per step; 16: 194 against 212; 8: 129 against 128; 6: 115 against 100 -- EXPERIMENTS R6.3)
            g_batch.ahead = opt_i(OPT_BATCH_AHEAD) != 0 && (double)m * (double)rows0 >= opt(OPT_BATCH_AHEAD_MIN_ROWS);
        }
        ~MultGuard() { g_batch.mult = 1; g_batch.rec = nullptr; g_batch.ahead = false; }
    } guard(n, h0->pop.empty() ? 0 : h0->pop[0]);
    for (int i = 0; i < n; ++i)
        if (const char* why = batch_ineligible(handles[i], &buffers[i], mpc_step))
            return fail(ICEM_E_UNSUPPORTED, std::string("icem_plan_step_batch: ") + why);
    // ---- the host side of every problem's step, recorded ----
    std::vector<std::vector<BatchRecord>> recs(n);
    int rc = ICEM_OK;
    for (int i = 0; i < n && rc == ICEM_OK; ++i) {
        g_batch.rec = &recs[i];
        rc = disarm_on_error(handles[i], plan_step_body(handles[i], &buffers[i], mpc_step, stream));
    }
    g_batch.rec = nullptr;
    if (rc == ICEM_OK && g_batch.unsupported) rc = fail(ICEM_E_UNSUPPORTED, "icem_plan_step_batch: a launch without a batched form was reached");
    const size_t L = recs[0].size();
    for (int i = 0; i < n && rc == ICEM_OK; ++i) {
        if (recs[i].size() != L || L == 0) rc = fail(ICEM_E_STATE, "icem_plan_step_batch: the problems' steps took different launches");
        for (size_t l = 0; l < L && rc == ICEM_OK; ++l) {
            const BatchRecord &x = recs[i][l], &y = recs[0][l];
            bool same = x.kind == y.kind;
            if (same && x.kind == 1)
                same = x.h == y.h && x.d == y.d && x.O == y.O && x.model_kind == y.model_kind && x.rw == y.rw && x.grid == y.grid &&
                       x.prologue == y.prologue && x.it.r.arith == y.it.r.arith;
            if (same && x.kind == 4)
                same = x.h == y.h && x.d == y.d && x.O == y.O && x.model_kind == y.model_kind && x.rw == y.rw && x.grid == y.grid &&
                       x.ia.has_merge == y.ia.has_merge && x.ia.r.arith == y.ia.r.arith && x.ia.z.n == y.ia.z.n &&
                       x.ia.s.n_shift == y.ia.s.n_shift && x.ia.n_noise == y.ia.n_noise;
            if (same && x.kind != 1 && x.kind != 4)
                same = x.m.h == y.m.h && x.m.d == y.m.d && (x.kind == 2 || (x.z1.n == y.z1.n && x.z2.n == y.z2.n && x.z1.d == y.z1.d && x.z1.h == y.z1.h));
            if (!same) rc = fail(ICEM_E_STATE, "icem_plan_step_batch: the problems' launches differ in shape");
        }
    }
    if (rc != ICEM_OK) {   // nothing was launched: the handles' half-armed state goes
        for (int i = 0; i < n; ++i) (void)disarm_on_error(handles[i], rc);
        return rc;
    }
    // ---- argument blocks: offsets relative to each problem's base of this step, one array per launch ----
    BatchBases bases{};
    for (int i = 0; i < n; ++i)
        bases.v[i] = (handles[i]->episode << 32) + (unsigned long long)mpc_step * (unsigned long long)(handles[i]->cfg.opt_iters + 1);
    std::vector<size_t> at(L);
    size_t bytes = 0;
    for (size_t l = 0; l < L; ++l) {
        at[l] = bytes;
        const size_t one = recs[0][l].kind == 1 ? sizeof(FastIterArgs) : recs[0][l].kind == 4 ? sizeof(IterAheadArgs) : sizeof(MergeNoiseBatchArgs);
        bytes += ((one * (size_t)n + 255) / 256) * 256;
    }
    std::vector<unsigned char> blob(bytes, 0);
    for (size_t l = 0; l < L; ++l)
        for (int i = 0; i < n; ++i) {
            BatchRecord& r = recs[i][l];
            if (r.kind == 1) {
                FastIterArgs a = r.it;
                if (a.s.n_shift == 0) a.s.off2_lo = (uint32_t)bases.v[i], a.s.off2_hi = (uint32_t)(bases.v[i] >> 32);   // (unused: kept at relative 0)
                sub_base(a.s.off_lo, a.s.off_hi, bases.v[i]);
                sub_base(a.s.off2_lo, a.s.off2_hi, bases.v[i]);
                std::memcpy(blob.data() + at[l] + (size_t)i * sizeof(FastIterArgs), &a, sizeof(a));
            } else if (r.kind == 4) {
                IterAheadArgs a = r.ia;
                a.r.dbg = nullptr;
                if (a.z.n > 0) sub_base(a.z.off_lo, a.z.off_hi, bases.v[i]);
                else a.z.off_lo = a.z.off_hi = 0;
                if (a.s.n_shift > 0) {
                    sub_base(a.s.off_lo, a.s.off_hi, bases.v[i]);
                    sub_base(a.s.off2_lo, a.s.off2_hi, bases.v[i]);
                } else {
                    a.s.off_lo = a.s.off_hi = a.s.off2_lo = a.s.off2_hi = 0;
                }
                unsigned char* dst = blob.data() + at[l] + (size_t)i * sizeof(IterAheadArgs);
                std::memcpy(dst, &a, sizeof(a));
                // (the struct's tail padding is not carried by its copies: defined here, or every step's block "changes")
                constexpr size_t tail = offsetof(IterAheadArgs, dbg_slot) + sizeof(int);
                std::memset(dst + tail, 0, sizeof(IterAheadArgs) - tail);
            } else {
                MergeNoiseBatchArgs g{};
                g.a = r.m;
                if (r.kind == 3) {
                    g.z1 = r.z1;
                    g.z2 = r.z2;
                    if (g.z1.n > 0) sub_base(g.z1.off_lo, g.z1.off_hi, bases.v[i]);
                    else g.z1.off_lo = g.z1.off_hi = 0;
                    if (g.z2.n > 0) sub_base(g.z2.off_lo, g.z2.off_hi, bases.v[i]);
                    else g.z2.off_lo = g.z2.off_hi = 0;
                    g.z1.off2_lo = g.z1.off2_hi = g.z2.off2_lo = g.z2.off2_hi = 0;
                } else {
                    g.z1.n = g.z2.n = 0;
                }
                std::memcpy(blob.data() + at[l] + (size_t)i * sizeof(MergeNoiseBatchArgs), &g, sizeof(g));
            }
        }
    icem_handle* owner = handles[0];
    BatchCtx* ctx = (BatchCtx*)owner->batch_ctx;
    if (!ctx) {
        ctx = new BatchCtx();
        owner->batch_ctx = ctx;
        owner->batch_ctx_free = batch_ctx_free;
    }
    const int slot = mpc_step % BatchCtx::SLOTS;
    if (ctx->cap[slot] < bytes) {
        if (ctx->dev[slot]) {
            ICEM_HIP_TRY(hipStreamSynchronize(st));   // (launches of an earlier step may still read the old array)
            (void)hipFree(ctx->dev[slot]);
        }
        ctx->dev[slot] = nullptr;
        ctx->cap[slot] = 0;
        ctx->shadow[slot].clear();
        ICEM_HIP_TRY(hipMalloc(&ctx->dev[slot], bytes + 4096));
        ctx->cap[slot] = bytes + 4096;
    }
    if (ctx->shadow[slot].size() != bytes || std::memcmp(ctx->shadow[slot].data(), blob.data(), bytes) != 0) {
        if (opt_i(OPT_AHEAD_STAMPS) && ctx->shadow[slot].size() == bytes) {   // development: which bytes moved
            int shown = 0;
            for (size_t l = 0; l < L && shown < 12; ++l) {
                const size_t one = recs[0][l].kind == 1 ? sizeof(FastIterArgs) : recs[0][l].kind == 4 ? sizeof(IterAheadArgs) : sizeof(MergeNoiseBatchArgs);
                for (size_t o = 0; o < one * (size_t)n && shown < 12; ++o)
                    if (blob[at[l] + o] != ctx->shadow[slot][at[l] + o]) {
                        std::fprintf(stderr, "batch args changed: step %d launch %zu kind %d problem %zu byte %zu\n", mpc_step, l, recs[0][l].kind, o / one, o % one);
                        ++shown;
                        o = (o / 8 + 1) * 8 - 1;
                    }
            }
        }
        // (pageable source: the runtime stages it before returning; ordered behind the earlier steps' launches on `st`)
        ICEM_HIP_TRY(hipMemcpyAsync(ctx->dev[slot], blob.data(), bytes, hipMemcpyHostToDevice, st));
        ctx->shadow[slot] = blob;
        ++owner->batch_uploads;
    }
    // ---- the launches ----
    for (size_t l = 0; l < L; ++l) {
        const BatchRecord& s = recs[0][l];
        const unsigned char* base = (const unsigned char*)ctx->dev[slot] + at[l];
        if (s.kind == 1) launch_sample_rollout_batch(s, (const FastIterArgs*)base, bases, n, st);
        else if (s.kind == 4) launch_iter_ahead_batch(s, (const IterAheadArgs*)base, bases, n, st);
        else launch_merge_batch(s, (const MergeNoiseBatchArgs*)base, bases, n, st);
        ICEM_HIP_TRY(hipGetLastError());
    }
    return ICEM_OK;
}

extern "C" int64_t icem_batch_uploads(const icem_handle* h) { return h ? (int64_t)h->batch_uploads : 0; }

// MpcICem.get_action as one call for a host caller: observation in, executed action (+ its pool's best cost) out.
// The handle owns a small pinned, device-mapped block [obs | action, best cost | flag].  On the f32 fast path the first
// launch reads the observation straight from it (no H2D copy command in front of the step) and a one-thread kernel
// behind the last merge writes the result and a sequence flag into it, which the host polls (no D2H copy commands,
// no stream synchronisation wake-up).  Other configurations stage through the same block with copy commands.
int icem_get_action(icem_handle* h, const icem_plan_buffers* b, int32_t mpc_step, const double* obs_host,
                    double* action_host, double* best_cost_host, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!b || !obs_host || !action_host) return fail(ICEM_E_INVALID, "null argument");
    if (!h->has_model) return fail(ICEM_E_STATE, "icem_set_model / icem_set_cost must be called first");
    hipStream_t st = (hipStream_t)stream;
    const int o = h->obs_dim, d = h->cfg.act_dim;
    const size_t ts = h->tsize;
    constexpr size_t OUT_OFF = ICEM_MAX_OBS_DIM * sizeof(double), FLAG_OFF = OUT_OFF + (ICEM_MAX_ACT_DIM + 1) * sizeof(double);
    if (!h->host_stage) {
        ICEM_HIP_TRY(hipHostMalloc(&h->host_stage, FLAG_OFF + 64, hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(h->host_stage, 0, FLAG_OFF + 64);
        ICEM_HIP_TRY(hipHostGetDevicePointer(&h->host_stage_dev, h->host_stage, 0));
    }
    unsigned char* stage = (unsigned char*)h->host_stage;
    for (int k = 0; k < o; ++k) {
        if (h->cfg.dtype == ICEM_F64) ((double*)stage)[k] = obs_host[k];
        else ((float*)stage)[k] = (float)obs_host[k];
    }
    unsigned char* out = stage + OUT_OFF;
    volatile unsigned* flag = (volatile unsigned*)(stage + FLAG_OFF);
    const bool mapped = h->cfg.dtype == ICEM_F32 && h->use_fast && b->z_r == nullptr && fast_rollout_ok(h, h->cfg.num_elites) &&
                        fast_sample_ok(h);
    icem_plan_buffers bb = *b;
    if (mapped) {
        std::atomic_thread_fence(std::memory_order_release);
        bb.obs0 = h->host_stage_dev;
    } else {
        ICEM_HIP_TRY(hipMemcpyAsync(b->obs0, stage, o * ts, hipMemcpyHostToDevice, st));
    }
    const int rc = icem_plan_step(h, &bb, mpc_step, stream);
    if (rc) return rc;
    if (mapped) {
        // (publishing from inside the last merge kernel instead was tried: no faster than this one-wave launch)
        const unsigned seq = ++h->io_seq;
        unsigned char* dev = (unsigned char*)h->host_stage_dev;
        hipLaunchKernelGGL(publish_result_kernel, dim3(1), dim3(128), 0, st, (const float*)b->executed, (const float*)b->best_cost,
                           (const unsigned*)h->nonfinite_dev, d, (float*)(dev + OUT_OFF), (unsigned*)(dev + FLAG_OFF), seq);
        ICEM_HIP_TRY(hipGetLastError());
        long long spins = 0;
        while (*flag != seq) {
            __builtin_ia32_pause();
            if (++spins > (1ll << 26)) {  // ~ a second: something is wrong on the stream -- let the runtime report it
                ICEM_HIP_TRY(hipStreamSynchronize(st));
                if (*flag != seq) return fail(ICEM_E_HIP, "result flag never arrived");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else {
        ICEM_HIP_TRY(hipMemcpyAsync(out, b->executed, d * ts, hipMemcpyDeviceToHost, st));
        ICEM_HIP_TRY(hipMemcpyAsync(out + (size_t)d * ts, b->best_cost, ts, hipMemcpyDeviceToHost, st));
        ICEM_HIP_TRY(hipMemcpyAsync(out + (size_t)(d + 1) * ts, h->nonfinite_dev, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        ICEM_HIP_TRY(hipStreamSynchronize(st));
    }
    unsigned nonfinite_now = 0;
    std::memcpy(&nonfinite_now, out + (size_t)(d + 1) * ts, sizeof(unsigned));
    for (int j = 0; j <= d; ++j) {
        const double v = h->cfg.dtype == ICEM_F64 ? ((const double*)out)[j] : (double)((const float*)out)[j];
        if (j < d) action_host[j] = v;
        else if (best_cost_host) *best_cost_host = v;
    }
    // Trajectories of THIS step whose cost left a tile kernel non-finite although the observation was finite: a state left
    // the arithmetic's range (f32's, or -- actions outside the bounds the scale was taken from -- the fp16 planes'), where
    // the reference's float64 ranks a finite cost (icem.py:147-159, 199).  The action is returned, and so is the fact.
    const unsigned fresh = nonfinite_now - h->nonfinite_seen;
    h->nonfinite_seen = nonfinite_now;
    if (fresh != 0) {
        bool obs_finite = true;
        for (int k = 0; k < o; ++k) obs_finite = obs_finite && std::isfinite(obs_host[k]);
        if (obs_finite)
            return fail(ICEM_E_RANGE, std::to_string(fresh) + " trajectories of this MPC step came back with a non-finite cost from a finite "
                        "observation: a state left the range of the rollout's arithmetic (icem_tile_arith / icem_tile_growth); the "
                        "reference's float64 ranks finite costs there -- use ICEM_TILE_F32 / dtype f64 for this model");
    }
    return ICEM_OK;
}

}  // extern "C"
