// k_merge.hip -- K3 + K4 as a launch of its own: merge_single_kernel (global sorted top-K of the candidate lists or
// all-gathered records, elite gather, refit, last-iteration epilogue; one workgroup, wave 0 selects) and
// pack_records_kernel (sharded runs: this rank's K best as records for the exchange).
#include "fused_dev.h"
#include "refit.h"

namespace icem {

namespace {

template <int KREG, bool REC>
__device__ __forceinline__ void merge_single_body(const MergeSingleArgs& a, unsigned char* smem_raw) {
    __shared__ unsigned long long sel[64];
    __shared__ unsigned long long cand[64];
    __shared__ int slot[64];
    float* new_mean = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int hd = a.h * a.d;
    if (a.dbg && threadIdx.x == 0) a.dbg[0] = wall_clock64();
    // old mean/std of this thread's elements: issued now, consumed after the selection
    constexpr int EPL = 4;
    const bool pre = hd <= MERGE_WG * EPL;
    float om[EPL], os[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
        const int e = tid + i * MERGE_WG;
        om[i] = (pre && e < hd) ? a.mean[e] : 0.f;
        os[i] = (pre && e < hd) ? a.std[e] : 0.f;
    }
    if constexpr (REC) {
        merge_select_records_wg(a, tid < 64, lane, tid, MERGE_WG, sel, slot);   // (all threads: the records' keys ranked by counting)
        if (a.dbg && threadIdx.x == 0) a.dbg[4] = wall_clock64();
    } else {
        if (tid < 64) merge_select_shallow<3>(a, lane, cand, sel);
        if (a.dbg && threadIdx.x == 0) a.dbg[4] = wall_clock64();
        __syncthreads();
    }
    // ---- all 4 waves: gather + refit (icem.py:201-211); row pointers first, then all K loads in flight ----
    const float* rows[KREG];
    merge_rows<KREG, REC>(a, sel, slot, rows);
    auto finish_one = [&](int e, float old_mean, float old_std) {
        float xs[KREG];
#pragma unroll
        for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
#pragma unroll
        for (int r = 0; r < KREG; ++r)
            if (r < a.K) a.elites_next[(size_t)r * hd + e] = xs[r];
        float nm, ns;
        refit_element_regs<float, KREG>(a.K, a.alpha, old_mean, old_std, xs, nm, ns);
        if (!a.last) {
            a.mean_out[e] = nm;
            a.std_out[e] = ns;
        } else {
            new_mean[e] = nm;
        }
    };
    if (pre) {
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int e = tid + i * MERGE_WG;
            if (e < hd) finish_one(e, om[i], os[i]);
        }
    } else {
        for (int e = tid; e < hd; e += MERGE_WG) finish_one(e, a.mean[e], a.std[e]);
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[5] = wall_clock64();
    if (tid < a.K) a.elites_cost_next[tid] = key_cost(sel[tid]);
    if (a.last) {
        __syncthreads();
        for (int e = tid; e < hd; e += MERGE_WG) {
            const int j = e % a.d;
            a.mean_out[e] = (e + a.d < hd) ? new_mean[e + a.d] : new_mean[e];
            a.std_out[e] = (a.high[j] - a.low[j]) / 2.f * a.init_std;
        }
        if (tid < a.d) a.executed[tid] = rows[0][tid];
        if (tid == 0) a.best_cost[0] = key_cost(sel[0]);
    }
    if (a.dbg && threadIdx.x == 0) a.dbg[6] = wall_clock64();
}

template <int KREG, bool REC>
__global__ __launch_bounds__(MERGE_WG) void merge_single_kernel(MergeSingleArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    merge_single_body<KREG, REC>(a, smem_raw);
}

// The last merge of an MPC step with company (noise-ahead pipeline, plan.hip): workgroup 0 is merge_single_kernel, the
// other workgroups draw raw colored noise (noise_rows_kernel's work: sample_row into an LDS tile, coalesced copy-out) for
// iteration 0 of the NEXT MPC step -- the one launch of a step during which 255 of the 256 CUs had nothing to do.
template <int H, int KREG>
__global__ __launch_bounds__(MERGE_WG) void merge_noise_kernel(MergeSingleArgs a, FastSampleArgs z1, FastSampleArgs z2, int wgs1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (blockIdx.x == 0) {
        merge_single_body<KREG, false>(a, smem_raw);
        return;
    }
    // workgroups 1 .. wgs1: the sampling call z1; behind them: z2 (a few rows of another stream)
    const bool second = (int)blockIdx.x > wgs1;
    const FastSampleArgs& z = second ? z2 : z1;
    float* tile = reinterpret_cast<float*>(smem_raw);
    const int d = z.d, hd = H * d, tpw = MERGE_WG / d, tid = threadIdx.x;
    const int n_base = ((int)blockIdx.x - 1 - (second ? wgs1 : 0)) * tpw;
    const int n_here = cmin(tpw, z.n - n_base);
    if (tid < n_here * d) {
        const int nl = tid / d;
        const int j = tid - nl * d;
        float* trow = tile + nl * hd + j;
        sample_row<H, 10>(z.W, (unsigned)(z.first_index + n_base + nl), (unsigned)j, z.off_lo, z.off_hi, z.seed_lo, z.seed_hi,
                          [&](int t, float y) { trow[t * d] = y; }, z.white != 0);
    }
    __syncthreads();
    float* gdst = z.out + (size_t)n_base * hd;
    const int total = n_here * hd;
    if ((hd & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gdst);
        for (int e = tid; e < total / 4; e += MERGE_WG) g4[e] = t4[e];
    } else {
        for (int e = tid; e < total; e += MERGE_WG) gdst[e] = tile[e];
    }
}

// The same launch for B problems at once (icem_plan_step_batch): blockIdx.y = the problem, its arguments in device memory,
// the noise calls' stream offsets relative to the step's base of that problem.  z1.n == 0 for every problem: merge only.
template <int H, int KREG>
__global__ __launch_bounds__(MERGE_WG) void merge_noise_batch_kernel(const MergeNoiseBatchArgs* __restrict__ args, BatchBases bases, int wgs1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const MergeNoiseBatchArgs& g = args[blockIdx.y];
    if (blockIdx.x == 0) {
        const MergeSingleArgs a = from_device(g.a);
        merge_single_body<KREG, false>(a, smem_raw);
        return;
    }
    const bool second = (int)blockIdx.x > wgs1;
    const FastSampleArgs z = from_device(second ? g.z2 : g.z1);
    float* tile = reinterpret_cast<float*>(smem_raw);
    const int d = z.d, hd = H * d, tpw = MERGE_WG / d, tid = threadIdx.x;
    const int n_base = ((int)blockIdx.x - 1 - (second ? wgs1 : 0)) * tpw;
    const int n_here = cmin(tpw, z.n - n_base);
    if (n_here <= 0) return;
    uint32_t off_lo = z.off_lo, off_hi = z.off_hi;
    add_base64(off_lo, off_hi, bases.v[blockIdx.y]);
    if (tid < n_here * d) {
        const int nl = tid / d;
        const int j = tid - nl * d;
        float* trow = tile + nl * hd + j;
        sample_row<H, 10>(z.W, (unsigned)(z.first_index + n_base + nl), (unsigned)j, off_lo, off_hi, z.seed_lo, z.seed_hi,
                          [&](int t, float y) { trow[t * d] = y; }, z.white != 0);
    }
    __syncthreads();
    float* gdst = z.out + (size_t)n_base * hd;
    const int total = n_here * hd;
    if ((hd & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gdst);
        for (int e = tid; e < total / 4; e += MERGE_WG) g4[e] = t4[e];
    } else {
        for (int e = tid; e < total; e += MERGE_WG) gdst[e] = tile[e];
    }
}

// Sharded runs: this rank's K best candidates (same selection) packed as records (pack_records_body) -- a launch of
// its own where the pack cannot ride in the next iteration's launch (sample_rollout_kernel's workgroup 0).
// bytes of LDS the pack kernel may use to stage the K records for the push (larger records: separate push launch)
constexpr size_t PACK_STAGE_MAX = 48 * 1024;

template <int KREG>
__global__ __launch_bounds__(MERGE_WG) void pack_records_kernel(MergeSingleArgs a, int n_loc, int shard_lo, float* records, XchgPush px) {
    __shared__ unsigned long long sel[64];
    __shared__ unsigned long long cand[64];
    extern __shared__ __attribute__((aligned(16))) float stage[];  // [K, rs] when the kernel also pushes
    const int tid = threadIdx.x;
    if (tid < 64) merge_select<KREG>(a, tid, cand, sel);
    __syncthreads();
    pack_records_body<KREG>(a, n_loc, shard_lo, records, px, stage, sel, tid, MERGE_WG);
}

// ... and the step's LAST pack together with the records merge that waits for it (sharded runs): one launch instead of two
// one-workgroup launches back to back -- pack + push, then the wait for every rank's flag, selection over the gathered
// records, refit and the epilogue.  The dynamic LDS is the pack's stage first, the merge's new mean afterwards.
template <int KREG>
__global__ __launch_bounds__(MERGE_WG) void pack_merge_kernel(MergeSingleArgs pk, int n_loc, int shard_lo, float* records, XchgPush px,
                                                               MergeSingleArgs m) {
    __shared__ unsigned long long sel[64];
    __shared__ unsigned long long cand[64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    if (tid < 64) merge_select<KREG>(pk, tid, cand, sel);
    __syncthreads();
    pack_records_body<KREG>(pk, n_loc, shard_lo, records, px, reinterpret_cast<float*>(smem_raw), sel, tid, MERGE_WG);
    __syncthreads();
    merge_single_body<KREG, true>(m, smem_raw);
}

// icem_topk_sorted for small f32 pools (the stage-wise controller paths: learned dynamics, host models, the CEM
// baselines) in ONE launch of one workgroup instead of the generic partial + final pair (12.4 + 9.2 us at n = 1 024):
// every wave keeps a running sorted top-K over its 64-key batches (batch sort, running list parked in lanes 32.., one
// more sort), the 16 lists meet in wg_merge_emit's tree.  Same keys as everywhere: (cost, index) order, NaN = +inf,
// (+inf, INT_MAX) padding behind the n real entries.
__global__ __launch_bounds__(1024) void topk_small_kernel(const float* costs, int n, int K, float* out_c, int* out_i) {
    __shared__ unsigned long long wg_keys[2][16][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long run = KEY_SENTINEL;
    bool first = true;
    for (int base = wave * 64; base < n; base += 1024) {
        const int i = base + lane;
        unsigned long long key = wave_sort64(i < n ? make_key(costs[i], i) : KEY_SENTINEL, lane);
        if (!first) {
            const unsigned long long prev = __shfl(run, lane - 32, 64);
            key = (lane >= 32 && lane < 32 + K) ? prev : (lane < K ? key : KEY_SENTINEL);
            key = wave_sort64(key, lane);
        }
        run = key;
        first = false;
    }
    FastRolloutArgs fr{};  // wg_merge_emit only looks at the candidate outputs: list 0 of 1 = the result
    fr.part_c = out_c;
    fr.part_i = out_i;
    wg_merge_emit<16>(wg_keys, run, K, lane, wave, fr, 0, 1);
}

// icem_update_distribution: the reference's update_distributions in ONE launch for small f32 pools -- the sorted top-K
// over the pool's costs and the kept elites' (icem.py:143-145: appended behind the pool, index n + e), the gather of the
// K elite rows from pool / kept elites, the refit (refit.h: the arithmetic of gather_refit_kernel).  Phase 1 + 2 are
// topk_small_kernel's; the selection lands in LDS (wg_merge_emit's list 0 of 1, through a generic pointer).
__global__ __launch_bounds__(1024) void update_small_kernel(UpdateSmallArgs a) {
    __shared__ unsigned long long wg_keys[2][16][32];
    __shared__ unsigned long long sel[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_all = a.n + a.n_keep;
    unsigned long long run = KEY_SENTINEL;
    bool first = true;
    for (int base = wave * 64; base < n_all; base += 1024) {
        const int i = base + lane;
        unsigned long long key = KEY_SENTINEL;
        if (i < n_all) key = make_key(i < a.n ? a.costs[i] : a.keep_costs[i - a.n], i);
        key = wave_sort64(key, lane);
        if (!first) {
            const unsigned long long prev = __shfl(run, lane - 32, 64);
            key = (lane >= 32 && lane < 32 + a.K) ? prev : (lane < a.K ? key : KEY_SENTINEL);
            key = wave_sort64(key, lane);
        }
        run = key;
        first = false;
    }
    FastRolloutArgs fr{};
    fr.part_k = sel;
    wg_merge_emit<16>(wg_keys, run, a.K, lane, wave, fr, 0, 1);
    __syncthreads();
    if (tid < a.K) {
        a.elite_costs_out[tid] = key_cost(sel[tid]);
        a.idx_out[tid] = key_idx(sel[tid]);
    }
    // a padded selection (fewer than K candidates: (+inf, INT_MAX)) repeats its best row, as icem_gather_refit does
    auto row = [&](int r) -> const float* {
        int i = key_idx(sel[r]);
        if (i < 0 || i == INT_MAX) i = key_idx(sel[0]);
        return i < a.n ? a.pool + (size_t)i * a.hd : a.keep_actions + (size_t)(i - a.n) * a.hd;
    };
    for (int e = tid; e < a.hd; e += 1024) {
        for (int r = 0; r < a.K; ++r) a.elites_out[(size_t)r * a.hd + e] = row(r)[e];
        float nm, ns;
        refit_element<float>(a.K, a.alpha, a.mean[e], a.std[e], [&](int r) { return row(r)[e]; }, nm, ns);
        a.mean[e] = nm;
        a.std[e] = ns;
    }
}

}  // namespace

void launch_update_small(const UpdateSmallArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(update_small_kernel, dim3(1), dim3(1024), 0, st, a);
}

bool topk_small_ok(int n, int K) { return n >= 1 && n <= 16384 && K >= 1 && K <= 32; }

void launch_topk_small(const float* costs, int n, int K, float* out_c, int* out_i, hipStream_t st) {
    hipLaunchKernelGGL(topk_small_kernel, dim3(1), dim3(1024), 0, st, costs, n, K, out_c, out_i);
}

bool pack_can_push(int K, int h, int d) { return (size_t)K * (h * d + 2) * sizeof(float) <= PACK_STAGE_MAX; }

void launch_pack_records(const MergeSingleArgs& a, int n_loc, int shard_lo, float* records, hipStream_t st, const XchgPush& px) {
    const size_t lds = px.peers ? (size_t)a.K * (a.h * a.d + 2) * sizeof(float) : 0;
    if (a.K + 1 <= 12)
        hipLaunchKernelGGL((pack_records_kernel<12>), dim3(1), dim3(MERGE_WG), lds, st, a, n_loc, shard_lo, records, px);
    else
        hipLaunchKernelGGL((pack_records_kernel<34>), dim3(1), dim3(MERGE_WG), lds, st, a, n_loc, shard_lo, records, px);
}

void launch_pack_merge(const MergeSingleArgs& pk, int n_loc, int shard_lo, float* records, const XchgPush& px, const MergeSingleArgs& m,
                       hipStream_t st) {
    const size_t lds = std::max(px.peers ? (size_t)pk.K * (pk.h * pk.d + 2) : (size_t)0, (size_t)m.h * m.d) * sizeof(float);
    if (pk.K + 1 <= 12)
        hipLaunchKernelGGL((pack_merge_kernel<12>), dim3(1), dim3(MERGE_WG), lds, st, pk, n_loc, shard_lo, records, px, m);
    else
        hipLaunchKernelGGL((pack_merge_kernel<34>), dim3(1), dim3(MERGE_WG), lds, st, pk, n_loc, shard_lo, records, px, m);
}

// lists form, K <= 11, default generator, a compiled sampler horizon (else: launch_merge_single + launch_noise_rows)
bool merge_noise_ok(const MergeSingleArgs& a, int rounds) {
    return a.records == nullptr && a.K + 1 <= 12 && rounds == 10 && fast_sample_supported(a.h, a.d);
}

void launch_merge_noise(const MergeSingleArgs& a, const FastSampleArgs& z, const FastSampleArgs& z2, hipStream_t st) {
    if (g_batch.rec) {   // icem_plan_step_batch: recorded, launched for all problems at once (launch_merge_batch)
        BatchRecord r;
        r.kind = 3;
        r.m = a;
        r.z1 = z;
        r.z2 = z2;
        g_batch.rec->push_back(r);
        return;
    }
    const int tpw = MERGE_WG / z.d;
    const int wgs1 = (z.n + tpw - 1) / tpw, wgs2 = z2.n > 0 ? (z2.n + tpw - 1) / tpw : 0;
    const int grid = 1 + wgs1 + wgs2;
    const size_t lds = std::max((size_t)a.h * a.d, (size_t)tpw * z.h * z.d) * sizeof(float);
#define X(HH)                                                                                                    \
    if (a.h == HH) {                                                                                             \
        hipLaunchKernelGGL((merge_noise_kernel<HH, 12>), dim3(grid), dim3(MERGE_WG), lds, st, a, z, z2, wgs1);   \
        return;                                                                                                  \
    }
    ICEM_FAST_HORIZONS(X)
#undef X
}

// n problems' last merges (+ the next step's first noise, kind 3) in one launch; the lists form with K <= 11 only
void launch_merge_batch(const BatchRecord& s, const MergeNoiseBatchArgs* args_dev, const BatchBases& bases, int n, hipStream_t st) {
    const MergeSingleArgs& a = s.m;
    int wgs1 = 0, wgs2 = 0, tpw = 1;
    if (s.kind == 3) {
        tpw = MERGE_WG / s.z1.d;
        wgs1 = (s.z1.n + tpw - 1) / tpw;
        wgs2 = s.z2.n > 0 ? (s.z2.n + tpw - 1) / tpw : 0;
    }
    const dim3 grid(1 + wgs1 + wgs2, n);
    const size_t lds = std::max((size_t)a.h * a.d, s.kind == 3 ? (size_t)tpw * s.z1.h * s.z1.d : (size_t)0) * sizeof(float);
#define X(HH)                                                                                                              \
    if (a.h == HH) {                                                                                                       \
        hipLaunchKernelGGL((merge_noise_batch_kernel<HH, 12>), grid, dim3(MERGE_WG), lds, st, args_dev, bases, wgs1);      \
        return;                                                                                                            \
    }
    ICEM_FAST_HORIZONS(X)
#undef X
}

void launch_merge_single(const MergeSingleArgs& a, hipStream_t st) {
    if (g_batch.rec) {
        if (a.records || a.K + 1 > 12) {
            g_batch.unsupported = true;
            return;
        }
        BatchRecord r;
        r.kind = 2;
        r.m = a;
        g_batch.rec->push_back(r);
        return;
    }
    const size_t lds = (size_t)a.h * a.d * sizeof(float);
    if (a.records) {
        if (a.K + 1 <= 12)
            hipLaunchKernelGGL((merge_single_kernel<12, true>), dim3(1), dim3(MERGE_WG), lds, st, a);
        else
            hipLaunchKernelGGL((merge_single_kernel<34, true>), dim3(1), dim3(MERGE_WG), lds, st, a);
    } else if (a.K + 1 <= 12) {
        hipLaunchKernelGGL((merge_single_kernel<12, false>), dim3(1), dim3(MERGE_WG), lds, st, a);
    } else {
        hipLaunchKernelGGL((merge_single_kernel<34, false>), dim3(1), dim3(MERGE_WG), lds, st, a);
    }
}

}  // namespace icem
