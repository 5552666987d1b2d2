// k_sample.hip -- K1 for populations too large for the single-launch kernel.
// sample_folded_kernel<H, ROUNDS>: one thread per (trajectory, action-dim) row: per-row stream -> Box-Muller -> the h
//   white draws in registers; inverse real DFT folded on its cos/sin symmetry with the table rows as wave-uniform
//   scalar operands; affine (mean / std staged in LDS) + clip; samples parked in an LDS tile laid out like the
//   [n, h, d] output so the slab leaves as coalesced stores.
// sample_folded_merge_kernel: the same with the PREVIOUS iteration's top-K selection + refit in its prologue; in sharded
//   runs (records variant) workgroup 0 can be the previous iteration's record pack + push ("riding pack") and run the
//   launch's one records merge for everybody ("published merge"): DESIGN.md section 6.
#include "fused_dev.h"

namespace icem {

namespace {

template <int H, int ROUNDS>
__global__ __launch_bounds__(SWG) void sample_folded_kernel(FastSampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int d = a.d;
    const int hd = H * d;
    const int tpw = SWG / d;
    float* ms = smem;            // mean | std
    float* tile = smem + 2 * hd;  // [tpw, hd]
    const int tid = threadIdx.x;
    // mean / std are requested now and parked in LDS BEHIND the draws: the generator -- 60 % of a row's instructions --
    // needs neither, so the cold round trip of these loads hides under it instead of standing in front of it
    float mreg[2], sreg[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * SWG;
        mreg[i] = a.mean[e < hd ? e : 0];
        sreg[i] = a.std[e < hd ? e : 0];
    }
    const int n_base = blockIdx.x * tpw;
    const int n_here = cmin(tpw, a.n - n_base);
    const bool shift_wg = a.n_shift > 0 && blockIdx.x == gridDim.x - 1;
    const bool has_row = !shift_wg && tid < n_here * d;
    const int nl = tid / d;
    const int j = tid - nl * d;
    const float lo = a.low[j < d ? j : 0], hi = a.high[j < d ? j : 0];
    float g[HMAX];
    if (has_row)
        row_normals<H, ROUNDS>((unsigned)(a.first_index + n_base + nl), (unsigned)j, a.off_lo, a.off_hi, a.seed_lo, a.seed_hi, g);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * SWG;
        if (e < hd) {
            ms[e] = mreg[i];
            ms[hd + e] = sreg[i];
        }
    }
    for (int e = tid + 2 * SWG; e < hd; e += SWG) {
        ms[e] = a.mean[e];
        ms[hd + e] = a.std[e];
    }
    __syncthreads();
    if (shift_wg) {
        // the extra workgroup: shifted elites.  Row (e, j) keeps elites[e, 1:, j] and draws its last action
        // from the full (n_shift, d, h) noise batch of stream off2 (only t = h-1 is used, icem.py:102)
        if (tid < a.n_shift * d) {
            const int e = tid / d;
            const int js = tid - e * d;
            const float los = a.low[js], his = a.high[js];
            float last = 0.f;
            sample_row<H, ROUNDS>(a.W, (unsigned)e, (unsigned)js, a.off2_lo, a.off2_hi, a.seed_lo, a.seed_hi,
                                  [&](int t, float y) {
                                      if (t == H - 1) {
                                          float v = __builtin_fmaf(y, ms[hd + t * d + js], ms[t * d + js]);
                                          v = v < los ? los : v;
                                          last = v > his ? his : v;
                                      }
                                  }, a.white != 0);
            float* dst = a.out + (size_t)(a.n + e) * hd + js;
            const float* src = a.elites_src + (size_t)e * hd + js;
            for (int t = 0; t < H - 1; ++t) dst[t * d] = src[(t + 1) * d];
            dst[(H - 1) * d] = last;
        }
        return;
    }
    if (has_row) {
        float* trow = tile + nl * hd + j;
        const float* mrow = ms + j;
        row_synth<H>(a.W, g, [&](int t, float y) {
            const float v = __builtin_fmaf(y, mrow[hd + t * d], mrow[t * d]);
            trow[t * d] = __builtin_amdgcn_fmed3f(v, lo, hi);  // clip in one v_med3_f32
        }, a.white != 0);
    }
    __syncthreads();
    if (a.row0_mean && a.first_index + n_base == 0) {  // icem.py:87-88
        for (int e = tid; e < hd; e += SWG) tile[e] = ms[e];
        __syncthreads();
    }
    float* gdst = a.out + (size_t)n_base * hd;
    const int total = n_here * hd;
    if ((hd & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gdst);
        for (int e = tid; e < total / 4; e += SWG) g4[e] = t4[e];
    } else {
        for (int e = tid; e < total; e += SWG) gdst[e] = tile[e];
    }
}

// -------------------------------------------------------------------------------------------------
// K1 without the distribution ("noise ahead"): the raw colored samples y [n, h, d] of one sampling call
// -------------------------------------------------------------------------------------------------
// powerlaw_psd_gaussian itself (icem.py:73-75) -- 95 % of the sampler's instructions -- needs only (seed, call offset,
// global row): it runs on a second stream while the PREVIOUS iteration's rollout is still busy, and the rollout of
// this iteration applies `* std + mean` and the clip to every vector it loads (Stream16::run_xf).  Same sample_row, so
// the same y; no mean / std staging, no shifted-elite workgroup.
// DT: the action dimension as a compile-time constant for the benchmark shapes (the tile offsets of the 30 stores of a
// row become immediates, the row / dimension split a constant division), 0 = run-time d.
template <int H, int ROUNDS, int DT>
__global__ __launch_bounds__(SWG) void noise_rows_kernel(FastSampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int d = DT > 0 ? DT : a.d;
    const int hd = H * d;
    const int tpw = SWG / d;
    float* tile = smem;  // [tpw, hd]
    const int tid = threadIdx.x;
    const int n_base = blockIdx.x * tpw;
    const int n_here = cmin(tpw, a.n - n_base);
    if (tid < n_here * d) {
        const int nl = tid / d;
        const int j = tid - nl * d;
        float* trow = tile + nl * hd + j;
        sample_row<H, ROUNDS>(a.W, (unsigned)(a.first_index + n_base + nl), (unsigned)j, a.off_lo, a.off_hi, a.seed_lo,
                              a.seed_hi, [&](int t, float y) { trow[t * d] = y; }, a.white != 0);
    }
    __syncthreads();
    float* gdst = a.out + (size_t)n_base * hd;
    const int total = n_here * hd;
    if ((hd & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gdst);
        for (int e = tid; e < total / 4; e += SWG) g4[e] = t4[e];
    } else {
        for (int e = tid; e < total; e += SWG) gdst[e] = tile[e];
    }
}

// -------------------------------------------------------------------------------------------------
// K1 with the PREVIOUS iteration's K3 + K4 in its prologue (populations too large for the single-launch kernel)
// -------------------------------------------------------------------------------------------------
// sample_folded_kernel plus one wavefront per workgroup that runs the low-register selection
// (merge_select_stream) on the previous iteration's candidate lists while the 4 sampling waves draw their noise
// into the LDS tile; then all 5 waves gather the K elite rows and refit, the affine map + clip is applied to the
// tile and the tile leaves as before.  Every workgroup redoes the same merge (L2 serves the 20 KB of keys and 7 KB
// of elite rows), workgroup 0 publishes it.  Saves the merge launch (8.6 + 2.6 us) for ~3 us more sampler time.
template <int H, int ROUNDS, int KREG, bool REC>
__global__ __launch_bounds__(SWG + 64) void sample_folded_merge_kernel(FastSampleMergeArgs args) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ unsigned long long sel[64];
    __shared__ unsigned long long cand[64];
    __shared__ int slot[64];
    constexpr int NTT = SWG + 64;
    const FastSampleArgs& a = args.s;
    const MergeSingleArgs& m = args.m;
    const int d = a.d;
    const int hd = H * d;
    const int tpw = SWG / d;
    float* ms = smem;            // mean | std (computed here)
    float* tile = smem + 2 * hd;  // [tpw, hd]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // riding pack (sharded runs, PackPrev): workgroup 0 packs + pushes the previous iteration's records while the others
    // draw their noise; the launch has one workgroup more
    const bool has_pack = REC && args.p.part_k != nullptr;
    if constexpr (REC) {
        if (has_pack && blockIdx.x == 0) {
            MergeSingleArgs pk{};
            pk.n_lists = args.p.n_lists;
            pk.n_pool = args.p.n_pool;
            pk.n_global = args.p.n_global;
            pk.K = args.p.K;
            pk.h = H;
            pk.d = d;
            pk.part_k = args.p.part_k;
            pk.actions = args.p.actions;
            pk.n_keep = args.p.n_keep;
            pk.elites_cost_cur = args.p.keep_costs;
            pk.keep_base = args.p.n_loc;
            // everybody else's prologue waits for this workgroup's push: its waves go first on their SIMDs
            __builtin_amdgcn_s_setprio(3);
            if (tid >= SWG) merge_select_shallow<3>(pk, lane, cand, sel);
            __syncthreads();
            pack_records_body<KREG>(pk, args.p.n_loc, args.p.shard_lo, args.p.records, args.p.px, smem, sel, tid, NTT);
            if (args.p.pub != nullptr) {
                // Published merge: this workgroup also runs THE merge of the launch -- waits for every rank's records,
                // selects, gathers, refits -- and publishes the new mean | std to all the others (written through: the
                // XCDs' L2s are not coherent inside a launch; then one agent-scope flag).  Hundreds of workgroups each
                // reading the same 7 KB of records from uncached memory was the slowest part of this launch.
                __syncthreads();
                merge_select_records_wg(m, tid >= SWG, lane, tid, NTT, sel, slot);
                const float* rows[KREG];
                merge_rows<KREG, true>(m, sel, slot, rows);
                for (int e = tid; e < hd; e += NTT) {
                    float xs[KREG];
#pragma unroll
                    for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
                    float nm, ns;
                    refit_element_regs<float, KREG>(m.K, m.alpha, m.mean[e], m.std[e], xs, nm, ns);
                    __hip_atomic_store(args.p.pub + e, nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(args.p.pub + hd + e, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    m.mean_out[e] = nm;
                    m.std_out[e] = ns;
#pragma unroll
                    for (int r = 0; r < KREG; ++r)
                        if (r < m.K) m.elites_next[(size_t)r * hd + e] = xs[r];
                }
                if (tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(args.p.pub_flag, args.p.pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
    }
    const bool published = REC && has_pack && args.p.pub != nullptr;
    const int wg = blockIdx.x - (has_pack ? 1 : 0);
    const int n_base = wg * tpw;
    const int n_here = cmin(tpw, a.n - n_base);
    const bool has_row = tid < n_here * d;
    const int nl = tid / d;
    const int j = tid - nl * d;
    float* trow = tile + nl * hd + j;
    if (tid >= SWG) {
        if (published) {  // workgroup 0 merges for everybody: wait for its flag (bounded like every exchange wait)
            unsigned polls = 0;
            while (__hip_atomic_load(args.p.pub_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != args.p.pub_seq &&
                   ++polls <= args.m.xw.max_polls)
                __builtin_amdgcn_s_sleep(16);
            if (polls > args.m.xw.max_polls && lane == 0 && args.m.xw.status)
                __hip_atomic_store(args.m.xw.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        } else if constexpr (!REC) {
            merge_select_shallow<3>(m, lane, cand, sel);
        }
    } else if (has_row) {
        sample_row<H, ROUNDS>(a.W, (unsigned)(a.first_index + n_base + nl), (unsigned)j, a.off_lo, a.off_hi, a.seed_lo,
                              a.seed_hi, [&](int t, float y) { trow[t * d] = y; }, a.white != 0);
    }
    if (REC && !published) merge_select_records_wg(m, tid >= SWG, lane, tid, NTT, sel, slot);   // (the records' keys ranked by all threads)
    else __syncthreads();
    if (published) {
        for (int e = tid; e < 2 * hd; e += NTT) ms[e] = __hip_atomic_load(args.p.pub + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        const float* rows[KREG];
        merge_rows<KREG, REC>(m, sel, slot, rows);
        for (int e = tid; e < hd; e += NTT) {
            float xs[KREG];
#pragma unroll
            for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
            float nm, ns;
            refit_element_regs<float, KREG>(m.K, m.alpha, m.mean[e], m.std[e], xs, nm, ns);
            ms[e] = nm;
            ms[hd + e] = ns;
            if (wg == 0) {
                m.mean_out[e] = nm;
                m.std_out[e] = ns;
#pragma unroll
                for (int r = 0; r < KREG; ++r)
                    if (r < m.K) m.elites_next[(size_t)r * hd + e] = xs[r];
            }
        }
        if (wg == 0 && tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
    }
    __syncthreads();
    if (has_row) {
        const float lo = a.low[j], hi = a.high[j];
        const float* mrow = ms + j;
        for (int t = 0; t < H; ++t) {
            const float v = __builtin_fmaf(trow[t * d], mrow[hd + t * d], mrow[t * d]);
            trow[t * d] = __builtin_amdgcn_fmed3f(v, lo, hi);
        }
    }
    __syncthreads();
    if (a.row0_mean && a.first_index + n_base == 0) {  // icem.py:87-88
        for (int e = tid; e < hd; e += NTT) tile[e] = ms[e];
        __syncthreads();
    }
    float* gdst = a.out + (size_t)n_base * hd;
    const int total = n_here * hd;
    if ((hd & 3) == 0) {
        const float4* t4 = reinterpret_cast<const float4*>(tile);
        float4* g4 = reinterpret_cast<float4*>(gdst);
        for (int e = tid; e < total / 4; e += NTT) g4[e] = t4[e];
    } else {
        for (int e = tid; e < total; e += NTT) gdst[e] = tile[e];
    }
}

}  // namespace

bool fast_sample_supported(int h, int d) {
    if (d > SWG) return false;
#define X(HH) \
    if (h == HH) return true;
    ICEM_FAST_HORIZONS(X)
#undef X
    return false;
}

// sampler with the previous iteration's merge in its prologue (default generator only, K <= 11, no shifted elites)
bool sample_folded_merge_ok(int h, int d, int rounds, int K) {
    const int on = opt_i(OPT_MERGE_PROLOGUE);
    return on && rounds == 10 && K + 1 <= 12 && fast_sample_supported(h, d);
}

bool sample_folded_pack_ok(int h, int d, int rounds, int K) {
    const int on = opt_i(OPT_RIDING_PACK);
    const int tpw = SWG / d;
    return on && sample_folded_merge_ok(h, d, rounds, K) && K * (h * d + 2) <= (2 + tpw) * h * d;
}

void launch_sample_folded_merge(const FastSampleMergeArgs& a, hipStream_t st) {
    const int tpw = SWG / a.s.d;
    const int grid = (a.s.n + tpw - 1) / tpw + (a.m.records && a.p.part_k ? 1 : 0);  // + workgroup 0: the riding pack
    const size_t lds = ((size_t)2 * a.s.h * a.s.d + (size_t)tpw * a.s.h * a.s.d) * sizeof(float);
#define X(HH)                                                                                                  \
    if (a.s.h == HH) {                                                                                         \
        if (a.m.records)                                                                                       \
            hipLaunchKernelGGL((sample_folded_merge_kernel<HH, 10, 12, true>), dim3(grid), dim3(SWG + 64), lds, st, a);  \
        else                                                                                                   \
            hipLaunchKernelGGL((sample_folded_merge_kernel<HH, 10, 12, false>), dim3(grid), dim3(SWG + 64), lds, st, a); \
        return;                                                                                                \
    }
    ICEM_FAST_HORIZONS(X)
#undef X
}

void launch_noise_rows(const FastSampleArgs& a, int rounds, hipStream_t st) {
    if (a.n <= 0) return;
    const int tpw = SWG / a.d;
    const int grid = (a.n + tpw - 1) / tpw;
    // The LDS request also bounds how many of these workgroups a CU takes (the tile is 30 KB at d = 6: five would fit and
    // leave a rollout workgroup of the noise-ahead pipeline -- 37 KB -- no room): at least 40.5 KB, i.e. three per CU.
    const size_t min_lds = (size_t)((opt(OPT_AHEAD_NOISE_LDS_KB) >= 0.0 ? opt(OPT_AHEAD_NOISE_LDS_KB) : 40.5) * 1024.0);   // (option ahead_noise_lds_kb: development)
    const size_t lds = std::max((size_t)tpw * a.h * a.d * sizeof(float), min_lds);
#define XS(HH, DD, OO)                                                                                \
    if (a.h == HH && a.d == DD && rounds == 10) {                                                     \
        hipLaunchKernelGGL((noise_rows_kernel<HH, 10, DD>), dim3(grid), dim3(SWG), lds, st, a);       \
        return;                                                                                       \
    }
    ICEM_FAST_SHAPES(XS)
#undef XS
#define X(HH)                                                                                         \
    if (a.h == HH) {                                                                                  \
        if (rounds == 7)                                                                              \
            hipLaunchKernelGGL((noise_rows_kernel<HH, 7, 0>), dim3(grid), dim3(SWG), lds, st, a);     \
        else                                                                                          \
            hipLaunchKernelGGL((noise_rows_kernel<HH, 10, 0>), dim3(grid), dim3(SWG), lds, st, a);    \
        return;                                                                                       \
    }
    ICEM_FAST_HORIZONS(X)
#undef X
}

void launch_sample_folded(const FastSampleArgs& a, int rounds, hipStream_t st) {
    const int tpw = SWG / a.d;
    const int grid = (a.n + tpw - 1) / tpw + (a.n_shift > 0 ? 1 : 0);
    const size_t lds = ((size_t)2 * a.h * a.d + (size_t)tpw * a.h * a.d) * sizeof(float);
#define X(HH)                                                                                        \
    if (a.h == HH) {                                                                                 \
        if (rounds == 7)                                                                             \
            hipLaunchKernelGGL((sample_folded_kernel<HH, 7>), dim3(grid), dim3(SWG), lds, st, a);    \
        else                                                                                         \
            hipLaunchKernelGGL((sample_folded_kernel<HH, 10>), dim3(grid), dim3(SWG), lds, st, a);   \
        return;                                                                                      \
    }
    ICEM_FAST_HORIZONS(X)
#undef X
}

}  // namespace icem
