// k_step_xcd.hip -- the WHOLE MPC step of a small population (every iteration at most 4096 rows) as ONE launch: the body of
// MpcICem.get_action (icem/controllers/icem.py:106-189) -- opt_iters x (map the colored noise, roll out, cost, top-K, refit), the
// executed action, the mean shift and the std reset -- with every iteration boundary INSIDE one XCD.
//
// Why.  At the metric's population (N = 4096) a step of the launch-per-iteration path is a chain of six launches, and a launch is
// 5.5 us of merge prologue (two cold round trips: the 256 lists' keys, then the elite rows), 3.9 us of 30 dependent model steps
// and 1.5 us to the next kernel (EXPERIMENTS R5.5).  The 256 tiles of such a population fit the 32 CUs of ONE XCD, eight waves
// each, and an XCD has one coherent L2: what one of its workgroups stores, the others read back from that L2 (loads that bypass
// their own L1: sc1) -- no write-back, no invalidate, no kernel boundary.  tools/ubench/xcd_barrier.hip prices the boundary: a
// 32-arrival barrier on a counter in the XCD's L2 + every member reading all 32 candidate lists = 1.2 us per round.
//
// Roles (256 workgroups of 768 threads, one per CU; a workgroup reads HW_REG_XCC_ID and claims its place):
//   member   the first 32 workgroups that find themselves on XCD 0.  Member m owns rows [128 m, 128 m + 128) of every iteration:
//            raw colored noise of the iteration -> LDS tile (drawn elsewhere: powerlaw_psd_gaussian needs no distribution,
//            icem.py:73-79) -> clip(y std + mean) (icem.py:79) -> pool (HBM) and, straight from the tile, rollout + cost by
//            eight waves (Tile16H / Tile16, rollout_slab: the bits of every other kernel) -> one sorted K-list -> barrier ->
//            EVERY member selects the global top-K from the 32 lists (+ kept / shifted elites), gathers the elite rows and
//            refits (icem.py:194-211: the same code, so the same bits in every member) -> next iteration.  Member 0 keeps the
//            elite set for the caller and runs the step's epilogue (icem.py:163-175).
//   worker   every other workgroup (the 224 CUs of the other XCDs): draws the noise -- this step's iterations 1 .. and the NEXT
//            step's iteration 0 -- chunk by chunk from a queue, written through (sc1 stores) with a counter per iteration the
//            members wait on; the queue's first job builds and rolls out the shifted elites (icem.py:91-104, 131-137).
// Visibility never rests on placement: a member IS on XCD 0 (it read its own XCC_ID), cross-XCD data is sc1-stored and
// sc1-loaded, every wait is bounded (a timeout raises the handle's status word and the host retires the path).
#include "fused_dev.h"

namespace icem {

namespace {

constexpr int SX_MEMBERS = 32, SX_RW = 8, SX_NT = 768, SX_KREG = 12, SX_ROWS = 16 * SX_RW;
// state words (unsigned), zero between launches except BAR (cumulative, only ever touched from XCD 0) and STATUS (sticky)
constexpr int SXS_CLAIM = 0, SXS_HEAD = 8, SXS_SHIFT = 9, SXS_DONE = 10, SXS_READY = 12 /* .. + STEP_XCD_MAX_SEG */, SXS_STATUS = 28,
              SXS_BAR = 32 /* its own 128-byte line */;

__device__ __forceinline__ unsigned sx_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
// bounded wait until *p (read past the L1) has reached `target` (wrap-safe); false: gave up
__device__ __forceinline__ bool sx_wait_ge(const unsigned* p, unsigned target, unsigned max_polls, unsigned* status) {
    unsigned polls = 0;
    while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        if (++polls > max_polls) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

typedef unsigned sx_u32x4 __attribute__((ext_vector_type(4)));

template <int H, int D, int O, int KIND, int ARITH>
__global__ __launch_bounds__(SX_NT) void step_xcd_kernel(StepXcdArgs a) {
    using Tile = typename TileSel<H, D, O, KIND, ARITH>::type;
    constexpr int HD = H * D, KREG = SX_KREG, NT = SX_NT, RW = SX_RW, ROWS = SX_ROWS;
    static_assert(HD % 4 == 0 && NT == ROWS * D, "16-byte rows; one sampling thread per (row, dim) of a chunk");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tilebuf = smem;                                             // [SLACK + ROWS * HD + TAIL]
    float* tile_rows = tilebuf + Tile::SLACK;
    constexpr int TILE_F = ((Tile::SLACK + ROWS * HD + Tile::TAIL + 3) / 4) * 4;
    float* ms = smem + TILE_F;                                         // mean | std
    float* obs_stage = ms + 2 * HD;                                    // [32]
    auto wg_keys = reinterpret_cast<unsigned long long(*)[RW][32]>(obs_stage + 32);   // [2][RW][32]
    unsigned long long* sel = &wg_keys[0][0][0] + 2 * RW * 32;         // [64]
    unsigned long long* cand = sel + 64;                               // [64]
    float* new_mean = reinterpret_cast<float*>(cand + 64);             // [HD] (the epilogue)
    float* lohi = new_mean + HD;                                       // low | high of every (step, dim) element: [2 HD]
    int* ctl = reinterpret_cast<int*>(lohi + 2 * HD);                  // [4]
    long long* wstamp = reinterpret_cast<long long*>(ctl + 4);         // [12] development: when each wave left the rollout / the copy
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned* st = a.state;
    const unsigned xcc = sx_xcc_id();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(st + SXS_CLAIM + (xcc & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[0] = (xcc == 0 && t < SX_MEMBERS) ? (int)t : -1;
    }
    __syncthreads();
    const int member = ctl[0];
    const int iters = a.iters;
    auto leave = [&]() {   // the last workgroup out puts the launch's state back (everything but BAR and STATUS)
        __syncthreads();
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(st + SXS_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == gridDim.x - 1) {
                for (int w = 0; w < SXS_STATUS; ++w) __hip_atomic_store(st + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    // ================================================================================================ worker
    if (member < 0) {
        for (;;) {
            __syncthreads();
            if (tid == 0) ctl[1] = (int)__hip_atomic_fetch_add(st + SXS_HEAD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            int job = ctl[1];
            if (job >= a.n_jobs) break;
            if (a.n_shift > 0 && job == 0) {
                // ---- the shifted elites (icem.py:91-104): elites[e, 1:, :] ++ a last action from the (n_shift, d, h) noise batch of
                // stream base + iters (only t = h - 1 is used), rolled out from the start observation -> rows and costs of their own
                const FastRolloutArgs& ra = a.r;
                const float obs_reg = ra.obs0[(tid < 32 && tid < ra.o) ? tid : 0];
                Tile tile;
                if (wave == 0) tile.load(ra, lane);
                for (int e = tid; e < HD; e += NT) {
                    ms[e] = a.mean[e];
                    ms[HD + e] = a.std[e];
                }
                for (int e = tid; e < 16 * HD; e += NT) tile_rows[e] = 0.f;
                if (tid < 32) obs_stage[tid] = tid < ra.o ? obs_reg : 0.f;
                __syncthreads();
                if (tid < a.n_shift * D) {
                    const int e = tid / D, j = tid - e * D;
                    const float lo = a.low[j], hi = a.high[j];
                    float last = 0.f;
                    sample_row<H, 10>(a.W, (unsigned)e, (unsigned)j, a.shift_off_lo, a.shift_off_hi, a.seed_lo, a.seed_hi,
                                      [&](int t, float y) {
                                          if (t == H - 1) {
                                              float v = __builtin_fmaf(y, ms[HD + t * D + j], ms[t * D + j]);
                                              v = v < lo ? lo : v;
                                              last = v > hi ? hi : v;
                                          }
                                      }, a.white != 0);
                    const float* src = a.shift_src + (size_t)e * HD + j;
                    float* dst = a.shift_rows + (size_t)e * HD + j;
                    float* trow = tile_rows + e * HD + j;
                    for (int t = 0; t < H - 1; ++t) {
                        const float v = src[(t + 1) * D];
                        __hip_atomic_store(dst + t * D, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        trow[t * D] = v;
                    }
                    __hip_atomic_store(dst + (H - 1) * D, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    trow[(H - 1) * D] = last;
                }
                __syncthreads();
                if (wave == 0) {
                    tile.load_obs(obs_stage);
                    typename Tile::State stt;
                    tile.init(stt);
                    const float* rd0 = tile.read_ptr(tilebuf, lane, HD);
#pragma unroll
                    for (int t = 0; t < H; ++t) tile.step(stt, rd0 + t * D);
                    const float cost = tile.cost(stt);
                    const bool live = lane < 16 && (lane & 15) < a.n_shift;
                    if (live) __hip_atomic_store(a.shift_costs + (lane & 15), cost, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    note_nonfinite(ra, cost, live);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(st + SXS_SHIFT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
            }
            // ---- a chunk of raw colored noise: rows [c * 128, ..) of segment s
            job -= a.n_shift > 0 ? 1 : 0;
            int s = 0;
            while (s + 1 < a.n_seg && job >= a.seg_chunk0[s + 1]) ++s;
            const int n_base = (job - a.seg_chunk0[s]) * ROWS;
            const int n_here = cmin(ROWS, a.seg_n[s] - n_base);
            if (tid < n_here * D) {
                const int nl = tid / D, j = tid - nl * D;
                float* trow = tile_rows + nl * HD + j;
                sample_row<H, 10>(a.W, (unsigned)(n_base + nl), (unsigned)j, a.seg_off_lo[s], a.seg_off_hi[s], a.seed_lo, a.seed_hi,
                                  [&](int t, float y) { trow[t * D] = y; }, a.white != 0);
            }
            __syncthreads();
            {   // written through: the readers sit on another XCD
                float* gdst = a.seg_out[s] + (size_t)n_base * HD;
                const int total4 = n_here * (HD / 4);
                auto rs = __builtin_amdgcn_make_buffer_rsrc(gdst, 0, total4 * 16, 0x00020000);
                const sx_u32x4* t4 = reinterpret_cast<const sx_u32x4*>(tile_rows);
                for (int e = tid; e < total4; e += NT) __builtin_amdgcn_raw_buffer_store_b128(t4[e], rs, e * 16, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(st + SXS_READY + s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        leave();
        return;
    }
    // ================================================================================================ member
    FastRolloutArgs ra = a.r;
    const int base = member * ROWS;
    // development (icem_debug_stamps + option ahead_stamps): wall_clock64 stamps of thread 0 of members 0 and 31:
    // [member == 31][iteration][8] behind word 16; [0] = entry, the epilogue's in iteration slot `iters`
    long long* stamps = (a.r.dbg && tid == 0 && (member == 0 || member == SX_MEMBERS - 1)) ? a.r.dbg + 16 + (member ? 128 : 0) : nullptr;
    if (stamps) a.r.dbg[member ? 1 : 0] = wall_clock64();
#define SX_STAMP(it, k) if (stamps) stamps[(it) * 8 + (k)] = wall_clock64();
    const float obs_reg = ra.obs0[(tid < 32 && tid < ra.o) ? tid : 0];
    Tile tile;
    if (wave < RW) tile.load(ra, lane);
    for (int e = tid; e < HD; e += NT) {
        ms[e] = a.mean[e];
        ms[HD + e] = a.std[e];
        lohi[e] = a.low[e % D];
        lohi[HD + e] = a.high[e % D];
    }
    if (tid < 32) obs_stage[tid] = tid < ra.o ? obs_reg : 0.f;
    // this thread's 16-byte vectors of a slab: vector e = tid + k NT holds elements [4 e, 4 e + 4) of row 4 e / HD
    constexpr int NLD = (ROWS * (HD / 4) + NT - 1) / NT;
    int vidx[NLD];   // (4 e) % HD: where the vector's distribution and bounds sit (a row is HD % 4 == 0 floats: no vector straddles rows)
#pragma unroll
    for (int k = 0; k < NLD; ++k) vidx[k] = (4 * (tid + k * NT)) % HD;
    __syncthreads();
    if (wave < RW) tile.load_obs(obs_stage);
    const float* rd0 = tile.read_ptr(tilebuf + (wave < RW ? wave : 0) * 16 * HD, lane, HD);
    MergeSingleArgs m{};
    m.K = a.K, m.h = H, m.d = D, m.n_lists = SX_MEMBERS;
    m.alpha = a.alpha, m.init_std = a.init_std;
    m.low = a.low, m.high = a.high;
    // the merge of iteration `it` (run at the top of iteration it + 1, or -- the last one -- by member 0 at the end)
    auto merge_args = [&](int it) {
        const int cur = (a.g0 + it) & 1;
        m.part_k = a.lists[it & 1];
        m.actions = a.pool[(iters - 1 - it) & 1];
        m.n_pool = m.n_global = a.pop[it];
        m.keep_base = -1;
        if (it == 0 && a.n_shift > 0) {   // the shifted elites: candidates of their own, rows and costs in their own buffers
            m.n_keep = a.n_shift;
            m.elites_cur = a.shift_rows;
            m.elites_cost_cur = a.shift_costs;
        } else {
            m.n_keep = (it > 0 && a.keep) ? a.n_reuse : 0;
            m.elites_cur = a.elites + (size_t)cur * a.K * HD;
            m.elites_cost_cur = a.elites_cost + (size_t)cur * a.K;
        }
        m.elites_next = a.elites + (size_t)(cur ^ 1) * a.K * HD;
        m.elites_cost_next = a.elites_cost + (size_t)(cur ^ 1) * a.K;
    };
    for (int it = 0; it < iters; ++it) {
        const int n_it = a.pop[it];
        const int rows_here = cmin(ROWS, n_it - base) > 0 ? cmin(ROWS, n_it - base) : 0;
        float* pool = a.pool[(iters - 1 - it) & 1];
        SX_STAMP(it, 0)
        // ---- the iteration's raw noise -> registers (on its way to the tile: the map below)
        sx_u32x4 rawv[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) rawv[k] = sx_u32x4{0u, 0u, 0u, 0u};
        if (rows_here > 0) {
            if (a.seg_chunk0[it + 1] > a.seg_chunk0[it] && lane == 0)   // (drawn in this launch: wait for its chunks)
                (void)sx_wait_ge(st + SXS_READY + it, (unsigned)(a.seg_chunk0[it + 1] - a.seg_chunk0[it]), a.max_polls, st + SXS_STATUS);
            const float* src = a.raw[it] + (size_t)base * HD;
            const int total4 = rows_here * (HD / 4);
            auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, total4 * 16, 0x00020000);
            // (all requests first, consumed behind the merge below: one round trip, hidden; a vector past the slab's rows
            //  reads as zeros -- the descriptor's bound)
#pragma unroll
            for (int k = 0; k < NLD; ++k) rawv[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + k * NT) * 16, 0, 16);
        }
        // ---- the previous iteration's merge: every member for itself, out of the XCD's L2
        if (it > 0) {
            merge_args(it - 1);
            if (wave == RW) {
                bool ok = true;
                if (lane == 0) {
                    ok = sx_wait_ge(st + SXS_BAR, a.bar_base + (unsigned)(SX_MEMBERS * it), a.max_polls, st + SXS_STATUS);
                    if (it == 1 && a.n_shift > 0) ok = sx_wait_ge(st + SXS_SHIFT, 1u, a.max_polls, st + SXS_STATUS) && ok;
                }
                (void)ok;
                merge_select<KREG, true, true, 1>(m, lane, cand, sel);
            }
            __syncthreads();
            SX_STAMP(it, 2)
            const float* rows[KREG];
            merge_rows<KREG, false>(m, sel, nullptr, rows);
            for (int e = tid; e < HD; e += NT) {
                float xs[KREG];
#pragma unroll
                for (int r = 0; r < KREG; ++r) xs[r] = __hip_atomic_load(rows[r] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float nm, ns;
                refit_element_regs<float, KREG>(m.K, m.alpha, ms[e], ms[HD + e], xs, nm, ns);
                ms[e] = nm;
                ms[HD + e] = ns;
                if (member == 0) {
#pragma unroll
                    for (int r = 0; r < KREG; ++r)
                        if (r < m.K) m.elites_next[(size_t)r * HD + e] = xs[r];
                }
            }
            if (member == 0 && tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
        }
        __syncthreads();
        SX_STAMP(it, 3)
        // ---- y std + mean, clipped (icem.py:79); row 0 <- mean on the last iteration (icem.py:87-88)
        // (rows past the end come out as clip(mean): rolled out, dropped)
        {
            float4* t4 = reinterpret_cast<float4*>(tile_rows);
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int e = tid + k * NT;
                if (e < ROWS * (HD / 4)) {
                    const float4 mu = *reinterpret_cast<const float4*>(ms + vidx[k]);
                    const float4 sg = *reinterpret_cast<const float4*>(ms + HD + vidx[k]);
                    const float4 lo = *reinterpret_cast<const float4*>(lohi + vidx[k]);
                    const float4 hi = *reinterpret_cast<const float4*>(lohi + HD + vidx[k]);
                    float4 y;
                    __builtin_memcpy(&y, &rawv[k], sizeof(y));
                    float4 v;
                    v.x = __builtin_amdgcn_fmed3f(__builtin_fmaf(y.x, sg.x, mu.x), lo.x, hi.x);
                    v.y = __builtin_amdgcn_fmed3f(__builtin_fmaf(y.y, sg.y, mu.y), lo.y, hi.y);
                    v.z = __builtin_amdgcn_fmed3f(__builtin_fmaf(y.z, sg.z, mu.z), lo.z, hi.z);
                    v.w = __builtin_amdgcn_fmed3f(__builtin_fmaf(y.w, sg.w, mu.w), lo.w, hi.w);
                    t4[e] = v;
                }
            }
        }
        __syncthreads();
        if (a.use_mean && it == iters - 1 && member == 0) {
            for (int e = tid; e < HD; e += NT) tile_rows[e] = ms[e];
            __syncthreads();
        }
        SX_STAMP(it, 4)
        // ---- the tile -> pool (waves behind the rollout waves) beside the rollout
        ra.n_rows = n_it;
        ra.n_cand = n_it;
        ra.actions = pool;
        ra.part_k = a.lists[it & 1];
        if (wave >= RW) {
            const int total4 = rows_here * (HD / 4);
            const float4* t4 = reinterpret_cast<const float4*>(tile_rows);
            float4* g4 = reinterpret_cast<float4*>(pool + (size_t)base * HD);
            for (int e = tid - 64 * RW; e < total4; e += NT - 64 * RW) g4[e] = t4[e];
        }
        unsigned long long run_key = KEY_SENTINEL;
        if (wave < RW) run_key = rollout_slab<Tile, H, D>(tile, ra, rd0, base + wave * 16 + (lane & 15), n_it, run_key, true, lane);
        if (a.r.dbg && lane == 0) wstamp[wave] = wall_clock64();
        wg_merge_emit<RW>(wg_keys, run_key, ra.K, lane, wave, ra, member, SX_MEMBERS);
        if (stamps) {   // [5]: the LAST rollout wave out; [1]: the last copy wave out
            long long r = 0, c = 0;
            for (int w = 0; w < NT / 64; ++w) {
                if (w < RW) r = wstamp[w] > r ? wstamp[w] : r;
                else c = wstamp[w] > c ? wstamp[w] : c;
            }
            stamps[it * 8 + 5] = r;
            stamps[it * 8 + 1] = c;
        }
        SX_STAMP(it, 6)
        // ---- arrive: this member's pool rows, costs, list (and elites) are in the XCD's L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(st + SXS_BAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        SX_STAMP(it, 7)
    }
    // ================================================================================================ the step's last merge + epilogue
    if (member == 0) {
        merge_args(iters - 1);
        if (wave == RW) {
            if (lane == 0) {
                (void)sx_wait_ge(st + SXS_BAR, a.bar_base + (unsigned)(SX_MEMBERS * iters), a.max_polls, st + SXS_STATUS);
                if (iters == 1 && a.n_shift > 0) (void)sx_wait_ge(st + SXS_SHIFT, 1u, a.max_polls, st + SXS_STATUS);
            }
            merge_select<KREG, true, true, 1>(m, lane, cand, sel);
        }
        __syncthreads();
        const float* rows[KREG];
        merge_rows<KREG, false>(m, sel, nullptr, rows);
        for (int e = tid; e < HD; e += NT) {
            float xs[KREG];
#pragma unroll
            for (int r = 0; r < KREG; ++r) xs[r] = __hip_atomic_load(rows[r] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int r = 0; r < KREG; ++r)
                if (r < m.K) m.elites_next[(size_t)r * HD + e] = xs[r];
            float nm, ns;
            refit_element_regs<float, KREG>(m.K, m.alpha, ms[e], ms[HD + e], xs, nm, ns);
            new_mean[e] = nm;
        }
        if (tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
        __syncthreads();
        // executed action (icem.py:163), mean shifted with its last row kept (:167-171, 191-192), std back to its initial value (:175)
        for (int e = tid; e < HD; e += NT) {
            const int j = e % D;
            a.mean_out[e] = (e + D < HD) ? new_mean[e + D] : new_mean[e];
            a.std_out[e] = (a.high[j] - a.low[j]) / 2.f * a.init_std;
        }
        if (tid < D) a.executed[tid] = __hip_atomic_load(rows[0] + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) a.best_cost[0] = key_cost(sel[0]);
        SX_STAMP(iters, 0)
    }
    leave();
    SX_STAMP(iters, 1)
}

}  // namespace

size_t step_xcd_state_bytes() { return 64 * sizeof(unsigned); }
int step_xcd_rows_per_member() { return SX_ROWS; }
int step_xcd_max_rows() { return SX_ROWS * SX_MEMBERS; }

// shapes with a compiled step kernel: one-tile widths, 16-byte action rows, eight tiles + a row of sampling threads in 768 threads
bool step_xcd_supported(int h, int d, int O, int K) {
    bool ok = false;
#define XR(HH, DD, OO) \
    if constexpr (OO <= 20 && (HH * DD) % 4 == 0 && SX_NT == SX_ROWS * DD) ok = ok || (h == HH && d == DD && O == OO);
    ICEM_FAST_SHAPES(XR)
#undef XR
    return ok && K + 1 <= SX_KREG;
}

void launch_step_xcd(const StepXcdArgs& a, int h, int d, int O, int kind, hipStream_t st) {
#define XK(HH, DD, OO, KK, AA)                                                                                                     \
    {                                                                                                                              \
        using Tile = typename TileSel<HH, DD, OO, KK, AA>::type;                                                                   \
        constexpr int HDc = HH * DD;                                                                                               \
        constexpr size_t lds = (size_t)(((Tile::SLACK + SX_ROWS * HDc + Tile::TAIL + 3) / 4) * 4 + 2 * HDc + 32 + HDc + 2 * HDc + 4 + 24) * 4 +   \
                               (size_t)(2 * SX_RW * 32 + 128) * 8;                                                                 \
        hipLaunchKernelGGL((step_xcd_kernel<HH, DD, OO, KK, AA>), dim3(256), dim3(SX_NT), lds, st, a);                             \
        return;                                                                                                                    \
    }
#define XR(HH, DD, OO)                                                        \
    if constexpr (OO <= 20 && (HH * DD) % 4 == 0 && SX_NT == SX_ROWS * DD) { \
        if (h == HH && d == DD && O == OO) {                                  \
            if (a.r.arith == 1) {                                             \
                if (kind == 1) XK(HH, DD, OO, 1, 1) else XK(HH, DD, OO, 0, 1) \
            } else {                                                          \
                if (kind == 1) XK(HH, DD, OO, 1, 0) else XK(HH, DD, OO, 0, 0) \
            }                                                                 \
        }                                                                     \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XK
}

}  // namespace icem
