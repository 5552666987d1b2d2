// k_iter_small.hip -- K1 + K2 + K3 in one launch for small populations (+ the PREVIOUS iteration's K3 + K4 in its
// prologue): sample_rollout_kernel, one slab of 16 * RW trajectories per workgroup.
#include "fused_dev.h"

namespace icem {

namespace {

// -------------------------------------------------------------------------------------------------
// K1 + K2 + K3 in one launch for small populations (+ the PREVIOUS iteration's K3 + K4 in its prologue)
// -------------------------------------------------------------------------------------------------
// With a few thousand trajectories the chip is mostly empty and an iteration is a chain of latencies: launch,
// prologue, one thread's RNG -> DFT chain, HBM round trip of the actions, launch, prologue, 30 dependent model
// steps, launch, merge.  Here a workgroup samples 16 * RW trajectories into an LDS tile (one thread per
// (trajectory, dim) row, same code as sample_folded_kernel), writes the tile to HBM for the elite gather, and its
// first RW waves roll the trajectories out straight from the tile (same code as rollout16_kernel): one launch, no
// HBM round trip.
// KREG > 0 ("merge prologue"): the distribution this iteration samples from is not in memory yet -- every
// workgroup computes it itself from the previous iteration's candidate lists.  An extra wavefront runs the
// selection (merge_select) WHILE the sampling waves draw their noise (which does not depend on mean / std: the
// raw colored samples are parked in the tile); then all threads gather the K elite rows and refit (same
// arithmetic as merge_single_kernel, so every workgroup gets the same bits), the affine map + clip is applied to
// the tile, and the iteration proceeds as above.  Workgroup 0 also writes the new distribution and elite set for
// the host / the next launch.  That removes the merge launch and hides its latency behind the sampling.  The
// previous pool, lists and distribution are read while this launch writes new ones: all three are ping-pong
// buffers (icem_plan_step).
// REC + PackPrev ("riding pack", sharded runs): workgroup 0 of the launch is not a slab but the PREVIOUS iteration's
// record pack + push (pack_records_body), running while the slabs' workgroups draw their noise; DESIGN section 6.
// QS ("quad sampling", RW <= 2): four lanes per (trajectory, dim) row (row_normals_quad / row_synth_quad): with at most
// two tiles per CU the sampling phase is one thread's instruction chain, and a quarter of it is ~2.4x shorter.
constexpr bool sr_quad(int d, int rw) { return rw <= 2 && ((4 * 16 * rw * d + 63) / 64) * 64 + 64 <= 1024; }
constexpr int sr_threads(int d, int rw) { return (((sr_quad(d, rw) ? 4 : 1) * 16 * rw * d + 63) / 64) * 64; }

// ARITH (FastRolloutArgs::arith): 1 = the model step on the 16-bit matrix cores (Tile16H) -- there is no VALU twin of
// that arithmetic, so the slab is rolled out by RW waves of Tile16H whatever the sampling shape.
// (the body is a device function: sample_rollout_kernel reads its argument block from the kernel-argument segment,
//  sample_rollout_batch_kernel -- B problems in one launch, icem_plan_step_batch -- from an array in device memory)
template <int H, int D, int O, int KIND, int ROUNDS, int RW, int KREG, bool REC, int ARITH>
__device__ __forceinline__ void sample_rollout_body(const FastSampleArgs& sa, const FastRolloutArgs& ra, const MergeSingleArgs& am,
                                                    const PackPrev& ap) {
    constexpr bool PM = KREG > 0;
    constexpr bool QS = sr_quad(D, RW);
    // T4: with quad sampling there are >= 4 * RW wavefronts in the workgroup anyway: the rollout runs on Tile4 (VALU +
    // DPP row broadcast, four trajectories per wave, 4 * RW waves) instead of Tile16 (RW waves): same bits, a shorter
    // latency chain per model step
    constexpr bool T4 = QS && O <= 20 && sr_threads(D, RW) >= 256 * RW && ARITH == 0;
    using T16 = typename TileSel<H, D, O, KIND, ARITH>::type;
    using Tile = typename std::conditional<T4, Tile4<H, D, (O <= 20 ? O : 17), KIND>, T16>::type;
    constexpr int RWV = T4 ? 4 * RW : RW;            // rollout wavefronts
    constexpr int HD = H * D;
    constexpr int TPB = 16 * RW;                     // trajectories per workgroup
    constexpr int ROWS = TPB * D;                    // (trajectory, dim) rows: one thread (QS: one quad) each
    constexpr int NT = sr_threads(D, RW);            // sampling threads
    constexpr int NTT = NT + (PM ? 64 : 0);          // + the selection wavefront
    static_assert(NTT <= 1024 && HD % 2 == 0, "workgroup shape");
    constexpr int VW = HD % 4 == 0 ? 4 : 2;  // floats per vector of the tile -> HBM copy (rows are 4 * HD bytes)
    using Vec = typename VecOf<VW>::type;
    __shared__ __attribute__((aligned(16))) float ms[2 * HD];  // mean | std
    __shared__ __attribute__((aligned(16))) float tilebuf[T16::SLACK + TPB * HD + T16::TAIL];
    __shared__ unsigned long long wg_keys[2][RW][32];
    __shared__ float obs_stage[32];
    constexpr int WROWS = H / 2 + 1;                 // table rows the folded synthesis reads
    // QS: every sampling wave keeps its OWN copy of those rows (no workgroup barrier between filling and reading it)
    __shared__ __attribute__((aligned(16))) float Wl[QS ? NT / 64 : 1][QS ? WROWS * WQ_STRIDE : 4];
    __shared__ unsigned long long sel[PM ? 64 : 1];   // prologue selection (workgroup 0 of a riding pack: the pack's)
    __shared__ unsigned long long cand[PM ? 64 : 1];
    __shared__ int slot[PM ? 64 : 1];
    float* tile_rows = tilebuf + T16::SLACK;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_rows = ra.n_rows;       // sa.n sampled rows, then sa.n_shift shifted elites
    // riding pack (sharded runs, merge-prologue launches only): workgroup 0 packs + pushes the previous iteration's records
    const bool has_pack = PM && REC && ap.part_k != nullptr;
    if constexpr (PM && REC) {
        if (has_pack && blockIdx.x == 0) {
            MergeSingleArgs pk{};
            pk.n_lists = ap.n_lists;
            pk.n_pool = ap.n_pool;
            pk.n_global = ap.n_global;
            pk.K = ap.K;
            pk.h = H;
            pk.d = D;
            pk.part_k = ap.part_k;
            pk.actions = ap.actions;
            pk.n_keep = ap.n_keep;
            pk.elites_cost_cur = ap.keep_costs;
            pk.keep_base = ap.n_loc;
            // everybody else's prologue waits for this workgroup's push: its waves go first on their SIMDs
            __builtin_amdgcn_s_setprio(3);
            if (ra.dbg && tid == 0) ra.dbg[0] = wall_clock64();
            // the register-resident selection (174 registers, 1 us faster) where at most two waves share a SIMD anyway
            if (tid < 64) {
                if constexpr (NTT <= 512)
                    merge_select<12>(pk, lane, cand, sel);
                else
                    merge_select_stream(pk, lane, cand, sel);
            }
            __syncthreads();
            if (ra.dbg && tid == 0) ra.dbg[1] = wall_clock64();
            pack_records_body<12>(pk, ap.n_loc, ap.shard_lo, ap.records, ap.px, tilebuf, sel, tid, NTT);
            if (ra.dbg && tid == 0) ra.dbg[2] = wall_clock64();
            return;
        }
    }
    const int wg = blockIdx.x - (has_pack ? 1 : 0);
    // workgroups that emit a candidate list (behind them: shifted-elite rows scored through the cost array)
    const int n_wg = ra.list_wgs > 0 ? ra.list_wgs : (int)gridDim.x - (has_pack ? 1 : 0);
    const int base = wg * TPB;          // one slab of TPB trajectories per workgroup (launch_sample_rollout)
    if (base >= n_rows) return;
    if (ra.dbg && tid == 0 && wg == 0) ra.dbg[8] = wall_clock64();
    // Order matters at this size: kernel arguments arrive through serialized scalar loads, so everything the RNG
    // chain does not need (start observation, model operands, bounds) is fetched AFTER the sampling got going.
    const int rowi = QS ? tid >> 2 : tid, q = tid & 3;  // QS: lane q of the row's quad
    const int nl = rowi / D, jd = rowi - nl * D;
    const bool has_row = rowi < ROWS;
    float* trow = tile_rows + nl * HD + jd;
    const float* mrow = ms + jd;
    Tile tile;
    float obs_reg = 0.f;
    // QS: this wave's table rows, requested now, parked in LDS behind the draws (sample_into_tile_quad's publish)
    constexpr int WPL = (WROWS * HMAX + 63) / 64;
    float wreg[QS ? WPL : 1];
    // iteration 0 with its noise drawn ahead (beside the previous MPC step's last merge): nothing to sample here
    const bool pre_drawn = !PM && sa.raw_src != nullptr;
    if constexpr (QS) {
        if (tid < NT && !pre_drawn) {
#pragma unroll
            for (int i = 0; i < WPL; ++i) wreg[i] = sa.W[(i * 64 + lane) < WROWS * HMAX ? i * 64 + lane : 0];
        }
    }
    if (!PM) {  // iteration 0 of an MPC step: the distribution is in memory; the model operands ride the same wait
        obs_reg = ra.obs0[(tid < 32 && tid < ra.o) ? tid : 0];
        if (wave < RWV) tile.load(ra, lane);
        for (int e = tid; e < HD; e += NTT) {
            ms[e] = sa.mean[e];
            ms[HD + e] = sa.std[e];
        }
        __syncthreads();
    }
    if (ra.dbg && tid == 0 && wg == 0) ra.dbg[9] = wall_clock64();
    const int r_mine = base + nl;
    if (pre_drawn) {
        // the slab's raw rows (sampled rows, then the shifted elites' rows of stream off2) are one contiguous block
        {
            const int total4 = (n_rows - base < TPB ? n_rows - base : TPB) * (HD / VW);
            const Vec* g4 = reinterpret_cast<const Vec*>(sa.raw_src + (size_t)base * HD);
            Vec* t4 = reinterpret_cast<Vec*>(tile_rows);
            for (int e = tid; e < total4; e += NTT) t4[e] = g4[e];
        }
        __syncthreads();
        if (has_row && tid < NT) {
            const float lo = sa.low[jd], hi = sa.high[jd];
            constexpr int QN = QS ? 4 : 1;
            const int q0 = QS ? q : 0;
            if (r_mine < sa.n) {  // y * std + mean, clipped (icem.py:79): the samplers' fmaf + v_med3
                for (int t = q0; t < H; t += QN) {
                    const float v = __builtin_fmaf(trow[t * D], mrow[HD + t * D], mrow[t * D]);
                    trow[t * D] = __builtin_amdgcn_fmed3f(v, lo, hi);
                }
            } else if (r_mine < n_rows) {  // shifted elite: elites[e, 1:, j] ++ its freshly drawn last action (icem.py:91-104)
                const float* src = sa.elites_src + (size_t)(r_mine - sa.n) * HD + jd;
                if ((H - 1) % QN == q0) {
                    const float v = __builtin_fmaf(trow[(H - 1) * D], mrow[HD + (H - 1) * D], mrow[(H - 1) * D]);
                    trow[(H - 1) * D] = __builtin_amdgcn_fmed3f(v, lo, hi);
                }
                for (int t = q0; t < H - 1; t += QN) trow[t * D] = src[(t + 1) * D];
            } else {
                for (int t = q0; t < H; t += QN) trow[t * D] = 0.f;  // past the end: rolled out, dropped
            }
        }
    } else
    if constexpr (QS) {
        if (tid < NT) {
            float* wl = Wl[wave];
            sample_into_tile_quad<H, D, ROUNDS, PM>(sa, has_row, n_rows, r_mine, jd, q, trow, mrow, wl, [&]() {
#pragma unroll
                for (int i = 0; i < WPL; ++i) {
                    const int e = i * 64 + lane;
                    if (e < WROWS * HMAX) wl[(e / HMAX) * WQ_STRIDE + (e % HMAX)] = wreg[i];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            });
        }
    } else if (has_row) {
        sample_into_tile<H, D, ROUNDS, PM>(sa, n_rows, r_mine, jd, trow, mrow);
    }
    if constexpr (PM) {
        if (tid >= NT) {
            // the selection is the longer of the two concurrent chains on its SIMD: it goes first (c2 74.5 -> 72.7 us; with 8
            // rollout waves the sampling is the longer one and the priority costs 0.7 us)
            if constexpr (RW <= 4) __builtin_amdgcn_s_setprio(3);
            if constexpr (REC) {
                // (the records: staged by this wave, ranked by ALL threads behind the sampling -- merge_select_records_wg below)
                if (ra.dbg && lane == 0 && wg == 0) ra.dbg[3] = wall_clock64();
            }
            else if constexpr (RW >= 8)  // 13 waves share the register file: the low-register selection
                merge_select_stream(am, lane, cand, sel);
            else
                merge_select_shallow<3>(am, lane, cand, sel);
        }
    }
    if (PM) {  // now the rest of the inputs: in flight across the barriers below
        obs_reg = ra.obs0[(tid < 32 && tid < ra.o) ? tid : 0];
        if (wave < RWV) tile.load(ra, lane);
    }
    const float* rd0 = nullptr;
    if constexpr (!T4) rd0 = tile.read_ptr(tilebuf + (wave < RW ? wave : 0) * 16 * HD, lane, HD);
    if constexpr (PM) {
        const MergeSingleArgs& m = am;
        if constexpr (REC) {
            merge_select_records_wg(am, tid >= NT, lane, tid, NTT, sel, slot, (ra.dbg && wg == 0) ? ra.dbg + 7 : nullptr);
            if (ra.dbg && tid == 0 && wg == 0) ra.dbg[4] = wall_clock64();
        } else {
            __syncthreads();
        }
        if (ra.dbg && tid == 0 && wg == 0) ra.dbg[5] = wall_clock64();
        // all threads: gather the elite rows + refit (icem.py:201-211) -> this workgroup's mean / std
        const float* rows[KREG > 0 ? KREG : 1];
        merge_rows<KREG, REC>(m, sel, slot, rows);
        for (int e = tid; e < HD; e += NTT) {
            float xs[KREG > 0 ? KREG : 1];
#pragma unroll
            for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
            float nm, ns;
            refit_element_regs<float, KREG>(m.K, m.alpha, m.mean[e], m.std[e], xs, nm, ns);
            ms[e] = nm;
            ms[HD + e] = ns;
            if (wg == 0) {
                m.mean_out[e] = nm;
                m.std_out[e] = ns;
#pragma unroll
                for (int r = 0; r < KREG; ++r)
                    if (r < m.K) m.elites_next[(size_t)r * HD + e] = xs[r];
            }
        }
        if (wg == 0 && tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
        __syncthreads();
        if (ra.dbg && tid == 0 && wg == 0) ra.dbg[6] = wall_clock64();
        if (has_row && r_mine < sa.n) {  // y * std + mean, clipped (icem.py:79)
            const float lo = sa.low[jd], hi = sa.high[jd];
            for (int t = QS ? q : 0; t < H; t += QS ? 4 : 1) {
                const float v = __builtin_fmaf(trow[t * D], mrow[HD + t * D], mrow[t * D]);
                trow[t * D] = __builtin_amdgcn_fmed3f(v, lo, hi);
            }
        }
    }
    if (tid < 32) {  // start observation -> LDS (consumed after the barrier)
        float ov = obs_reg;
        asm volatile("" : "+v"(ov));  // wait for the load here, not where it was issued
        obs_stage[tid] = tid < ra.o ? ov : 0.f;
    }
    if (ra.dbg && tid == 0 && wg == 0) ra.dbg[10] = wall_clock64();
    __syncthreads();
    if (sa.row0_mean && sa.first_index + base == 0) {  // icem.py:87-88
        for (int e = tid; e < HD; e += NTT) tile_rows[e] = ms[e];
        __syncthreads();
    }
    {   // the tile is a contiguous block of the action tensor
        const int total4 = (n_rows - base < TPB ? n_rows - base : TPB) * (HD / VW);
        const Vec* t4 = reinterpret_cast<const Vec*>(tile_rows);
        Vec* g4 = reinterpret_cast<Vec*>(sa.out + (size_t)base * HD);
        // where the workgroup has at least two wavefronts besides the rollout waves, those copy and the rollout starts
        constexpr int CP0 = (NTT - 64 * RWV >= 128) ? 64 * RWV : 0;
        if (tid >= CP0)
            for (int e = tid - CP0; e < total4; e += NTT - CP0) g4[e] = t4[e];
    }
    if (ra.dbg && tid == 0 && wg == 0) ra.dbg[11] = wall_clock64();
    unsigned long long run_key = KEY_SENTINEL;
    if constexpr (T4) {
        // four trajectories per wave: row `wave * 4 + lane / 16` of the slab, lane % 16 = observation column
        if (wave < RWV) {
            tile.load_obs(obs_stage);
            const int rs = wave * 4 + (lane >> 4);  // trajectory of this row of lanes inside the slab
            const int row = base + rs;
            const float* rd = tile.read_ptr(tilebuf, rs, HD);
            const float* ar = tile.row_ptr(tilebuf, rs, HD);
            typename Tile::State st;
            tile.init(st);
#pragma unroll
            for (int t = 0; t < H; ++t) tile.step(st, rd + t * D, ar + t * D);
            const float cost = tile.cost(st);
            const bool live = row < n_rows;
            if ((lane & 15) == 0) {
                if (live) ra.costs[row] = cost;
                note_nonfinite(ra, cost, live);
                // TPB <= 32 keys; two-tile slabs (counted, below): a row that is no candidate leaves a filler -- (+inf, INT_MAX - 511
                // + rs), all different, behind every real key (wg_merge_emit's)
                wg_keys[0][0][rs] = (live && row < ra.n_cand) ? make_key(cost, row)
                                    : TPB <= 16 ? KEY_SENTINEL : ((KEY_SENTINEL & 0xFFFFFFFF00000000ull) | (KEY_FILL_LO + (unsigned)rs));
            }
        }
        if (ra.dbg && tid == 0 && wg == 0) ra.dbg[13] = wall_clock64();
        if (ra.K > 0 && wg < n_wg) {   // the slab's keys -> one sorted list (one wave)
            __syncthreads();
            if (wave == 0 && TPB <= 16) {   // 16 keys: the 10-stage network is shorter than counting + two cross-row adds (measured)
                const unsigned long long key = wave_sort_first<16>(lane < TPB ? wg_keys[0][0][lane] : KEY_SENTINEL, lane);
                if (lane < ra.K) {
                    if (ra.part_k) {
                        ra.part_k[(size_t)lane * n_wg + wg] = key;
                    } else {
                        ra.part_c[(size_t)wg * ra.K + lane] = key_cost(key);
                        ra.part_i[(size_t)wg * ra.K + lane] = key_idx(key);
                    }
                }
            } else if (wave == 0) {
                // 32 keys: the K best, ascending, by counting (no sort: the keys are all different, a key's place is the
                // number of smaller ones -- EXPERIMENTS R4.13): lane = key + NK x the part of the keys it reads
                constexpr int NK = TPB <= 16 ? 16 : 32, PARTS = 64 / NK, PER = NK / PARTS;
                static_assert(TPB <= 32 && PER % 2 == 0, "one wave counts for the slab");
                const unsigned long long* keys = &wg_keys[0][0][0];
                const int i = lane & (NK - 1), part = lane / NK;
                const unsigned long long mine = i < TPB ? keys[i] : KEY_SENTINEL;
                unsigned place = 0;
#pragma unroll
                for (int c = 0; c < PER; ++c) {
                    const int e = part * PER + c;
                    place += (e < TPB && keys[e < TPB ? e : 0] < mine) ? 1u : 0u;
                }
                if (NK == 16) place += (unsigned)__shfl_xor((int)place, 16, 64);
                place += (unsigned)__shfl_xor((int)place, 32, 64);
                if (part == 0 && i < TPB && place < (unsigned)ra.K) {
                    const bool filler = (mine >> 32) == (KEY_SENTINEL >> 32) && (unsigned)mine >= KEY_FILL_LO;
                    const unsigned long long key = filler ? KEY_SENTINEL : mine;
                    if (ra.part_k) {
                        ra.part_k[(size_t)place * n_wg + wg] = key;
                    } else {
                        ra.part_c[(size_t)wg * ra.K + place] = key_cost(key);
                        ra.part_i[(size_t)wg * ra.K + place] = key_idx(key);
                    }
                }
            }
        }
    } else {
        if (wave < RW) {
            tile.load_obs(obs_stage);
            run_key = rollout_slab<Tile, H, D, true>(tile, ra, rd0, base + wave * 16 + (lane & 15), n_rows, run_key, true, lane);
            if (ra.dbg && tid == 0 && wg == 0) ra.dbg[12] = wall_clock64();
        }
        if (ra.dbg && tid == 0 && wg == 0) ra.dbg[13] = wall_clock64();
        if (ra.K > 0 && wg < n_wg) wg_merge_emit<RW>(wg_keys, run_key, ra.K, lane, wave, ra, wg, n_wg);
    }
    if (ra.dbg && tid == 0 && wg == 0) ra.dbg[14] = wall_clock64();
}


template <int H, int D, int O, int KIND, int ROUNDS, int RW, int KREG, bool REC, int ARITH>
__global__ __launch_bounds__(sr_threads(D, RW) + (KREG > 0 ? 64 : 0)) void sample_rollout_kernel(FastIterArgs a) {
    sample_rollout_body<H, D, O, KIND, ROUNDS, RW, KREG, REC, ARITH>(a.s, a.r, a.m, a.p);
}

// B problems in one launch: blockIdx.y = the problem, its argument block in device memory (scalar loads: the index is
// uniform), the sampling calls' stream offsets stored relative to the step's base of that problem (BatchBases)
template <int H, int D, int O, int KIND, int ROUNDS, int RW, int KREG, int ARITH>
__global__ __launch_bounds__(sr_threads(D, RW) + (KREG > 0 ? 64 : 0)) void sample_rollout_batch_kernel(const FastIterArgs* __restrict__ args,
                                                                                                      BatchBases bases) {
    const FastIterArgs& a = args[blockIdx.y];
    FastSampleArgs s = from_device(a.s);
    const unsigned long long base = bases.v[blockIdx.y];
    add_base64(s.off_lo, s.off_hi, base);
    add_base64(s.off2_lo, s.off2_hi, base);
    const FastRolloutArgs r = from_device(a.r);
    const MergeSingleArgs m = from_device(a.m);
    sample_rollout_body<H, D, O, KIND, ROUNDS, RW, KREG, false, ARITH>(s, r, m, a.p);
}

}  // namespace

// rollout waves a single-launch workgroup can hold for this shape: 16 * RW * D sampling threads (+ 64) within 1024
// threads, the [16 * RW, H, D] tile within ~120 KB of LDS
constexpr int single_launch_max_rw(int h, int d) {
    int best = 0;
    for (int rw = 1; rw <= 8; rw *= 2)
        if (sr_threads(d, rw) + 64 <= 1024 && 16 * rw * h * d * 4 <= 120 * 1024) best = rw;
    return best;
}

// single-launch iteration: compiled for the default generator (10 Philox rounds) and 1, 2, 4 or 8 rollout waves per
// workgroup, one slab of 16 * rw trajectories each, at most FAST_MAX_LISTS workgroups (= candidate lists).
// sample_rollout_lists: workgroups of the launch, 0 when the shape or size is outside that (use the two-kernel
// path).  (Several slabs per workgroup through the same LDS tile were tried for larger populations: with one
// 92 KB tile per CU the sampling and rollout phases of a workgroup run back to back at 2-3 waves per SIMD, and
// N=65 536 took 297 instead of 220 us per MPC step -- the two full-occupancy kernels win there.)
// n_tail: the last n_tail rows are shifted elites.  Where the sampled rows alone fill exactly FAST_MAX_LISTS slabs
// (N = 4 096 at one tile per workgroup, 8 192 at two, ...) the shifted elites get workgroups of their own BEHIND the
// list-writing ones: those roll their rows out and store the costs but emit no list -- the merge takes the rows as
// extra candidates straight from the cost array (its kept-elite slot, free at iteration 0; *tail_out rows from pool row
// n on).  Otherwise three shifted rows would push the launch over the list limit and onto twice the trajectories per
// workgroup on half the CUs (N = 4 096: 15.5 instead of 12.5 us).
static bool sample_rollout_shape(int h, int d, int O, int rounds, int n_rows, int n_tail, int* grid_out, int* rw_out,
                                 int* tail_out = nullptr) {
    const int max_rw = opt_i(OPT_FUSE_MAX_RW);  // read per call: the path-equivalence test flips it between planners
    int grid, rw, tail = 0;
    r16_shape(n_rows, &grid, &rw);
    const int mult = g_batch.mult;
    if (mult > 1) {
        // a batch of `mult` problems of this size in one launch (icem_plan_step_batch): the slab size the chip would get for
        // all their rows together, the workgroup count of ONE problem (blockIdx.y is the problem); no tail shape
        int g_all, rw_all;
        r16_shape(n_rows * mult, &g_all, &rw_all);
        const int cap = opt_i(OPT_BATCH_MAX_RW);
        rw = std::min(std::max(rw, rw_all), std::min(cap > 0 ? cap : 8, single_launch_max_rw(h, d)));
        while (rw > 1 && rw > single_launch_max_rw(h, d)) rw /= 2;
        grid = (std::max(1, (n_rows + 15) / 16) + rw - 1) / rw;
        if (grid > FAST_MAX_LISTS) return false;
    } else
    if (n_tail > 0 && n_tail <= 64) {
        int g2, w2;
        r16_shape(n_rows - n_tail, &g2, &w2);
        // (only shapes whose workgroups fit a CU twice -- at most 512 threads: 1 or 4 tiles per workgroup at d = 6 -- so that
        // the extra workgroups run beside the others; a 2-tile workgroup (832 threads with quad sampling) or an 8-tile one
        // owns its CU, workgroup 257 would be a second round: N = 8 192 measured 23.5 us against 18.7 for the 4-tile shape)
        if (w2 < rw && sr_threads(d, w2) + 64 <= 512 && g2 == FAST_MAX_LISTS && n_rows - n_tail == FAST_MAX_LISTS * 16 * w2) {
            rw = w2;
            tail = n_tail;
        }
    }
    if (rounds != 10 || rw > max_rw || n_rows <= 0 || !fast_rollout_supported(h, d, O, 1) || !fast_sample_supported(h, d))
        return false;
    if (rw > single_launch_max_rw(h, d)) return false;  // one slab of 16 * rw trajectories per workgroup
    *grid_out = tail ? (n_rows + 16 * rw - 1) / (16 * rw) : std::min(grid, (n_rows + 16 * rw - 1) / (16 * rw));
    *rw_out = rw;
    if (tail_out) *tail_out = tail;
    return true;
}

int sample_rollout_lists(int h, int d, int O, int rounds, int n_rows, int n_tail, int* tail_out) {
    int grid, rw, tail = 0;
    if (!sample_rollout_shape(h, d, O, rounds, n_rows, n_tail, &grid, &rw, &tail)) return 0;
    if (tail_out) *tail_out = tail;
    return tail ? FAST_MAX_LISTS : grid;
}

// merge prologue: the selection wavefront joins the sampling waves (8 rollout waves: 13 waves share the register
// file, the selection runs in its low-register form)
bool sample_rollout_merge_ok(int h, int d, int O, int rounds, int n_rows, int K) {
    const int on = opt_i(OPT_MERGE_PROLOGUE);
    int grid, rw;
    return on && K + 1 <= 12 && sample_rollout_shape(h, d, O, rounds, n_rows, 0, &grid, &rw);
}

bool sample_rollout_pack_ok(int h, int d, int O, int rounds, int n_rows, int K) {
    const int on = opt_i(OPT_RIDING_PACK);
    int grid, rw;
    return on && sample_rollout_merge_ok(h, d, O, rounds, n_rows, K) && sample_rollout_shape(h, d, O, rounds, n_rows, 0, &grid, &rw) &&
           K * (h * d + 2) <= 16 * rw * h * d;
}

void launch_sample_rollout(const FastIterArgs& a, int h, int d, int O, int kind, bool merge_prologue, hipStream_t st) {
    int grid, rw;
    // (the tail shape only where the caller chose it -- it sets list_wgs then: the list count is the caller's contract)
    if (!sample_rollout_shape(h, d, O, 10, a.r.n_rows, a.r.list_wgs > 0 ? a.s.n_shift : 0, &grid, &rw)) return;
    if (g_batch.rec) {   // icem_plan_step_batch: recorded, launched for all problems at once (launch_sample_rollout_batch)
        if (merge_prologue && a.m.records) {
            g_batch.unsupported = true;
            return;
        }
        BatchRecord r;
        r.kind = 1;
        r.it = a;
        r.h = h, r.d = d, r.O = O, r.model_kind = kind, r.rw = rw, r.grid = grid;
        r.prologue = merge_prologue;
        g_batch.rec->push_back(r);
        return;
    }
    if (merge_prologue && a.m.records && a.p.part_k) grid += 1;  // workgroup 0: the riding pack
#define XK(HH, DD, OO, WW, KR, RC)                                                                                      \
    {                                                                                                                   \
        constexpr int NT = sr_threads(DD, WW) + (KR > 0 ? 64 : 0);                                                     \
        if constexpr (OO <= 20) {                                                                                       \
            if (a.r.arith == 1) {                                                                                       \
                if (kind == 1)                                                                                          \
                    hipLaunchKernelGGL((sample_rollout_kernel<HH, DD, OO, 1, 10, WW, KR, RC, 1>), dim3(grid), dim3(NT), 0, st, a); \
                else                                                                                                    \
                    hipLaunchKernelGGL((sample_rollout_kernel<HH, DD, OO, 0, 10, WW, KR, RC, 1>), dim3(grid), dim3(NT), 0, st, a); \
                return;                                                                                                 \
            }                                                                                                           \
        }                                                                                                               \
        if (kind == 1)                                                                                                  \
            hipLaunchKernelGGL((sample_rollout_kernel<HH, DD, OO, 1, 10, WW, KR, RC, 0>), dim3(grid), dim3(NT), 0, st, a); \
        else                                                                                                            \
            hipLaunchKernelGGL((sample_rollout_kernel<HH, DD, OO, 0, 10, WW, KR, RC, 0>), dim3(grid), dim3(NT), 0, st, a); \
        return;                                                                                                         \
    }
#define XW(HH, DD, OO, WW)                                                   \
    if constexpr (WW <= single_launch_max_rw(HH, DD)) {                      \
        if (rw == WW) {                                                      \
            if (merge_prologue && a.m.records) XK(HH, DD, OO, WW, 12, true)  \
            if (merge_prologue) XK(HH, DD, OO, WW, 12, false)                \
            XK(HH, DD, OO, WW, 0, false)                                     \
        }                                                                    \
    }
#define XR(HH, DD, OO)                   \
    if (h == HH && d == DD && O == OO) { \
        XW(HH, DD, OO, 1)                \
        XW(HH, DD, OO, 2)                \
        XW(HH, DD, OO, 4)                \
        XW(HH, DD, OO, 8)                \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XW
#undef XK
}

// ... and the same launch for n problems (blockIdx.y): shape = one problem's record (all equal: plan.hip checked)
void launch_sample_rollout_batch(const BatchRecord& s, const FastIterArgs* args_dev, const BatchBases& bases, int n, hipStream_t st) {
    const int h = s.h, d = s.d, O = s.O, kind = s.model_kind, rw = s.rw, arith = s.it.r.arith;
    const dim3 grid(s.grid, n);
#define XK(HH, DD, OO, WW, KR)                                                                                          \
    {                                                                                                                   \
        constexpr int NT = sr_threads(DD, WW) + (KR > 0 ? 64 : 0);                                                     \
        if constexpr (OO <= 20) {                                                                                       \
            if (arith == 1) {                                                                                           \
                if (kind == 1)                                                                                          \
                    hipLaunchKernelGGL((sample_rollout_batch_kernel<HH, DD, OO, 1, 10, WW, KR, 1>), grid, dim3(NT), 0, st, args_dev, bases); \
                else                                                                                                    \
                    hipLaunchKernelGGL((sample_rollout_batch_kernel<HH, DD, OO, 0, 10, WW, KR, 1>), grid, dim3(NT), 0, st, args_dev, bases); \
                return;                                                                                                 \
            }                                                                                                           \
        }                                                                                                               \
        if (kind == 1)                                                                                                  \
            hipLaunchKernelGGL((sample_rollout_batch_kernel<HH, DD, OO, 1, 10, WW, KR, 0>), grid, dim3(NT), 0, st, args_dev, bases); \
        else                                                                                                            \
            hipLaunchKernelGGL((sample_rollout_batch_kernel<HH, DD, OO, 0, 10, WW, KR, 0>), grid, dim3(NT), 0, st, args_dev, bases); \
        return;                                                                                                         \
    }
#define XW(HH, DD, OO, WW)                                  \
    if constexpr (WW <= single_launch_max_rw(HH, DD)) {     \
        if (rw == WW) {                                     \
            if (s.prologue) XK(HH, DD, OO, WW, 12)          \
            XK(HH, DD, OO, WW, 0)                           \
        }                                                   \
    }
#define XR(HH, DD, OO)                   \
    if (h == HH && d == DD && O == OO) { \
        XW(HH, DD, OO, 1)                \
        XW(HH, DD, OO, 2)                \
        XW(HH, DD, OO, 4)                \
        XW(HH, DD, OO, 8)                \
    }
    ICEM_FAST_SHAPES(XR)
#undef XR
#undef XW
#undef XK
}

}  // namespace icem
