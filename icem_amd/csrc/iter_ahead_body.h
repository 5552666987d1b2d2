// iter_ahead_body.h -- the BODY of iter_ahead_kernel (k_rollout_ahead.hip), included textually into the kernel that takes its
// argument block by value and into iter_ahead_batch_kernel, which builds `args` from an array in device memory
// (icem_plan_step_batch): one text, and the by-value kernel compiles exactly as it did before the batched form existed (a
// shared device function changed its register allocation: 16 -> 18 spilled VGPRs on the tanh instantiations).
// Expects: template parameters H, D, O, KIND, WAVES, PM, ARITH and `args` (IterAheadArgs) in scope.  No include guard.
    using Stream = Stream16<H, D, O, KIND, ARITH>;
    using Tile = typename Stream::Tile;
    static_assert(Tile::SLACK == Tile16<H, D, O, KIND>::SLACK && Tile::TAIL == Tile16<H, D, O, KIND>::TAIL, "one staging layout");
    using L = AheadLds<H, D, O, KIND, WAVES>;
    constexpr int HD = H * D, NTT = 64 * WAVES, KREG = AHEAD_KREG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int n_roll = args.n_roll;
    // ------------------------------------------------------------------------------------------------ pack role
    // (sharded runs) workgroup 0: the previous iteration's K best of this rank's lists -> records, pushed into every rank's
    // exchange block; then THE records merge of the launch for everybody (waits for all ranks' flags, selects, gathers,
    // refits) and its publication (written through; one agent-scope flag).  sample_folded_merge_kernel's workgroup 0,
    // with the local selection shared by all waves.
    if constexpr (PM == 2) {
        if (blockIdx.x == 0) {
            unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem + L::SEL);
            unsigned long long* wsel = reinterpret_cast<unsigned long long*>(smem + L::WSEL);
            int* slot = reinterpret_cast<int*>(smem + L::SLOT);
            unsigned long long* cand = reinterpret_cast<unsigned long long*>(smem + L::STAGE) + wave * 64;
            float* stage = smem + L::STAGE + WAVES * 128;  // behind the compaction scratch: [K, rs] floats of records
            MergeSingleArgs pk{};
            pk.n_lists = args.p.n_lists;
            pk.n_pool = args.p.n_pool;
            pk.n_global = args.p.n_global;
            pk.K = args.p.K;
            pk.h = H;
            pk.d = D;
            pk.part_k = args.p.part_k;
            pk.actions = args.p.actions;
            pk.n_keep = args.p.n_keep;
            pk.elites_cost_cur = args.p.keep_costs;
            pk.keep_base = args.p.n_loc;
            __builtin_amdgcn_s_setprio(3);  // everybody else waits for this workgroup
            const bool by_rank = merge_select_split_by_rank<WAVES>(pk);   // (uniform) the K best by counting, all waves
            const float pk_keep = by_rank ? merge_keep_cost_split<WAVES>(pk, lane, wave) : merge_keep_cost(pk, lane);
            merge_select_split_stage1<KREG>(pk, lane, wave, WAVES, cand, wsel);
            if (by_rank) merge_select_split_keep<WAVES>(pk, lane, wave, wsel, sel, pk_keep);
            __syncthreads();
            if (by_rank) merge_select_split_rank<WAVES>(pk, lane, wave, wsel, sel);
            else if (wave == 0) merge_select_split_stage2(pk, lane, WAVES, wsel, cand, sel, pk_keep);
            __syncthreads();
            pack_records_body<KREG>(pk, args.p.n_loc, args.p.shard_lo, args.p.records, args.p.px, stage, sel, tid, NTT);
            __syncthreads();
            const MergeSingleArgs& m = args.m;
            merge_select_records_wg(m, wave == 0, lane, tid, NTT, sel, slot);
            const float* rows[KREG];
            merge_rows<KREG, true>(m, sel, slot, rows);
            for (int e = tid; e < HD; e += NTT) {
                float xs[KREG];
#pragma unroll
                for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
                float nm, ns;
                refit_element_regs<float, KREG>(m.K, m.alpha, m.mean[e], m.std[e], xs, nm, ns);
                __hip_atomic_store(args.p.pub + e, nm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(args.p.pub + HD + e, ns, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                m.mean_out[e] = nm;
                m.std_out[e] = ns;
#pragma unroll
                for (int r = 0; r < KREG; ++r)
                    if (r < m.K) m.elites_next[(size_t)r * HD + e] = xs[r];
            }
            if (tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(args.p.pub_flag, args.p.pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    // workgroup order: [pack (PM == 2)] [shift (iteration 0)] [rollout x n_roll] [noise x n_noise].  The shift workgroup is
    // the launch's longest chain of ONE wave (three rows sampled by single lanes, then 30 steps of a lone wave: ~12 us) and
    // used to be dispatched LAST, behind every noise workgroup -- it started 22 us into the launch and ended 10 us after the
    // last rollout workgroup (stamps: EXPERIMENTS R5.2).  First in line it ends long before them.
    const bool has_shift = args.s.n_shift > 0;
    const int raw = (int)blockIdx.x - (PM == 2 ? 1 : 0);
    const bool is_shift = has_shift && raw == 0;
    const int bid = is_shift ? n_roll + args.n_noise : raw - (has_shift ? 1 : 0);  // workgroup number within [rollout | noise | shift]
    // ------------------------------------------------------------------------------------------------ noise role
    if (bid >= n_roll && bid < n_roll + args.n_noise) {
        const FastSampleArgs& z = args.z;
        long long* zs = (args.r.dbg && tid == 0 && (bid == n_roll || bid == n_roll + args.n_noise - 1))
                            ? args.r.dbg + 16 + 32 * args.dbg_slot + (bid == n_roll ? 20 : 24) : nullptr;
        if (zs) zs[0] = wall_clock64();
        float* tile = smem;  // [TPW, HD]
        const int n_base = (bid - n_roll) * L::TPW;
        const int n_here = cmin(L::TPW, z.n - n_base);
        if (tid < n_here * D) {
            const int nl = tid / D;
            const int j = tid - nl * D;
            float* trow = tile + nl * HD + j;
            sample_row<H, 10>(z.W, (unsigned)(z.first_index + n_base + nl), (unsigned)j, z.off_lo, z.off_hi, z.seed_lo, z.seed_hi,
                              [&](int t, float y) { trow[t * D] = y; }, z.white != 0);
        }
        if (zs) zs[2] = wall_clock64();
        __syncthreads();
        if (zs) zs[3] = wall_clock64();
        float* gdst = z.out + (size_t)n_base * HD;
        const int total = n_here * HD;
        if constexpr ((HD & 3) == 0) {
            const float4* t4 = reinterpret_cast<const float4*>(tile);
            float4* g4 = reinterpret_cast<float4*>(gdst);
            for (int e = tid; e < total / 4; e += NTT) g4[e] = t4[e];
        } else {
            const float2* t2 = reinterpret_cast<const float2*>(tile);
            float2* g2 = reinterpret_cast<float2*>(gdst);
            for (int e = tid; e < total / 2; e += NTT) g2[e] = t2[e];
        }
        if (zs) zs[1] = wall_clock64();
        return;
    }
    const FastRolloutArgs& a = args.r;
    // ------------------------------------------------------------------------------------------------ shift role
    if (bid >= n_roll) {
        // shifted elite e: elites[e, 1:, j] and a last action drawn from the full (n_shift, d, h) noise batch of stream off2
        // (only t = h-1 is used, icem.py:102) -> pool rows [n, n + n_shift) and a 16-row LDS tile; then one wave rolls the
        // tile out (Tile16: the bits the rollout role would produce for these rows) -> costs [n, n + n_shift)
        const FastSampleArgs& s = args.s;
        long long* ss = (a.dbg && tid == 0) ? a.dbg + 16 + 32 * args.dbg_slot + 16 : nullptr;
        if (ss) ss[0] = wall_clock64();
        float* ms = smem + L::SH_DIST;
        float* tilebuf = smem + L::SH_TILE;
        float* obs_stage = tilebuf + Tile::SLACK + 16 * HD + Tile::TAIL;
        float* rows = tilebuf + Tile::SLACK;
        const float obs_reg = a.obs0[(tid < 32 && tid < a.o) ? tid : 0];
        Tile tile;
        if (wave == 0) tile.load(a, lane);
        for (int e = tid; e < HD; e += NTT) {
            ms[e] = s.mean[e];
            ms[HD + e] = s.std[e];
        }
        for (int e = tid; e < 16 * HD; e += NTT) rows[e] = 0.f;
        if (tid < 32) obs_stage[tid] = tid < a.o ? obs_reg : 0.f;
        __syncthreads();
        if (tid < s.n_shift * D) {
            const int e = tid / D;
            const int j = tid - e * D;
            const float lo = s.low[j], hi = s.high[j];
            float last = 0.f;
            sample_row<H, 10>(s.W, (unsigned)e, (unsigned)j, s.off2_lo, s.off2_hi, s.seed_lo, s.seed_hi,
                              [&](int t, float y) {
                                  if (t == H - 1) {
                                      float v = __builtin_fmaf(y, ms[HD + t * D + j], ms[t * D + j]);
                                      v = v < lo ? lo : v;
                                      last = v > hi ? hi : v;
                                  }
                              }, s.white != 0);
            float* dst = s.out + (size_t)(s.n + e) * HD + j;
            const float* src = s.elites_src + (size_t)e * HD + j;
            float* trow = rows + e * HD + j;
            for (int t = 0; t < H - 1; ++t) {
                const float v = src[(t + 1) * D];
                dst[t * D] = v;
                trow[t * D] = v;
            }
            dst[(H - 1) * D] = last;
            trow[(H - 1) * D] = last;
        }
        __syncthreads();
        if (wave == 0) {
            tile.load_obs(obs_stage);
            FastRolloutArgs ta = a;
            ta.K = 0;
            ta.costs = a.costs + s.n;
            (void)rollout_slab<Tile, H, D>(tile, ta, tile.read_ptr(tilebuf, lane, HD), lane & 15, s.n_shift, KEY_SENTINEL, true, lane);
        }
        if (ss) ss[1] = wall_clock64();
        return;
    }
    // ------------------------------------------------------------------------------------------------ rollout role
    // (s_setprio for this role's waves -- the longer chain -- measured: no effect, 185.0 vs 184.7 us per MPC step)
    float* stage = smem + L::STAGE;
    auto wg_keys = reinterpret_cast<unsigned long long(*)[WAVES][32]>(smem + L::KEYS);
    float* dist = smem + L::DIST;  // mean | std this iteration samples from
    float* obs_stage = smem + L::OBS;
    unsigned long long* sel = reinterpret_cast<unsigned long long*>(smem + L::SEL);
    unsigned long long* wsel = reinterpret_cast<unsigned long long*>(smem + L::WSEL);
    int* slot = reinterpret_cast<int*>(smem + L::SLOT);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(stage) + wave * 64;  // this wave's compaction scratch
    // development (icem_debug_stamps + ICEM_AHEAD_STAMPS=1; tools/dbg/ahead_stamps.py): wall_clock64 phase stamps of thread 0
    // of the first and the last rollout workgroup, [16 + 32 x iteration]: [2][8], then entry / exit of the shift workgroup and of the last and the first noise workgroup; a.dbg == nullptr in production
    // (slots 16 + 32 x iteration: the merge kernels stamp the first eight words)
    long long* stamps = (a.dbg && tid == 0 && (bid == 0 || bid == n_roll - 1)) ? a.dbg + 16 + 32 * args.dbg_slot + (bid == 0 ? 0 : 8) : nullptr;
    if (stamps) stamps[0] = wall_clock64();
    // model operands and start observation in flight in front of the merge
    const float obs_reg = a.obs0[(tid < 32 && tid < a.o) ? tid : 0];
    Tile tile;
    tile.load(a, lane);
    Stream stream;
    stream.init(tile, stage + wave * Stream::STG, lane);
    const int tiles = (a.n_rows + 15) / 16;
    const int tile0 = wave * n_roll + bid;
    typename Stream::Vec pre[Stream::NLD];
    if constexpr (PM == 1) {
        // all waves share the selection: one cold round trip instead of a dozen dependent ones (the kept elites' costs,
        // stage 2's other input, travel with it)
        // ONE wave selects, the lists' first three depths in registers (merge_select_shallow): 4.7 us from the launch's entry where
        // the selection shared by all eight waves -- stage 1 per wave, a barrier, stage 2 -- took 5.5-6.2 (c4 146.5 -> 140.8 us
        // per MPC step, EXPERIMENTS R6.17); the others request their first noise vectors and wait at the barrier below.
        // (the streaming single-wave form, which re-reads the lists depth by depth: 7.7 us, R6.16)
        if (wave == 0) merge_select_shallow<3>(args.m, lane, cand, sel);
        if (tile0 < tiles) stream.first_loads(args.pool, a.n_rows, tile0, pre);
        if (stamps) stamps[1] = stamps[2] = wall_clock64();
    } else if constexpr (PM == 2) {
        // sharded: the pack role merges for everybody -- wait for its flag (bounded like every exchange wait)
        if (tile0 < tiles) stream.first_loads(args.pool, a.n_rows, tile0, pre);
        if (wave == 0) {
            unsigned polls = 0;
            while (__hip_atomic_load(args.p.pub_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != args.p.pub_seq &&
                   ++polls <= args.m.xw.max_polls)
                __builtin_amdgcn_s_sleep(16);
            if (polls > args.m.xw.max_polls && lane == 0 && args.m.xw.status)
                __hip_atomic_store(args.m.xw.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else {
        if (tile0 < tiles) stream.first_loads(args.pool, a.n_rows, tile0, pre);
        for (int e = tid; e < HD; e += NTT) {
            dist[e] = args.mean[e];
            dist[HD + e] = args.std[e];
        }
    }
    if (tid < 32) obs_stage[tid] = tid < a.o ? obs_reg : 0.f;
    __syncthreads();
    if (stamps) stamps[3] = wall_clock64();
    if constexpr (PM == 2) {
        for (int e = tid; e < 2 * HD; e += NTT) dist[e] = __hip_atomic_load(args.p.pub + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    if constexpr (PM == 1) {
        const MergeSingleArgs& m = args.m;
        const float* rows[KREG];
        merge_rows<KREG, false>(m, sel, slot, rows);
        for (int e = tid; e < HD; e += NTT) {
            float xs[KREG];
#pragma unroll
            for (int r = 0; r < KREG; ++r) xs[r] = rows[r][e];
            const float om = m.mean[e], os = m.std[e];
            // elite rows the previous launch left as raw noise: the action it rolled out (same fmaf + v_med3, same bits)
#pragma unroll
            for (int r = 0; r < KREG; ++r) {
                const bool raw = key_idx(sel[r < m.K ? r : 0]) < m.n_raw;
                const float v = __builtin_amdgcn_fmed3f(__builtin_fmaf(xs[r], os, om), m.xf_lo, m.xf_hi);
                xs[r] = raw ? v : xs[r];
            }
            float nm, ns;
            refit_element_regs<float, KREG>(m.K, m.alpha, om, os, xs, nm, ns);
            dist[e] = nm;
            dist[HD + e] = ns;
            if (bid == 0) {
                m.mean_out[e] = nm;
                m.std_out[e] = ns;
#pragma unroll
                for (int r = 0; r < KREG; ++r)
                    if (r < m.K) m.elites_next[(size_t)r * HD + e] = xs[r];
            }
        }
        if (bid == 0 && tid < m.K) m.elites_cost_next[tid] = key_cost(sel[tid]);
        __syncthreads();
    }
    if (stamps) stamps[4] = wall_clock64();
    tile.load_obs(obs_stage);
    unsigned long long run_key = KEY_SENTINEL;
    bool first = true;
    // tile t of the launch belongs to wave t / n_roll of rollout workgroup t % n_roll (as rollout16_kernel)
    for (int tile_id = tile0; tile_id < tiles; tile_id += WAVES * n_roll) {
        if (!first) stream.first_loads(args.pool, a.n_rows, tile_id, pre);
        run_key = stream.run_xf(tile, a, args.pool, args.n_xf, args.row0_mean != 0, args.store_back != 0, dist, args.lo, args.hi, tile_id, lane, run_key, first, pre);
        if (stamps && first) stamps[5] = wall_clock64();
        first = false;
    }
    if (stamps) stamps[6] = wall_clock64();
    if (a.K > 0) wg_merge_emit<WAVES>(wg_keys, run_key, a.K, lane, wave, a, bid, n_roll);
    if (stamps) stamps[7] = wall_clock64();
