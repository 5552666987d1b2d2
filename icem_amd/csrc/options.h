// options.h -- the library's development options (icem_set_option / icem_get_option; table and storage in abi.hip).
// Internal; not part of the public ABI.
#pragma once

namespace icem {

// ---- development options (icem_set_option / icem_get_option; abi.hip) ---------------------------------------------------
// Which of several BIT-IDENTICAL launch arrangements serves a call (the equivalence tests flip them), tuning fractions of
// the noise-ahead launches and the bound of the exchange's device-side waits.  One process-wide table set through the C
// ABI; the library itself never reads the environment (tools / bench.py map ICEM_<NAME> variables onto it explicitly:
// icem_amd._lib.apply_env_options).  Nothing here selects an ARITHMETIC: that is per handle (icem_set_tile_arith /
// icem_set_wide_arith), so that every rank of a sharded run computes the same bits whatever its process environment.
#define ICEM_OPTIONS(X)                                                                                                   \
    X(DISABLE_FAST, "disable_fast", 0.0)             /* 1: f32 plans on the generic one-kernel-per-stage path (read at icem_create) */ \
    X(FUSE_MAX_RW, "fuse_max_rw", 8.0)               /* rollout waves per workgroup of the single-launch kernel; 0: sampler + rollout pair */ \
    X(MERGE_PROLOGUE, "merge_prologue", 1.0)         /* 0: every merge a launch of its own */                             \
    X(RIDING_PACK, "riding_pack", 1.0)               /* 0: every record pack a launch of its own (sharded) */              \
    X(PUBLISHED_MERGE, "published_merge", 1.0)       /* 0: every workgroup merges the records for itself (sharded) */      \
    X(PACK_MERGE, "pack_merge", 1.0)                 /* 0: a sharded step's last pack and records merge as two launches */ \
    X(NOISE_AHEAD, "noise_ahead", 1.0)               /* 0: large populations on the sampler + rollout pair */              \
    X(NOISE_AHEAD_MIN_ROWS, "noise_ahead_min_rows", 0.0)                                                                \
    X(NOISE_AHEAD_SHARDED, "noise_ahead_sharded", 1.0)                                                                     \
    X(AHEAD_STAMPS, "ahead_stamps", 0.0)                                                                                   \
    X(AHEAD_TAIL_FRAC, "ahead_tail_frac", 0.4)                                                                             \
    X(AHEAD_NEXT1_FRAC, "ahead_next1_frac", 0.3)                                                                           \
    X(AHEAD_NOISE_LDS_KB, "ahead_noise_lds_kb", -1.0)                                                                      \
    X(PREDRAW, "predraw", 1.0)                       /* 0: iteration 0 of small populations samples its own noise */       \
    X(GK_SAMPLE, "gk_sample", 1.0)                   /* strict-parity path: 0 = one thread per row in the sampler */       \
    X(GK_ROLLOUT_THREAD, "gk_rollout_thread", 0.0)   /* ... 1 = one thread per trajectory in the rollout */                \
    X(GK_SELECT, "gk_select", 1.0)                   /* ... 0 = topk_partial + local_pack + merge_refit */                 \
    X(HN_PAIR, "hn_pair", 1.0)                       /* TileHN wave arrangements */                                        \
    X(HN_SPLIT, "hn_split", 1.0)                                                                                           \
    X(RSSM_SPLIT, "rssm_split", 1.0)                                                                                       \
    X(RSSM_SPLIT_MAX_N, "rssm_split_max_n", 65536.0)                                                                       \
    X(RSSM_SPLIT_TT, "rssm_split_tt", 0.0)                                                                                 \
    X(STEP_XCD, "step_xcd", 0.0)                     /* 1: populations <= 4096 rows take ONE launch per MPC step inside one XCD (k_step_xcd.hip: bit-equal, measured SLOWER -- 87 vs 61 us; EXPERIMENTS R6.4) */ \
    X(BATCH_AHEAD, "batch_ahead", 1.0)               /* icem_plan_step_batch: 0 = always the single-launch kernels (never the noise-ahead launches) */ \
    X(BATCH_AHEAD_MIN_ROWS, "batch_ahead_min_rows", 49152.0)   /* ... from this many rows of all problems' first iterations together */ \
    X(BATCH_MAX_RW, "batch_max_rw", 0.0)             /* icem_plan_step_batch: cap of the tiles per workgroup (0: as one population of all rows) */ \
    X(XCHG_LOOPBACK, "xchg_loopback", 0.0)           /* 1: time one rank without its peers (tools/sharded_rank_bench.py) */ \
    X(XCHG_MAX_POLLS, "xchg_max_polls", 0.0)         /* bound of the exchange's device-side waits (0: the default) */
enum Opt {
#define X(id, name, def) OPT_##id,
    ICEM_OPTIONS(X)
#undef X
    OPT_COUNT
};
double opt(Opt k);                 // the option's current value (abi.hip)
inline int opt_i(Opt k) { return (int)opt(k); }

}  // namespace icem
