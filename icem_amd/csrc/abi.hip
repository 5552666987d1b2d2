// abi.hip -- handle life cycle and the stateless operators of include/icem_hip.h: argument checking (the reference's
// ValueError / AttributeError / NotImplementedError cases as ICEM_E_* codes), the host-side colored-noise tables,
// model / cost registration; the kernels behind the operators live in generic_kernels.hip (gk_*) and k_*.hip.
#include "host_common.h"
#include "cost_terms_dev.h"
#include "icem_rssm.h"

using namespace icem;

namespace {

void psd_scale_host(int h, double beta, std::vector<double>& s, double& sigma) {
    // colorednoise.powerlaw_psd_gaussian (third-party, call site icem.py:73): f = rfftfreq(h),
    // DC takes the first bin's value, s = f^(-beta/2), sigma = 2*sqrt(sum w^2)/h.
    const int F = h / 2 + 1;
    s.resize(F);
    for (int k = 0; k < F; ++k) s[k] = (double)k * (1.0 / (double)h);
    const double fmin = 1.0 / (double)h;
    int ix = 0;
    for (int k = 0; k < F; ++k) ix += s[k] < fmin ? 1 : 0;
    if (ix && ix < F)
        for (int k = 0; k < ix; ++k) s[k] = s[ix];
    for (int k = 0; k < F; ++k) s[k] = std::pow(s[k], -beta / 2.0);
    double acc = 0.0;
    for (int k = 1; k < F; ++k) {
        double w = s[k];
        if (k == F - 1) w *= (1 + (h % 2)) / 2.0;
        acc += w * w;
    }
    sigma = 2.0 * std::sqrt(acc) / (double)h;
}

void noise_tables(int h, double beta, std::vector<double>& cr, std::vector<double>& ci) {
    std::vector<double> s;
    double sigma;
    psd_scale_host(h, beta, s, sigma);
    const int F = h / 2 + 1;
    cr.assign((size_t)F * h, 0.0);
    ci.assign((size_t)F * h, 0.0);
    for (int k = 0; k < F; ++k) {
        double mult = 2.0;
        if (k == 0 || (h % 2 == 0 && k == F - 1)) mult = 1.0;
        const double amp = mult * s[k] / ((double)h * sigma);
        const bool imag_dropped = (k == 0) || (h % 2 == 0 && k == F - 1);
        for (int t = 0; t < h; ++t) {
            const double ang = 2.0 * M_PI * (double)k * (double)t / (double)h;
            cr[(size_t)k * h + t] = amp * std::cos(ang);
            ci[(size_t)k * h + t] = imag_dropped ? 0.0 : -amp * std::sin(ang);
        }
    }
}

std::vector<int> population_sizes(const icem_config& c) {
    std::vector<int> out;
    int n = c.num_traj;
    for (int i = 0; i < c.opt_iters; ++i) {
        if (i > 0) n = std::max(c.elites_size * 2, (int)((double)n / c.factor_decrease));
        out.push_back(n);
    }
    return out;
}

template <typename T>
int upload(void** dev, const std::vector<double>& host) {
    std::vector<T> tmp(host.size());
    for (size_t i = 0; i < host.size(); ++i) tmp[i] = (T)host[i];
    if (*dev) {
        (void)hipFree(*dev);
        *dev = nullptr;
    }
    ICEM_HIP_TRY(hipMalloc(dev, tmp.size() * sizeof(T)));
    ICEM_HIP_TRY(hipMemcpy(*dev, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return ICEM_OK;
}

// ---- development options: one process-wide table (options.h) -----------------------------------------------------------
struct OptRow {
    const char* name;
    double def;
};
const OptRow OPT_ROWS[OPT_COUNT] = {
#define X(id, name, def) {name, def},
    ICEM_OPTIONS(X)
#undef X
};
std::atomic<double> g_opt[OPT_COUNT] = {
#define X(id, name, def) {def},
    ICEM_OPTIONS(X)
#undef X
};

// Worst case, over every start observation with |entries| <= m and every action sequence with |entries| <= m, of
// |state entry| / m inside `horizon` steps of the LINEAR model x' = x A + a B (A [o, o], B [d, o], row vectors):
//   x_t = x_0 A^t + sum_{s < t} a_s B A^(t-1-s)   =>   |x_t[j]| <= m (sum_k |A^t[k][j]| + sum_{s < t} sum_k |(B A^s)[k][j]|),
// attained for every (t, j) by a sign pattern of x_0 and bang-bang actions: the bound is the reachable maximum, not an
// estimate.  What the fp16-plane tiles need to know: their operands are x S with S m in [16, 32), fp16 ends at 65 504.
double linear_growth_bound(int o, int d, int horizon, const std::vector<double>& A, const std::vector<double>& B) {
    std::vector<double> P((size_t)o * o, 0.0), Q((size_t)d * o), T2, colB(o, 0.0);
    for (int k = 0; k < o; ++k) P[(size_t)k * o + k] = 1.0;
    Q.assign(B.begin(), B.begin() + (size_t)d * o);
    double worst = 1.0;
    auto times_A = [&](std::vector<double>& M, int rows) {
        T2.assign((size_t)rows * o, 0.0);
        for (int r = 0; r < rows; ++r)
            for (int k = 0; k < o; ++k) {
                const double v = M[(size_t)r * o + k];
                if (v == 0.0) continue;
                for (int j = 0; j < o; ++j) T2[(size_t)r * o + j] += v * A[(size_t)k * o + j];
            }
        M.swap(T2);
    };
    for (int t = 1; t <= horizon; ++t) {
        for (int j = 0; j < o; ++j)
            for (int k = 0; k < d; ++k) colB[j] += std::fabs(Q[(size_t)k * o + j]);   // += |B A^(t-1)| column sums
        times_A(P, o);   // A^t
        times_A(Q, d);   // B A^t
        for (int j = 0; j < o; ++j) {
            double g = colB[j];
            for (int k = 0; k < o; ++k) g += std::fabs(P[(size_t)k * o + j]);
            if (!(g < 1e300)) return INFINITY;
            worst = std::max(worst, g);
        }
    }
    return worst;
}

int pick_O(int o) {
    const int sizes[] = {8, 16, 17, 18, 24, 32};
    for (int s : sizes)
        if (o <= s) return s;
    return -1;
}

}  // namespace

namespace icem {
double opt(Opt k) { return g_opt[k].load(std::memory_order_relaxed); }

// fp16-plane tile arithmetic: the launch's scale needs the action bounds' magnitude -- fetched once per (low, high) buffer
// pair (check_plan) and again at every icem_reset_distribution (where a caller may have rewritten the bounds in place)
int refresh_act_mag(icem_handle* h, const void* low, const void* high, hipStream_t st, bool force) {
    if (h->cfg.dtype != ICEM_F32 || !low || !high) return ICEM_OK;
    if (!force && h->am_lo == low && h->am_hi == high) return ICEM_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(ICEM_E_STATE, "the action bounds are fetched at the first call with a (low, high) buffer pair: make one call outside the "
                                  "stream capture first");
    const int d = h->cfg.act_dim;
    std::vector<float> lo(d), hi(d);
    ICEM_HIP_TRY(hipMemcpyAsync(lo.data(), low, d * sizeof(float), hipMemcpyDeviceToHost, st));
    ICEM_HIP_TRY(hipMemcpyAsync(hi.data(), high, d * sizeof(float), hipMemcpyDeviceToHost, st));
    ICEM_HIP_TRY(hipStreamSynchronize(st));
    float m = 0.f;
    for (int j = 0; j < d; ++j) m = std::max(m, std::max(std::fabs(lo[j]), std::fabs(hi[j])));
    h->act_mag = (m == m && m < 1e30f) ? m : 1.f;
    h->am_lo = low;
    h->am_hi = high;
    return ICEM_OK;
}
}  // namespace icem

// what icem_profile_overhead times: one wave that spins for `ticks` of the 100 MHz wall clock and reports how long it
// really ran (a kernel of KNOWN duration; an empty one would overstate the bracket: the command processor sets up the
// closing event while a real kernel is still running)
__global__ void profile_spin_kernel(long long ticks, long long* ran) {
    long long t0 = wall_clock64(), t = t0;
    // (bounded: a wall clock that does not advance must not hang the stream -- every read is at least a few clocks)
    for (long long polls = 0; t - t0 < ticks && polls < (1ll << 26); ++polls) t = wall_clock64();
    if (threadIdx.x == 0) *ran = t - t0;
}

extern "C" {

int icem_abi_version(void) { return ICEM_ABI_VERSION; }

#ifndef ICEM_BUILD_HASH
#define ICEM_BUILD_HASH "0000000000000000"  // built by hand, not through icem_amd/build.py
#endif
// the marker icem_amd/build.py looks for in the file (no dlopen needed to tell a stale binary)
const char* icem_build_hash(void) {
    static const char tag[] = "ICEM_BUILD_HASH=" ICEM_BUILD_HASH;
    return tag + 16;
}

const char* icem_last_error(void) { return g_err.c_str(); }

int icem_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int icem_noise_tables_host(int32_t horizon, double beta, double* cr_host, double* ci_host) {
    if (horizon < 2 || !cr_host || !ci_host) return fail(ICEM_E_INVALID, "bad horizon / null output");
    std::vector<double> cr, ci;
    noise_tables(horizon, beta, cr, ci);
    std::memcpy(cr_host, cr.data(), cr.size() * sizeof(double));
    std::memcpy(ci_host, ci.data(), ci.size() * sizeof(double));
    return ICEM_OK;
}

int icem_create(const icem_config* cfg, icem_handle** out) {
    if (!cfg || !out) return fail(ICEM_E_INVALID, "null argument");
    const icem_config& c = *cfg;
    if (c.num_traj < 2) return fail(ICEM_E_INVALID, "At least two trajectories needed!");  // mpc.py:30-31
    if (c.horizon < 2 || c.horizon > ICEM_MAX_HORIZON) return fail(ICEM_E_UNSUPPORTED, "horizon must be in [2, 64]");
    if (c.act_dim < 1 || c.act_dim > ICEM_MAX_ACT_DIM) return fail(ICEM_E_UNSUPPORTED, "act_dim must be in [1, 64]");
    if (c.num_elites < 1 || c.num_elites > ICEM_MAX_ELITES) return fail(ICEM_E_UNSUPPORTED, "num_elites must be in [1, 64]");
    if (c.opt_iters < 1) return fail(ICEM_E_INVALID, "opt_iters < 1");
    if (c.dtype != ICEM_F32 && c.dtype != ICEM_F64) return fail(ICEM_E_INVALID, "dtype");
    if (c.rng_rounds != 10 && c.rng_rounds != 7) return fail(ICEM_E_INVALID, "rng_rounds must be 10 or 7");
    if (c.world < 1 || c.rank < 0 || c.rank >= c.world) return fail(ICEM_E_INVALID, "rank/world");
    if (c.noise_beta != c.noise_beta) return fail(ICEM_E_INVALID, "noise_beta is NaN");  // <= 0: white branch, icem.py:77
    if (!(c.factor_decrease >= 1.0)) return fail(ICEM_E_INVALID, "factor_decrease must be >= 1");
    if (c.cost_mode < 0 || c.cost_mode > 2)
        return fail(ICEM_E_UNSUPPORTED, "Implement method to compute cost along trajectory");  // abstract_controller.py:88-91
    if (icem_device_count() < 1) return fail(ICEM_E_NO_DEVICE, "no HIP device visible");
    icem_handle* h = new icem_handle();
    h->cfg = c;
    h->F = c.horizon / 2 + 1;
    h->HMAX = c.horizon <= 32 ? 32 : 64;
    h->hd = c.horizon * c.act_dim;
    h->tsize = c.dtype == ICEM_F64 ? 8 : 4;
    h->pop = population_sizes(c);
    h->n_reuse = (int)((double)c.num_elites * c.fraction_reused);  // int(len(elites)*xi), icem.py:98,145
    // the population can GROW after iteration 0 when N < 2*elites_size (icem.py:127 floors N_i at 2*elites_size)
    h->n_local_max = 0;
    for (int n_it : h->pop) h->n_local_max = std::max(h->n_local_max, shard_chunk(n_it, c.world));
    h->use_fast = opt_i(OPT_DISABLE_FAST) == 0;
    // synthesis table W[t][m]: m < F real part of bin m, F <= m < h imaginary part of bin m-F+1
    // noise_beta <= 0 is the reference's white branch (np.random.randn(N, h, d), icem.py:77): draw t of a row is
    // its sample at step t, i.e. the identity table
    std::vector<double> cr, ci, W((size_t)c.horizon * h->HMAX, 0.0);
    if (c.noise_beta > 0) {
        noise_tables(c.horizon, c.noise_beta, cr, ci);
        for (int t = 0; t < c.horizon; ++t)
            for (int m = 0; m < c.horizon; ++m)
                W[(size_t)t * h->HMAX + m] = m < h->F ? cr[(size_t)m * c.horizon + t] : ci[(size_t)(m - h->F + 1) * c.horizon + t];
    } else {
        for (int t = 0; t < c.horizon; ++t) W[(size_t)t * h->HMAX + t] = 1.0;
    }
    int rc = c.dtype == ICEM_F64 ? upload<double>(&h->W_dev, W) : upload<float>(&h->W_dev, W);
    if (!rc && (hipMalloc((void**)&h->nonfinite_dev, 2 * sizeof(unsigned)) != hipSuccess ||
                hipMemset(h->nonfinite_dev, 0, 2 * sizeof(unsigned)) != hipSuccess))
        rc = fail(ICEM_E_HIP, "hipMalloc of the handle's status word failed");
    if (rc) {
        if (h->W_dev) (void)hipFree(h->W_dev);
        if (h->nonfinite_dev) (void)hipFree(h->nonfinite_dev);
        delete h;
        return rc;
    }
    *out = h;
    return ICEM_OK;
}

int icem_destroy(icem_handle* h) {
    if (!h) return ICEM_OK;
    xchg_destroy(h);
    rccl_release(h);
    ahead_destroy(h);
    if (h->W_dev) (void)hipFree(h->W_dev);
    if (h->nonfinite_dev) (void)hipFree(h->nonfinite_dev);
    if (h->batch_ctx && h->batch_ctx_free) h->batch_ctx_free(h->batch_ctx);
    for (void* p : {h->sx.state, h->sx.raw, h->sx.pre[0], h->sx.pre[1], h->sx.shift})
        if (p) (void)hipFree(p);
    if (h->actions_alt) (void)hipFree(h->actions_alt);
    if (h->host_stage) (void)hipHostFree(h->host_stage);
    if (h->ws_alt) (void)hipFree(h->ws_alt);
    if (h->pp_stats) (void)hipFree(h->pp_stats);
    if (h->A_dev) (void)hipFree(h->A_dev);
    if (h->B_dev) (void)hipFree(h->B_dev);
    if (h->Mp_dev) (void)hipFree(h->Mp_dev);
    if (h->Mw_dev) (void)hipFree(h->Mw_dev);
    if (h->Mws_dev) (void)hipFree(h->Mws_dev);
    if (h->Mwh_dev) (void)hipFree(h->Mwh_dev);
    if (h->Mwh_ksc_dev) (void)hipFree(h->Mwh_ksc_dev);
    if (h->wide_cs_dev) (void)hipFree(h->wide_cs_dev);
    if (h->hn_cs_dev) (void)hipFree(h->hn_cs_dev);
    if (h->pub_dev) (void)hipFree(h->pub_dev);
    if (h->perm_dev) (void)hipFree(h->perm_dev);
    for (auto& sp : h->spans) {
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    for (auto e : h->free_events) (void)hipEventDestroy(e);
    delete h;
    return ICEM_OK;
}

int icem_set_episode(icem_handle* h, uint64_t episode) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (episode >> 32) return fail(ICEM_E_INVALID, "episode must fit 32 bits");
    h->episode = episode;
    return ICEM_OK;
}

int icem_population_sizes(const icem_handle* h, int32_t* out_host) {
    if (!h || !out_host) return fail(ICEM_E_INVALID, "null argument");
    for (size_t i = 0; i < h->pop.size(); ++i) out_host[i] = h->pop[i];
    return ICEM_OK;
}

// which kernel family serves this handle's f32 rollout (icem_handle::Of)
static void update_paths(icem_handle* h) {
    const bool tile = !h->wide && h->has_model && !h->has_terms && !(h->has_cost && h->cost.lin_weight == 0.0) &&
                      fast_rollout_supported(h->cfg.horizon, h->cfg.act_dim, h->O, 1);
    const int of = tile ? h->O : 0;
    if (of != h->Of) h->fast_model_ready = false;
    h->Of = of;
    // the tile's arithmetic (icem_set_tile_arith).  The fp16-plane tile exists for one-tile widths (O <= 20) and a flip
    // threshold >= 0, carries the model's planes scaled by one power of two for A and one for B (largest |entry| into
    // [64, 128): Tile16H) and is what large populations roll out with by default -- decided from the GLOBAL populations of
    // the configuration, so every rank and every launch of a handle computes in one arithmetic.
    bool split_ok = of >= 16 && of <= 20 && h->cfg.dtype == ICEM_F32 && !(h->cost.flip_idx >= 0 && h->cost.flip_thresh < 0.0);
    auto lift = [](const std::vector<double>& m, bool* finite) {
        double mx = 0.0;
        for (double v : m) {
            mx = std::max(mx, std::fabs(v));
            if (!(std::fabs(v) < 1e30)) *finite = false;
        }
        if (!(mx > 1e-30)) return 1.f;
        int e = 0;
        (void)std::frexp(mx, &e);            // mx = f 2^e, f in [0.5, 1)
        return (float)std::ldexp(1.0, 7 - e);  // mx x scale in [64, 128)
    };
    bool finite = true;
    h->tile_m_scale = lift(h->A_host, &finite);
    h->tile_b_scale = lift(h->B_host, &finite);
    // ... and the planes are served only where the model cannot take a state out of fp16's range, whatever the actions: the
    // operands are x S with S max(|obs0|, action bound) in [16, 32), so a state may grow to 2^11 times that magnitude before
    // f16(x S) is infinite -- and an infinite operand is a NaN cost where the reference ranks a finite one (icem.py:147-159,
    // 199).  linear_growth_bound is the REACHABLE maximum over the horizon (a tanh model's state is bounded by 1); beyond
    // 2^10 (a factor of two in hand) the handle computes on the exact tile.  The action operand a S sM / sB overflows where
    // B's largest entry is more than 2^10 x A's (a small B only costs ABSOLUTE accuracy, 2^-28 of the state's scale).
    const bool narrow_f32 = !h->wide && h->has_model && h->cfg.dtype == ICEM_F32 && finite;
    const bool hn_shape = h->has_model && h->cfg.dtype == ICEM_F32 && finite && h->obs_dim <= 48;
    h->tile_growth = 1.0;
    h->tile_ratio_log2 = 0;
    if ((narrow_f32 || hn_shape) && !h->A_host.empty()) {
        if (h->model_kind == ICEM_MODEL_LINEAR)
            h->tile_growth = linear_growth_bound(h->obs_dim, h->cfg.act_dim, h->cfg.horizon, h->A_host, h->B_host);
        int ea = 0, eb = 0;
        (void)std::frexp((double)h->tile_m_scale, &ea);   // scale = 2^(7 - e_max): log2(sM / sB) = e_max(B) - e_max(A)
        (void)std::frexp((double)h->tile_b_scale, &eb);
        double bmax = 0.0;
        for (double v : h->B_host) bmax = std::max(bmax, std::fabs(v));
        h->tile_ratio_log2 = bmax > 1e-30 ? ea - eb : 0;   // (a model without actions: nothing to scale)
    }
    const bool range_ok = finite && h->tile_growth <= 1024.0 && h->tile_ratio_log2 <= 10;
    split_ok = split_ok && range_ok;
    // (the arithmetic is the HANDLE's: no environment variable takes part -- the ranks of a sharded run are separate
    //  processes and must agree on it from the configuration alone)
    const int mode = h->tile_arith_mode;
    // by configuration (AUTO): the planes wherever they are served.  (ABI 3 kept populations of at most 8192 rows on the exact
    // tile's VALU twin, four waves per tile, on the argument that a lone wave's MFMA chain is the longer latency chain; measured
    // -- EXPERIMENTS R5.5 -- one Tile16H wave per tile is the SHORTER chain: 66.7 -> 61.9 us per MPC step at N = 4096, 84.1 ->
    // 74.6 at 8192.)  Still decided from the configuration alone, never from a rank's or a launch's row count.
    const bool want = mode == 1 || mode < 0;
    h->tile_arith = (split_ok && want) ? 1 : 0;
    // the reference's other narrow shapes (Door, Relocate, FetchPickAndPlace) have ONE fast rollout, and it computes in the
    // fp16 planes: theirs unless the exact arithmetic is asked for (icem_set_tile_arith 0: the exact-f32 GEMM kernel)
    // (the term list must fit one of the kernel's compiled programs: slices of at most 32 entries)
    int n32 = 0, n4 = 0, np = 0;
    bool terms_ok = true;
    if (h->has_terms && (h->terms.diff_idx >= 0 || h->terms.health_idx >= 0)) terms_ok = false;   // (Ant / Hopper / Humanoid: the GEMM kernel)
    if (h->has_terms)
        for (int j = 0; j < h->terms.n_terms; ++j) {
            const icem_cost_term& tm = h->terms.terms[j];
            if (tm.kind == ICEM_TERM_STEP_GT || tm.kind == ICEM_TERM_SQ_OFFSET) ++np;
            else if (tm.len <= 4) ++n4;
            else if (tm.len <= 32) ++n32;
            else terms_ok = false;
        }
    int prog[3] = {0, 0, 0};
    // (o = 39 is a WIDE observation: a caller that asked for the exact arithmetic there -- icem_set_wide_arith(ICEM_WIDE_F32),
    //  icem_set_wide_exact(1) -- gets the exact-f32 GEMM kernel, like one that passed ICEM_TILE_F32)
    const bool hn = of == 0 && h->has_model && h->has_cost && h->cfg.dtype == ICEM_F32 && range_ok && mode != 0 && terms_ok &&
                    !(h->wide && h->wide_mode == ICEM_WIDE_F32) &&
                    hn_cost_program(n32, n4, np, prog) &&
                    hn_rollout_supported(h->cfg.horizon, h->cfg.act_dim, h->obs_dim, h->cfg.num_elites);
    for (int k = 0; k < 3; ++k) h->hn_prog[k] = hn ? prog[k] : 0;
    if (hn != h->hn_tile) h->fast_model_ready = false;
    h->hn_tile = hn;
    // wide observations (icem_set_wide_arith): AUTO = the fp16 planes, unless one sweep of balancing leaves a row or column
    // of the model more than 2^13 below its largest weight (wide_model_imbalance_log2) -- then the bf16 planes, whose
    // operands are exact at any magnitude; a width the split kernel's LDS does not hold computes in exact f32 whatever is asked
    int eff = 0;
    if (h->wide && h->has_model) {
        eff = h->wide_mode >= 0 ? h->wide_mode : (h->wide_imbalance > 13 ? 2 : 0);
        if (!wide_split_fits(h->obs_dim, h->cfg.act_dim)) eff = 1;
    } else if (gemm_rollout(h)) {
        eff = 1;   // narrow models on the GEMM kernel: the exact-f32 form (two workgroup barriers per step buy nothing at o <= 32)
    }
    if (eff != h->wide_eff) h->fast_model_ready = false;
    h->wide_eff = eff;
}

static int sync_wide_cost(icem_handle* h);

int icem_set_tile_arith(icem_handle* h, int32_t mode) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (mode < -1 || mode > 1) return fail(ICEM_E_INVALID, "tile arithmetic: -1 (by configuration), 0 (exact f32) or 1 (fp16 planes)");
    if (h->pm_pending || h->pk_pending) return fail(ICEM_E_STATE, "a deferred merge is pending: finish the MPC step first");
    h->tile_arith_mode = mode;
    update_paths(h);
    h->ahead.next_valid = h->ahead.pre_valid = false;   // (noise drawn ahead for the other launch shapes is redrawn)
    return sync_wide_cost(h);
}

int icem_tile_arith(const icem_handle* h) { return h ? ((h->tile_arith || h->hn_tile) ? 1 : 0) : 0; }

double icem_tile_growth(const icem_handle* h) { return h ? h->tile_growth : 0.0; }

int icem_set_option(const char* name, double value) {
    if (!name || !(value == value)) return fail(ICEM_E_INVALID, "null option name / NaN value");
    for (int k = 0; k < OPT_COUNT; ++k)
        if (std::strcmp(name, OPT_ROWS[k].name) == 0) {
            g_opt[k].store(value, std::memory_order_relaxed);
            return ICEM_OK;
        }
    return fail(ICEM_E_INVALID, std::string("unknown option: ") + name);
}

int icem_get_option(const char* name, double* value_out) {
    if (!name || !value_out) return fail(ICEM_E_INVALID, "null argument");
    for (int k = 0; k < OPT_COUNT; ++k)
        if (std::strcmp(name, OPT_ROWS[k].name) == 0) {
            *value_out = g_opt[k].load(std::memory_order_relaxed);
            return ICEM_OK;
        }
    return fail(ICEM_E_INVALID, std::string("unknown option: ") + name);
}

int icem_reset_options(void) {
    for (int k = 0; k < OPT_COUNT; ++k) g_opt[k].store(OPT_ROWS[k].def, std::memory_order_relaxed);
    return ICEM_OK;
}

const char* icem_option_name(int32_t index) { return (index >= 0 && index < OPT_COUNT) ? OPT_ROWS[index].name : nullptr; }

int icem_nonfinite_costs(icem_handle* h, int64_t* count_out, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!count_out) return fail(ICEM_E_INVALID, "null output");
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(ICEM_E_STATE, "icem_nonfinite_costs synchronises: not on a capturing stream");
    unsigned v = 0;
    ICEM_HIP_TRY(hipMemcpyAsync(&v, h->nonfinite_dev, sizeof(v), hipMemcpyDeviceToHost, st));
    ICEM_HIP_TRY(hipStreamSynchronize(st));
    *count_out = (int64_t)v;
    return ICEM_OK;
}

int icem_set_model(icem_handle* h, int32_t kind, int32_t obs_dim, const double* A_host, const double* B_host) {
    if (!h || !A_host || !B_host) return fail(ICEM_E_INVALID, "null argument");
    if (kind != ICEM_MODEL_LINEAR && kind != ICEM_MODEL_TANH) return fail(ICEM_E_INVALID, "model kind");
    const int d = h->cfg.act_dim;
    if (obs_dim > 32) {
        // wide observations (HumanoidStandup's real o = 378, mujoco.py:241-252): the f32 GEMM rollout only
        if (h->cfg.dtype != ICEM_F32 || !wide_rollout_supported(obs_dim, d, 1))
            return fail(ICEM_E_UNSUPPORTED, "obs_dim in (32, 384] needs dtype f32 (k_rollout_wide); beyond 384 is not compiled");
        if (h->cfg.num_elites > 32) return fail(ICEM_E_UNSUPPORTED, "obs_dim > 32 needs num_elites <= 32 (candidate lists of k_rollout_wide)");
        if (h->A_dev) (void)hipFree(h->A_dev);
        if (h->B_dev) (void)hipFree(h->B_dev);
        h->A_dev = h->B_dev = nullptr;
        h->model_kind = kind;
        h->obs_dim = obs_dim;
        h->O = 0;
        h->wide = true;
        h->has_model = true;
        h->A_host.assign(A_host, A_host + (size_t)obs_dim * obs_dim);
        h->B_host.assign(B_host, B_host + (size_t)d * obs_dim);
        h->wide_imbalance = wide_model_imbalance_log2(obs_dim, d, h->A_host.data(), h->B_host.data());
        h->fast_model_ready = false;
        update_paths(h);
        return sync_wide_cost(h);
    }
    h->wide = false;
    const int O = pick_O(obs_dim);
    if (obs_dim < 1 || O < 0) return fail(ICEM_E_UNSUPPORTED, "obs_dim must be in [1, 384]");
    std::vector<double> A((size_t)O * O, 0.0), B((size_t)d * O, 0.0);
    for (int k = 0; k < obs_dim; ++k)
        for (int i = 0; i < obs_dim; ++i) A[(size_t)k * O + i] = A_host[(size_t)k * obs_dim + i];
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < obs_dim; ++i) B[(size_t)j * O + i] = B_host[(size_t)j * obs_dim + i];
    int rc = h->cfg.dtype == ICEM_F64 ? upload<double>(&h->A_dev, A) : upload<float>(&h->A_dev, A);
    if (rc) return rc;
    rc = h->cfg.dtype == ICEM_F64 ? upload<double>(&h->B_dev, B) : upload<float>(&h->B_dev, B);
    if (rc) return rc;
    h->model_kind = kind;
    h->obs_dim = obs_dim;
    h->O = O;
    h->has_model = true;
    h->A_host.assign(A_host, A_host + (size_t)obs_dim * obs_dim);
    h->B_host.assign(B_host, B_host + (size_t)d * obs_dim);
    h->fast_model_ready = false;
    update_paths(h);
    return sync_wide_cost(h);
}

// the device copy of the cost the wide rollout kernels read when cost terms are on (by value in the argument block it
// costs them 200 spilled scalar registers -- and 9 % of a launch -- whether a term is on or not)
static int sync_wide_cost(icem_handle* h) {
    if (!h->has_terms) return ICEM_OK;
    CostArgs<float> cs;
    fill_cost_args_f32(h, cs);
    if (h->hn_tile) {   // TileHN's copy: the terms sorted into its program's slots (long slices, short slices, points), null-padded
        CostArgs<float> hs = cs;
        const int cap[3] = {h->hn_prog[0], h->hn_prog[1], h->hn_prog[2]}, base[3] = {0, cap[0], cap[0] + cap[1]};
        int used[3] = {0, 0, 0};
        for (int j = 0; j < ICEM_MAX_COST_TERMS; ++j) {
            hs.terms[j] = CostArgs<float>::Term{0.f, 0.f, 0.f, -1, 0, -1, 1, -1};
        }
        for (int j = 0; j < cs.n_terms; ++j) {
            const auto& tm = cs.terms[j];
            const int cls = (tm.kind == ICEM_TERM_STEP_GT || tm.kind == ICEM_TERM_SQ_OFFSET) ? 2 : (tm.len <= 4 ? 1 : 0);
            if (used[cls] < cap[cls]) hs.terms[base[cls] + used[cls]++] = tm;
        }
        hs.n_terms = cap[0] + cap[1] + cap[2];
        if (!h->hn_cs_dev) ICEM_HIP_TRY(hipMalloc(&h->hn_cs_dev, sizeof(hs)));
        ICEM_HIP_TRY(hipMemcpy(h->hn_cs_dev, &hs, sizeof(hs), hipMemcpyHostToDevice));
    }
    if (!h->wide_cs_dev) ICEM_HIP_TRY(hipMalloc(&h->wide_cs_dev, sizeof(cs)));
    ICEM_HIP_TRY(hipMemcpy(h->wide_cs_dev, &cs, sizeof(cs), hipMemcpyHostToDevice));
    return ICEM_OK;
}

int icem_set_cost(icem_handle* h, const icem_cost_spec* spec) {
    if (!h || !spec) return fail(ICEM_E_INVALID, "null argument");
    h->cost = *spec;
    h->has_cost = true;
    // the tile kernels' packed model is permuted by the cost's columns; the GEMM kernels' is not: a cost-only change leaves
    // their packing alone unless the kernel family flips (update_paths clears the flag itself then)
    const bool was_gemm = gemm_rollout(h);
    update_paths(h);
    if (!(was_gemm && gemm_rollout(h))) h->fast_model_ready = false;
    return sync_wide_cost(h);
}

int icem_set_cost_terms(icem_handle* h, const icem_cost_terms* terms) {
    if (!h) return fail(ICEM_E_INVALID, "null argument");
    if (terms == nullptr) {
        h->has_terms = false;
        update_paths(h);
        return ICEM_OK;
    }
    if (terms->n_terms < 0 || terms->n_terms > ICEM_MAX_COST_TERMS) return fail(ICEM_E_INVALID, "n_terms must be in [0, 8]");
    for (int j = 0; j < terms->n_terms; ++j) {
        const icem_cost_term& tm = terms->terms[j];
        if (tm.kind < ICEM_TERM_NORM || tm.kind > ICEM_TERM_STEP_GT) return fail(ICEM_E_INVALID, "unknown cost term kind");
        if (tm.len < 1 || tm.len > ICEM_MAX_TERM_LEN) return fail(ICEM_E_INVALID, "cost term len must be in [1, 64]");
    }
    if (terms->box_from >= 0 && terms->health_idx < 0)
        return fail(ICEM_E_INVALID, "box_from is part of the health term: health_idx must be set");
    const bool on = terms->diff_idx >= 0 || terms->health_idx >= 0 || terms->n_terms > 0;
    h->terms = *terms;
    h->has_terms = on;
    update_paths(h);
    return sync_wide_cost(h);
}

int icem_trajectory_cost(icem_handle* h, int32_t n, int32_t obs_dim, const void* observations,
                         const void* next_observations, int64_t traj_stride, int64_t step_stride, const void* actions,
                         void* costs, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!h->has_cost) return fail(ICEM_E_STATE, "icem_set_cost must be called first");
    if (n < 0 || obs_dim < 1 || !observations || !actions || !costs) return fail(ICEM_E_INVALID, "null tensor / bad n or obs_dim");
    if (const char* e = cost_indices_error(h, obs_dim)) return fail(ICEM_E_INVALID, e);
    if (h->has_terms && h->terms.diff_idx >= 0 && !next_observations)
        return fail(ICEM_E_INVALID, "the difference term needs next_observations");
    if (n == 0) return ICEM_OK;
    hipStream_t st = (hipStream_t)stream;
    return gk_trajectory_cost(h, n, obs_dim, observations, next_observations, traj_stride, step_stride, actions, costs, st);
}

int icem_sample_clip(icem_handle* h, int32_t n, int64_t first_index, const void* mean, const void* std,
                     const void* low, const void* high, const void* z_r, const void* z_i, uint64_t offset,
                     int32_t t_begin, int32_t row0_mean, void* actions, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || !mean || !std || !low || !high || !actions) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (h->cfg.noise_beta > 0 && (z_r == nullptr) != (z_i == nullptr))
        return fail(ICEM_E_INVALID, "z_r and z_i must both be given or both NULL");
    if (t_begin < 0 || t_begin >= h->cfg.horizon) return fail(ICEM_E_INVALID, "t_begin out of range");
    hipStream_t st = (hipStream_t)stream;
    if (z_r == nullptr && t_begin == 0 && fast_sample_ok(h))
        return launch_fast_sample(h, n, first_index, mean, std, low, high, offset, row0_mean, actions, st);
    return gk_sample(h, n, first_index, mean, std, low, high, z_r, z_i, offset, t_begin, row0_mean, actions, st);
}

int icem_sample_truncnorm(icem_handle* h, int32_t n, int64_t first_index, const void* mean, const void* std,
                          const void* lower, const void* upper, const void* u, uint64_t offset, void* actions,
                          void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || !mean || !std || !lower || !upper || !actions) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (n == 0) return ICEM_OK;
    hipStream_t st = (hipStream_t)stream;
    return gk_sample_truncnorm(h, n, first_index, mean, std, lower, upper, u, offset, actions, st);
}

int icem_sample_piecewise(icem_handle* h, int32_t n, int64_t call_offset, int32_t change_freq, int64_t first_block,
                          const void* low, const void* high, const void* u, void* actions, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || call_offset < 0 || change_freq < 0 || !low || !high || !actions)
        return fail(ICEM_E_INVALID, "null tensor / negative n, call_offset or change_freq");
    if (n == 0) return ICEM_OK;
    return gk_sample_piecewise(h, n, call_offset, change_freq, first_block, low, high, u, actions, (hipStream_t)stream);
}

int icem_cem_bounds(icem_handle* h, int32_t like_levine, const void* mean, void* std, const void* low, const void* high,
                    void* lower, void* upper, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!mean || !std || !low || !high || !lower || !upper) return fail(ICEM_E_INVALID, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    return gk_cem_bounds(h, like_levine, mean, std, low, high, lower, upper, st);
}

int icem_philox_normals(icem_handle* h, int32_t n, int64_t first_index, uint64_t offset, void* z_r, void* z_i,
                        void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n <= 0 || !z_r || !z_i) return fail(ICEM_E_INVALID, "null tensor / n <= 0");
    hipStream_t st = (hipStream_t)stream;
    return gk_philox_normals(h, n, first_index, offset, z_r, z_i, st);
}

int icem_rollout_cost(icem_handle* h, int32_t n, const void* obs0, const void* actions, void* costs,
                      void* observations, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!h->has_model || !h->has_cost) return fail(ICEM_E_STATE, "icem_set_model / icem_set_cost must be called first");
    if (n < 0 || !obs0 || !actions || !costs) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (const char* e = cost_indices_error(h, h->obs_dim)) return fail(ICEM_E_INVALID, e);
    if (const char* e = wide_unsupported(h, 0, false, observations != nullptr)) return fail(ICEM_E_UNSUPPORTED, e);
    hipStream_t st = (hipStream_t)stream;
    if (observations == nullptr && n > 0 && fast_rollout_ok(h, 0))
        return launch_fast_rollout(h, n, 0, 0, obs0, actions, costs, nullptr, nullptr, st, nullptr);
    return gk_rollout(h, n, obs0, actions, costs, observations, st);
}

int icem_cost_reduce(icem_handle* h, int32_t n, const void* step_costs, void* costs, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 0 || !step_costs || !costs) return fail(ICEM_E_INVALID, "null tensor / negative n");
    if (n == 0) return ICEM_OK;
    hipStream_t st = (hipStream_t)stream;
    return gk_cost_reduce(h, n, step_costs, costs, st);
}

size_t icem_topk_workspace_bytes(const icem_handle* h, int32_t n, int32_t k) {
    if (!h || n < 1 || k < 1) return 0;
    return (size_t)topk_blocks(n) * (size_t)k * (h->tsize + sizeof(int));
}

int icem_topk_sorted(icem_handle* h, int32_t n, const void* costs, int32_t k, void* out_cost, int32_t* out_idx,
                     void* workspace, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 1 || k < 1 || k > ICEM_MAX_ELITES || !costs || !out_cost || !out_idx || !workspace)
        return fail(ICEM_E_INVALID, "bad n/k or null tensor");
    hipStream_t st = (hipStream_t)stream;
    if (h->use_fast && h->cfg.dtype == ICEM_F32 && topk_small_ok(n, k)) {  // small f32 pools: one launch
        launch_topk_small((const float*)costs, n, k, (float*)out_cost, out_idx, st);
        ICEM_HIP_TRY(hipGetLastError());
        return ICEM_OK;
    }
    return gk_topk(h, n, k, costs, out_cost, out_idx, workspace, st);
}

int icem_update_distribution(icem_handle* h, int32_t n, const void* costs, const void* pool, int32_t n_keep,
                             const void* keep_costs, const void* keep_actions, int32_t k, void* mean, void* std,
                             void* elites_out, void* elite_costs_out, int32_t* idx_out, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (n < 1 || k < 1 || n_keep < 0 || !costs || !pool || !mean || !std || !elites_out || !elite_costs_out || !idx_out ||
        (n_keep > 0 && (!keep_costs || !keep_actions)))
        return fail(ICEM_E_INVALID, "bad n / k / n_keep or null tensor");
    if (elites_out == keep_actions) return fail(ICEM_E_INVALID, "elites_out must not alias keep_actions");
    if (!h->use_fast || h->cfg.dtype != ICEM_F32 || !topk_small_ok(n + n_keep, k))
        return fail(ICEM_E_UNSUPPORTED, "one-launch update: f32, n + n_keep <= 16384, k <= 32 (else icem_topk_sorted + icem_gather_refit)");
    UpdateSmallArgs a{(const float*)costs, (const float*)pool, (const float*)keep_costs, (const float*)keep_actions, n, n_keep, k,
                      h->hd, (float)h->cfg.alpha, (float*)mean, (float*)std, (float*)elites_out, (float*)elite_costs_out, idx_out};
    launch_update_small(a, (hipStream_t)stream);
    ICEM_HIP_TRY(hipGetLastError());
    return ICEM_OK;
}

int icem_update_distribution_ok(const icem_handle* h, int32_t n_all, int32_t k) {
    return (h && h->use_fast && h->cfg.dtype == ICEM_F32 && n_all >= 1 && k >= 1 && topk_small_ok(n_all, k)) ? 1 : 0;
}

int icem_gather_refit(icem_handle* h, const void* actions, const int32_t* idx, int32_t k, void* mean, void* std,
                      void* elites_out, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (k < 1 || !actions || !idx || !mean || !std) return fail(ICEM_E_INVALID, "bad k or null tensor");
    hipStream_t st = (hipStream_t)stream;
    return gk_gather_refit(h, actions, idx, k, mean, std, elites_out, st);
}

int icem_shift(icem_handle* h, void* mean, void* std, const void* low, const void* high, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!mean || !std || !low || !high) return fail(ICEM_E_INVALID, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    return gk_shift(h, mean, std, low, high, st);
}

int icem_reset_distribution(icem_handle* h, void* mean, void* std, const void* low, const void* high, void* stream) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!mean || !std || !low || !high) return fail(ICEM_E_INVALID, "null tensor");
    hipStream_t st = (hipStream_t)stream;
    // the large-population path keeps a host copy of the action bounds per (low, high) buffer pair: an episode start is
    // where a caller may have rewritten them in place, so the copy is fetched again at the next step
    h->ahead.lo_ptr = h->ahead.hi_ptr = nullptr;
    if (h->tile_arith || h->hn_tile) {   // ... and so does the fp16-plane tiles' scale (FastRolloutArgs::act_mag)
        const int rc = refresh_act_mag(h, low, high, st, true);
        if (rc) return rc;
    }
    return gk_reset(h, mean, std, low, high, st);
}

int icem_debug_stamps(icem_handle* h, void* dev_ptr) {
    if (!h) {   // the stateless learned-dynamics rollout's stamps
        rssm_set_stamps((long long*)dev_ptr);
        return ICEM_OK;
    }
    if (check_handle(h)) return ICEM_E_INVALID;
    h->dbg = (long long*)dev_ptr;
    return ICEM_OK;
}

int icem_set_wide_arith(icem_handle* h, int32_t mode) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (mode < ICEM_WIDE_AUTO || mode > ICEM_WIDE_BF16X3)
        return fail(ICEM_E_INVALID, "wide arithmetic: ICEM_WIDE_AUTO (-1), ICEM_WIDE_F16X2 (0), ICEM_WIDE_F32 (1) or ICEM_WIDE_BF16X3 (2)");
    if (h->pm_pending || h->pk_pending) return fail(ICEM_E_STATE, "a deferred merge is pending: finish the MPC step first");
    h->wide_mode = mode;
    update_paths(h);
    return ICEM_OK;
}

// (a TileHN handle -- Door / Relocate at o = 39 -- launches the fp16 planes of icem_set_tile_arith, whatever wide_eff says)
int icem_wide_arith(const icem_handle* h) { return h ? (h->hn_tile ? ICEM_WIDE_F16X2 : h->wide_eff) : 0; }

int icem_wide_imbalance_log2(const icem_handle* h) { return h ? h->wide_imbalance : 0; }

int icem_wide_model_imbalance_log2(int32_t obs_dim, int32_t act_dim, const double* A_host, const double* B_host) {
    if (obs_dim < 1 || act_dim < 0 || !A_host || (act_dim > 0 && !B_host)) return -1;
    return wide_model_imbalance_log2(obs_dim, act_dim, A_host, B_host);
}

int icem_set_wide_exact(icem_handle* h, int32_t on) {   // ABI <= 3 spelling of icem_set_wide_arith (0 / 1 / 2)
    if (check_handle(h)) return ICEM_E_INVALID;
    if (on < 0 || on > 2) return fail(ICEM_E_INVALID, "icem_set_wide_exact: 0 (fp16 planes), 1 (exact f32) or 2 (bf16 planes)");
    return icem_set_wide_arith(h, on);
}

int icem_profile_enable(icem_handle* h, int32_t on) {
    if (check_handle(h)) return ICEM_E_INVALID;
    h->profiling = on != 0;
    return ICEM_OK;
}

int icem_profile_read(icem_handle* h, double* total_ms, int64_t* launches, int64_t* units) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!total_ms || !launches || !units) return fail(ICEM_E_INVALID, "null output");
    for (int k = 0; k < ICEM_K_COUNT; ++k) {
        total_ms[k] = 0.0;
        launches[k] = 0;
        units[k] = 0;
    }
    for (auto& sp : h->spans) {
        ICEM_HIP_TRY(hipEventSynchronize(sp.b));
        float ms = 0.f;
        ICEM_HIP_TRY(hipEventElapsedTime(&ms, sp.a, sp.b));
        total_ms[sp.kind] += ms;
        launches[sp.kind] += 1;
        units[sp.kind] += sp.units;
        h->free_events.push_back(sp.a);
        h->free_events.push_back(sp.b);
    }
    h->spans.clear();
    return ICEM_OK;
}

int icem_profile_overhead(void* stream, int32_t reps, double spin_us, double* pair_us, double* kernel_us) {
    if (!pair_us || !kernel_us || reps < 3 || reps > 4096 || !(spin_us >= 0.0) || spin_us > 1e4)
        return fail(ICEM_E_INVALID, "null output / bad reps or spin_us");
    hipStream_t st = (hipStream_t)stream;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(ICEM_E_STATE, "icem_profile_overhead synchronises: not on a capturing stream");
    (void)hipGetLastError();
    hipEvent_t a = nullptr, b = nullptr;
    long long* ran = nullptr;
    // (one exit: whatever was created is destroyed there)
    hipError_t e0 = hipEventCreate(&a);
    if (e0 == hipSuccess) e0 = hipEventCreate(&b);
    if (e0 == hipSuccess) e0 = hipHostMalloc((void**)&ran, sizeof(long long), hipHostMallocMapped);
    std::vector<float> pair;
    std::vector<double> kern;
    int rc = e0 == hipSuccess ? ICEM_OK : fail(ICEM_E_HIP, hipGetErrorString(e0));
    for (int r = 0; r < reps + 8 && rc == ICEM_OK; ++r) {
        hipError_t e = hipEventRecord(a, st);
        hipLaunchKernelGGL(profile_spin_kernel, dim3(1), dim3(64), 0, st, (long long)(spin_us * 100.0), ran);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(b, st);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
        if (e != hipSuccess) rc = fail(ICEM_E_HIP, hipGetErrorString(e));
        if (r >= 8) {   // the first few pay for code-object loading
            pair.push_back(ms);
            kern.push_back((double)*ran * 0.01);
        }
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (ran) (void)hipHostFree(ran);
    if (rc != ICEM_OK) return rc;
    std::sort(pair.begin(), pair.end());
    std::sort(kern.begin(), kern.end());
    *pair_us = 1e3 * (double)pair[pair.size() / 2];
    *kernel_us = kern[kern.size() / 2];
    return ICEM_OK;
}

size_t icem_record_bytes(const icem_handle* h) { return h ? (size_t)(h->hd + 2) * h->tsize : 0; }

size_t icem_rssm_param_elems(void) { return rssm::TOTAL; }

int icem_rssm_trim(void) {
    rssm_split_trim();
    return ICEM_OK;
}

int icem_rssm_rollout_cost(int32_t n, int32_t horizon, int32_t cost_mode, const void* params, const void* obs0,
                           const void* actions, void* costs, void* stream) {
    if (n < 0 || horizon < 1 || cost_mode < ICEM_COST_SUM || cost_mode > ICEM_COST_FINAL || !params || !obs0 || !actions || !costs)
        return fail(ICEM_E_INVALID, "null tensor / bad n, horizon or cost_mode");
    const hipError_t e = launch_rssm_rollout(n, horizon, cost_mode, (const unsigned short*)params, (const float*)obs0,
                                             (const float*)actions, (float*)costs, (hipStream_t)stream);
    if (e == hipErrorStreamCaptureUnsupported)
        return fail(ICEM_E_STATE, "learned-dynamics rollout: the stream is capturing and this call would have to synchronise it (a larger "
                                  "population or horizon than the staging area holds, or recovery from a timed-out wait): make one call "
                                  "of this size outside the capture first");
    if (e == hipErrorLaunchTimeOut)
        return fail(ICEM_E_STATE, "learned-dynamics rollout: a reward workgroup of an EARLIER launch on this stream gave up waiting "
                                  "for its recurrence (that launch's costs are NaN); the staging flags were reset, nothing was "
                                  "launched by this call -- call again");
    ICEM_HIP_TRY(e);
    return ICEM_OK;
}

}  // extern "C"
