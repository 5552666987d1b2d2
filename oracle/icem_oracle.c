/*
 * icem_oracle.c -- plain-C restatement of one CEM iteration of the iCEM hot path.
 * TEST INFRASTRUCTURE ONLY: the checker and the timed CPU baseline ("port"), never the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * float64 throughout, like the reference.  Follows (paths relative to /root/reference):
 *   sampling        colorednoise.powerlaw_psd_gaussian (third-party, unpinned; call site
 *                   icem/controllers/icem.py:73-75) + clip (icem.py:79), via the synthesis matrices
 *   rollout + cost  icem/models/abstract_models.py:17-53, icem/controllers/abstract_controller.py:74-91,
 *                   icem/environments/mujoco.py:67-99 / 259-277 (parametric form)
 *   elites + refit  icem/controllers/icem.py:194-211 (argsort()[:K], mean, population std, momentum)
 * RNG: Philox4x32-R-seeded xoshiro128++ + Box-Muller, keyed exactly like the device path (see oracle/icem_oracle.py
 * philox_white_noise); pinned against the NumPy oracle by tests/test_oracle_c.py, which is itself
 * pinned against golden vectors captured from the reference.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1, int rounds) {
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[1] = (uint32_t)p1;
        c[3] = (uint32_t)p0;
        c[0] = n0;
        c[2] = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

static void box_muller(uint32_t xa, uint32_t xb, double* g0, double* g1) {
    const double u1 = ((double)xa + 0.5) * 0x1p-32;
    const double v = (double)xb * 0x1p-32;
    const double r = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * v;
    *g0 = r * cos(ang);
    *g1 = r * sin(ang);
}

/* h white normals of row (n, j): one Philox call seeds xoshiro128++; normals 2i, 2i+1 from words 2i, 2i+1.
 * g[m], m < F real part of bin m, m >= F imaginary part of bin m-F+1 */
static void white_row(uint64_t seed, uint64_t offset, uint32_t n, uint32_t j, int h, int rounds, double* g) {
    uint32_t s[4] = {n, j << 16, (uint32_t)offset, (uint32_t)(offset >> 32)};
    philox4x32(s, (uint32_t)seed, (uint32_t)(seed >> 32), rounds);
    double tmp[2];
    for (int m = 0; m < h; m += 2) {
        uint32_t x[2];
        for (int q = 0; q < 2; ++q) {
            const uint32_t sum = s[0] + s[3];
            x[q] = ((sum << 7) | (sum >> 25)) + s[0];
            const uint32_t t = s[1] << 9;
            s[2] ^= s[0];
            s[3] ^= s[1];
            s[1] ^= s[2];
            s[0] ^= s[3];
            s[2] ^= t;
            s[3] = (s[3] << 11) | (s[3] >> 21);
        }
        box_muller(x[0], x[1], &tmp[0], &tmp[1]);
        g[m] = tmp[0];
        if (m + 1 < h) g[m + 1] = tmp[1];
    }
}

int icem_c_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void icem_c_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* colorednoise synthesis tables: cr, ci [F, h] */
void icem_c_noise_tables(int h, double beta, double* cr, double* ci) {
    const int F = h / 2 + 1;
    double* s = (double*)malloc(sizeof(double) * F);
    for (int k = 0; k < F; ++k) s[k] = (double)k * (1.0 / (double)h);
    const double fmin = 1.0 / (double)h;
    int ix = 0;
    for (int k = 0; k < F; ++k) ix += s[k] < fmin;
    if (ix && ix < F)
        for (int k = 0; k < ix; ++k) s[k] = s[ix];
    for (int k = 0; k < F; ++k) s[k] = pow(s[k], -beta / 2.0);
    double acc = 0;
    for (int k = 1; k < F; ++k) {
        double w = s[k];
        if (k == F - 1) w *= (1 + (h % 2)) / 2.0;
        acc += w * w;
    }
    const double sigma = 2.0 * sqrt(acc) / (double)h;
    for (int k = 0; k < F; ++k) {
        const int edge = (k == 0) || (h % 2 == 0 && k == F - 1);
        const double amp = (edge ? 1.0 : 2.0) * s[k] / ((double)h * sigma);
        for (int t = 0; t < h; ++t) {
            const double ang = 2.0 * M_PI * (double)k * (double)t / (double)h;
            cr[k * h + t] = amp * cos(ang);
            ci[k * h + t] = edge ? 0.0 : -amp * sin(ang);
        }
    }
    free(s);
}

/* K1: actions[n, h, d] = clip(colored * std + mean, low, high); Philox keyed by first_index + i. */
void icem_c_sample_clip(int n, int h, int d, double beta, uint64_t seed, uint64_t offset, int64_t first_index,
                        int rounds, const double* mean, const double* std, const double* low, const double* high,
                        int row0_mean, double* actions) {
    const int F = h / 2 + 1;
    double* cr = (double*)malloc(sizeof(double) * F * h);
    double* ci = (double*)malloc(sizeof(double) * F * h);
    if (beta > 0) icem_c_noise_tables(h, beta, cr, ci);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double g[256], y[256];
        for (int j = 0; j < d; ++j) {
            white_row(seed, offset, (uint32_t)(first_index + i), (uint32_t)j, h, rounds, g);
            if (beta > 0) {
                for (int t = 0; t < h; ++t) y[t] = 0.0;
                for (int k = 0; k < F; ++k) {
                    const double zr = g[k];
                    const double zi = (k >= 1 && F + k - 1 < h) ? g[F + k - 1] : 0.0;
                    for (int t = 0; t < h; ++t) y[t] += zr * cr[k * h + t] + zi * ci[k * h + t];
                }
            } else {
                for (int t = 0; t < h; ++t) y[t] = g[t]; /* icem.py:77: white noise, draw t of the row is step t */
            }
            for (int t = 0; t < h; ++t) {
                double v = y[t] * std[t * d + j] + mean[t * d + j];
                v = v < low[j] ? low[j] : v;
                v = v > high[j] ? high[j] : v;
                actions[((size_t)i * h + t) * d + j] = v;
            }
        }
    }
    if (row0_mean && first_index == 0 && n > 0) memcpy(actions, mean, sizeof(double) * h * d);
    free(cr);
    free(ci);
}

/* K2: costs[n]; model o' = act(o.A + a.B), cost on the pre-action observation. kind 0 linear, 1 tanh;
 * mode 0 sum, 1 best, 2 final. */
void icem_c_rollout_cost(int n, int h, int d, int o, int kind, int mode, const double* A, const double* B,
                         const double* obs0, const double* actions, double ctrl_w, int lin_idx, double lin_w,
                         int flip_idx, double flip_pen, double flip_th, double* costs) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double obs[512], nxt[512];  /* o <= 512 (HumanoidStandup: 378) */
        memcpy(obs, obs0, sizeof(double) * o);
        double acc = 0.0;
        for (int t = 0; t < h; ++t) {
            const double* a = actions + ((size_t)i * h + t) * d;
            for (int x = 0; x < o; ++x) nxt[x] = 0.0;
            for (int k = 0; k < o; ++k)
                for (int x = 0; x < o; ++x) nxt[x] += obs[k] * A[k * o + x];
            double ctrl = 0.0;
            for (int j = 0; j < d; ++j) {
                ctrl += a[j] * a[j];
                for (int x = 0; x < o; ++x) nxt[x] += a[j] * B[j * o + x];
            }
            double c = 0.0;
            if (flip_idx >= 0) {
                c += (obs[flip_idx] > flip_th) ? flip_pen : 0.0;
                c += (obs[flip_idx] < -flip_th) ? flip_pen : 0.0;
            }
            c += ctrl_w * ctrl;
            c += lin_w * obs[lin_idx];
            if (t == 0 || mode == 2)
                acc = c;
            else if (mode == 0)
                acc += c;
            else
                acc = c < acc ? c : acc;
            for (int x = 0; x < o; ++x) obs[x] = kind == 1 ? tanh(nxt[x]) : nxt[x];
        }
        costs[i] = acc;
    }
}

/* K3: k smallest (cost, index) ascending; NaN -> +inf.  O(n*k) selection. */
void icem_c_topk(int n, int k, const double* costs, int32_t* idx, double* out) {
    double pc = -INFINITY;
    int pi = -1;
    for (int r = 0; r < k; ++r) {
        double bc = INFINITY;
        int bi = INT32_MAX;
        for (int i = 0; i < n; ++i) {
            const double c = costs[i] != costs[i] ? INFINITY : costs[i];
            const int after = c > pc || (c == pc && i > pi);
            if (after && (c < bc || (c == bc && i < bi))) {
                bc = c;
                bi = i;
            }
        }
        idx[r] = bi;
        out[r] = bc;
        pc = bc;
        pi = bi;
    }
}

/* K4: refit over the K elite rows of actions[*, h, d] (icem.py:207-211). */
void icem_c_refit(int k, int hd, double alpha, const double* actions, const int32_t* idx, double* mean, double* std) {
    for (int e = 0; e < hd; ++e) {
        double s = 0.0;
        for (int r = 0; r < k; ++r) s += actions[(size_t)idx[r] * hd + e];
        const double m = s / k;
        double v = 0.0;
        for (int r = 0; r < k; ++r) {
            const double dx = actions[(size_t)idx[r] * hd + e] - m;
            v += dx * dx;
        }
        mean[e] = (1 - alpha) * m + alpha * mean[e];
        std[e] = (1 - alpha) * sqrt(v / k) + alpha * std[e];
    }
}

/* One whole CEM iteration (no elite keeping): sample -> rollout -> cost -> top-k -> refit. */
void icem_c_iteration(int n, int h, int d, int o, int k, double beta, double alpha, uint64_t seed, uint64_t offset,
                      int rounds, int kind, const double* A, const double* B, const double* obs0, const double* low,
                      const double* high, double ctrl_w, int lin_idx, double lin_w, int flip_idx, double flip_pen,
                      double flip_th, double* mean, double* std, double* actions, double* costs, int32_t* idx,
                      double* elite_costs) {
    icem_c_sample_clip(n, h, d, beta, seed, offset, 0, rounds, mean, std, low, high, 0, actions);
    icem_c_rollout_cost(n, h, d, o, kind, 0, A, B, obs0, actions, ctrl_w, lin_idx, lin_w, flip_idx, flip_pen, flip_th,
                        costs);
    icem_c_topk(n, k, costs, idx, elite_costs);
    icem_c_refit(k, h * d, alpha, actions, idx, mean, std);
}
