// icem_fused.hip -- the fused f32 CEM-iteration kernels for gfx950 (see icem_fused.h).
//
// fused_iter_kernel<H, D, O, L, KIND, ROUNDS>
//   Workgroup = 256 threads = 4 wavefronts, one tile of TPW trajectories per pass.
//   phase S  one thread per (trajectory, action-dim) row: Philox4x32 -> Box-Muller -> the h white
//            draws of the row in registers; inverse real DFT folded on its cos/sin symmetry
//            (t and h-t share the even sum and negate the odd one: ~h*h/2 FMAs instead of h*h),
//            table rows as wave-uniform scalar operands; affine + clip; samples parked in an LDS
//            tile laid out like the [n, h, d] output.
//   phase W  the tile goes to HBM as one contiguous, coalesced span (the reference's
//            `action_sequences`, icem/controllers/icem.py:73-79).
//   phase R  rollout + cost with L lanes per trajectory: each lane owns ceil(O/L) output columns
//            of the model, held in VGPRs for the whole kernel (no model traffic in the time
//            loop); the observation is exchanged inside the lane group with DPP quad permutes;
//            actions are read back from the LDS tile (never from HBM).
//   phase K  wave 0 bitonic-sorts {running top-K, this tile's (cost, index) keys} as 64 packed
//            u64 keys; after the last tile the workgroup emits its K sorted candidates.
// merge_single_kernel: 1024 threads, one per candidate list; K tournament rounds over the list
//   heads give the global sorted top-K; then gather + refit + epilogue (icem.py:163-211).
#include "icem_fused.h"

#include <climits>
#include <cmath>
#include <type_traits>
#include <utility>

#include "philox.h"

namespace icem {

namespace {

constexpr int HMAX = 32;

__host__ __device__ constexpr int cmin(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int tile_stride(int h, int d) {
    int s = h * d;
    s += s & 1;
    if (s % 32 == 0) s += 2;
    return s;
}
// trajectories per tile: one rollout lane each (<= 64), tile kept under 48 KiB of LDS
__host__ __device__ constexpr int tile_traj(int h, int d) { return cmin(64, (12288 / tile_stride(h, d)) & ~3); }

// ---- packed (cost, index) keys: unsigned order == (cost, index) lexicographic order ------------
__device__ __forceinline__ unsigned long long make_key(float c, int idx) {
    c = (c != c) ? INFINITY : c + 0.0f;  // NaN -> +inf; -0 -> +0
    unsigned u = __float_as_uint(c);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned)idx;
}
__device__ __forceinline__ float key_cost(unsigned long long k) {
    unsigned u = (unsigned)(k >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)(unsigned)k; }
constexpr unsigned long long KEY_SENTINEL = 0xFF8000007FFFFFFFull;  // (+inf, INT_MAX)

// ascending bitonic sort of one key per lane across the 64-lane wave
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long k, int lane) {
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j >= 1; j >>= 1) {
            const unsigned long long o = __shfl_xor(k, j, 64);
            const bool up = (lane & size) == 0;       // this block sorts ascending
            const bool lower = (lane & j) == 0;       // this lane keeps the smaller of the pair
            const bool take_min = (up == lower);
            const bool o_less = o < k;
            k = (take_min == o_less) ? o : k;
        }
    }
    return k;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_fn(float x, std::integral_constant<int, 0>) { return x; }
__device__ __forceinline__ float act_fn(float x, std::integral_constant<int, 1>) { return tanhf(x); }

// -------------------------------------------------------------------------------------------------
template <int H, int D, int O, int KIND, int ROUNDS>
__global__ __launch_bounds__(FUSED_WG) void fused_iter_kernel(FusedArgs a) {
    constexpr int F = H / 2 + 1;
    constexpr int HD = H * D;
    constexpr int TPW = tile_traj(H, D);
    constexpr int S = tile_stride(H, D);
    constexpr int CT = (O + 3) / 4;  // model output column tiles of 4
    constexpr int KK = O + D;        // contraction length of one model step: [o | a] . [A ; B]
    static_assert(H <= 31 && H >= 2, "folded DFT keeps real/imag halves in 16 + 16 registers");
    static_assert(TPW <= 64 && TPW % 4 == 0, "one rollout lane per trajectory, MFMA blocks of 4");

    __shared__ __attribute__((aligned(16))) float tile[TPW * S];
    __shared__ float ms_lds[2 * HD];  // mean | std, staged once: the S-phase reads them per sample

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int tiles_s = (a.n + TPW - 1) / TPW;
    const int tiles_x = (a.n_extra + TPW - 1) / TPW;
    const bool r_wave = tid < 64;  // wave 0 rolls the tile out on the matrix pipe

    // --- model operand of the MFMA, resident in VGPRs: lane holds M[k][4*ct + (lane & 3)] ---------
    // D^T = M^T . X^T with v_mfma_f32_4x4x1 (16 blocks of 4 trajectories): A-operand = 4 output
    // columns of the model (row i = lane & 3), B-operand = the lane's own trajectory value x_k,
    // result register i of tile ct = column 4*ct + i of THIS lane's trajectory: no transposes, the
    // accumulation is the same k-ordered fmaf chain as the scalar kernels.
    for (int e = tid; e < HD; e += FUSED_WG) {
        ms_lds[e] = a.mean[e];
        ms_lds[HD + e] = a.std[e];
    }
    __syncthreads();

    unsigned long long run_key = KEY_SENTINEL;  // wave 0: lane r < K holds the r-th best so far
    long long stamp[6] = {0, 0, 0, 0, 0, 0};
    if (a.dbg) stamp[0] = __builtin_readcyclecounter();

    for (int tile_id = blockIdx.x; tile_id < tiles_s + tiles_x; tile_id += gridDim.x) {
        const bool sampled = tile_id < tiles_s;
        const int n_base = sampled ? tile_id * TPW : a.n + (tile_id - tiles_s) * TPW;
        const int n_here = sampled ? cmin(TPW, a.n - n_base) : cmin(TPW, a.n + a.n_extra - n_base);
        float* gsrc = a.actions + (size_t)n_base * HD;

        if (sampled) {
            // ---------------- phase S ----------------
            for (int row = tid; row < n_here * D; row += FUSED_WG) {
                const int nl = row / D;
                const int j = row - nl * D;
                const unsigned gi = (unsigned)(a.first_index + n_base + nl);
                float g[HMAX];
#pragma unroll
                for (int b = 0; b < (H + 3) / 4; ++b) {
                    const U4 r = philox4x32<ROUNDS>(gi, ((unsigned)j << 16) | (unsigned)b, a.off_lo, a.off_hi,
                                                    a.seed_lo, a.seed_hi);
                    box_muller(r.x, r.y, g[4 * b], g[4 * b + 1]);
                    box_muller(r.z, r.w, g[4 * b + 2], g[4 * b + 3]);
                }
                const float lo = a.low[j], hi = a.high[j];
                float* trow = tile + nl * S + j;
                auto emit = [&](int t, float y) {
                    float v = __builtin_fmaf(y, ms_lds[HD + t * D + j], ms_lds[t * D + j]);
                    v = v < lo ? lo : v;
                    v = v > hi ? hi : v;
                    trow[t * D] = v;
                };
                {  // t = 0: every sine is zero
                    const float* __restrict__ w = a.W;
                    float e0 = 0.f, e1 = 0.f;
#pragma unroll
                    for (int m = 0; m < F; m += 2) {
                        e0 = __builtin_fmaf(g[m], w[m], e0);
                        if (m + 1 < F) e1 = __builtin_fmaf(g[m + 1], w[m + 1], e1);
                    }
                    emit(0, e0 + e1);
                }
#pragma unroll 1
                for (int tp = 1; tp <= H / 2; ++tp) {
                    const float* __restrict__ w = a.W + tp * HMAX;
                    float e0 = 0.f, e1 = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
                    for (int m = 0; m < F; m += 2) {
                        e0 = __builtin_fmaf(g[m], w[m], e0);
                        if (m + 1 < F) e1 = __builtin_fmaf(g[m + 1], w[m + 1], e1);
                    }
#pragma unroll
                    for (int m = F; m < H; m += 2) {
                        o0 = __builtin_fmaf(g[m], w[m], o0);
                        if (m + 1 < H) o1 = __builtin_fmaf(g[m + 1], w[m + 1], o1);
                    }
                    const float e = e0 + e1, od = o0 + o1;
                    emit(tp, e + od);
                    if (H - tp != tp) emit(H - tp, e - od);
                }
            }
            __syncthreads();
            if (a.dbg && tile_id == blockIdx.x) stamp[1] = __builtin_readcyclecounter();
            if (a.row0_mean && a.first_index == 0 && tile_id == 0) {  // icem.py:87-88
                for (int e = tid; e < HD; e += FUSED_WG) tile[e] = ms_lds[e];
                __syncthreads();
            }
            // ---------------- phase W ----------------
            for (int e = tid; e < n_here * HD; e += FUSED_WG) {
                const int nl = e / HD;
                gsrc[e] = tile[nl * S + (e - nl * HD)];
            }
        } else {
            for (int e = tid; e < n_here * HD; e += FUSED_WG) {
                const int nl = e / HD;
                tile[nl * S + (e - nl * HD)] = gsrc[e];
            }
            __syncthreads();
        }

        // ---------------- phase R + K (wave 0) ----------------
        if (a.dbg && tile_id == blockIdx.x) stamp[2] = __builtin_readcyclecounter();
        if (r_wave) {
            // (re)load the model operand here so it is not live across the sampling phase
            float mA[KK][CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int col = ct * 4 + (lane & 3);
                const bool ok = col < O;
#pragma unroll
                for (int k = 0; k < O; ++k) mA[k][ct] = ok ? a.A[k * O + col] : 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) mA[O + j][ct] = ok ? a.B[j * O + col] : 0.f;
            }
            const bool live = lane < n_here;
            float obs[O];
#pragma unroll
            for (int k = 0; k < O; ++k) obs[k] = k < a.o ? a.obs0[k] : 0.f;
            const float* arow = tile + (live ? lane : 0) * S;
            float acc_cost = 0.f;
#pragma unroll 1
            for (int t = 0; t < H; ++t) {
                float act[D];
#pragma unroll
                for (int j = 0; j < D; ++j) act[j] = arow[t * D + j];
                f32x4 acc[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < O; ++k) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_4x4x1f32(mA[k][ct], obs[k], acc[ct], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < D; ++j) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f32_4x4x1f32(mA[O + j][ct], act[j], acc[ct], 0, 0, 0);
                }
                // cost of (o_t, a_t) on the VALU while the matrix pipe works
                float ctrl = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) ctrl = __builtin_fmaf(act[j], act[j], ctrl);
                float lin = 0.f, ang = 0.f;
#pragma unroll
                for (int k = 0; k < O; ++k) {
                    lin = (k == a.lin_idx) ? obs[k] : lin;
                    ang = (k == a.flip_idx) ? obs[k] : ang;
                }
                float c = 0.f;
                if (a.flip_idx >= 0) {
                    c += (ang > a.flip_th) ? a.flip_pen : 0.f;
                    c += (ang < -a.flip_th) ? a.flip_pen : 0.f;
                }
                c += a.ctrl_w * ctrl;
                c += a.lin_w * lin;
                if (t == 0 || a.cost_mode == 2)
                    acc_cost = c;
                else if (a.cost_mode == 0)
                    acc_cost += c;
                else
                    acc_cost = c < acc_cost ? c : acc_cost;
#pragma unroll
                for (int k = 0; k < O; ++k) obs[k] = act_fn(acc[k / 4][k % 4], std::integral_constant<int, KIND>{});
            }
            if (live) a.costs[n_base + lane] = acc_cost;
            if (a.dbg && tile_id == blockIdx.x) stamp[3] = __builtin_readcyclecounter();
            // phase K: this tile's 64 keys, then a bitonic merge with the running top-K
            unsigned long long key = KEY_SENTINEL;
            if (live && n_base + lane < a.n_cand) key = make_key(acc_cost, n_base + lane);
            key = wave_sort64(key, lane);
            if (tile_id != (int)blockIdx.x) {  // not the first pass of this workgroup
                const unsigned long long top = __shfl(key, lane - a.K, 64);
                unsigned long long k2 = KEY_SENTINEL;
                if (lane < a.K)
                    k2 = run_key;
                else if (lane < 2 * a.K)
                    k2 = top;
                key = wave_sort64(k2, lane);
            }
            run_key = key;
        }
        __syncthreads();  // the tile is rewritten by the next pass
        if (a.dbg && tile_id == blockIdx.x) stamp[4] = __builtin_readcyclecounter();
    }
    if (a.dbg && tid == 0) {
        stamp[5] = __builtin_readcyclecounter();
        for (int i = 0; i < 6; ++i) a.dbg[(size_t)blockIdx.x * 8 + i] = stamp[i];
    }
    if (tid < a.K) {
        a.part_c[(size_t)blockIdx.x * a.K + tid] = key_cost(run_key);
        a.part_i[(size_t)blockIdx.x * a.K + tid] = key_idx(run_key);
    }
}

// -------------------------------------------------------------------------------------------------
constexpr int MERGE_WG = 1024;

__global__ __launch_bounds__(MERGE_WG) void merge_single_kernel(MergeSingleArgs a) {
    __shared__ unsigned long long red[MERGE_WG / 64];
    __shared__ unsigned long long sel[64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* new_mean = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = a.h * a.d;
    // thread t owns candidate list t (sorted) -- or kept elite t - n_lists (a one-entry list)
    int head = 0, len = 0;
    const float* lc = nullptr;
    const int* li = nullptr;
    float kept_c = 0.f;
    int kept_i = 0;
    if (tid < a.n_lists) {
        lc = a.part_c + (size_t)tid * a.K;
        li = a.part_i + (size_t)tid * a.K;
        len = a.K;
    } else if (tid < a.n_lists + a.n_keep) {
        kept_c = a.elites_cost_cur[tid - a.n_lists];
        kept_i = a.n_global + (tid - a.n_lists);
        len = -1;  // single pseudo-entry
    }
    auto head_key = [&]() -> unsigned long long {
        if (len > 0 && head < len) {
            const int i = li[head];
            return i == INT_MAX ? KEY_SENTINEL : make_key(lc[head], i);
        }
        if (len == -1 && head == 0) return make_key(kept_c, kept_i);
        return KEY_SENTINEL;
    };
    unsigned long long mine = head_key();
    for (int r = 0; r < a.K; ++r) {
        unsigned long long k = mine;
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const unsigned long long o = __shfl_xor(k, s, 64);
            k = o < k ? o : k;
        }
        if (lane == 0) red[wave] = k;
        __syncthreads();
        unsigned long long best = red[0];
#pragma unroll
        for (int w = 1; w < MERGE_WG / 64; ++w) best = red[w] < best ? red[w] : best;
        if (mine == best && best != KEY_SENTINEL) {  // unique: keys embed the trajectory index
            ++head;
            mine = head_key();
        }
        if (tid == 0) sel[r] = best;
        __syncthreads();
    }
    // gather + refit (icem.py:201-211)
    auto src_row = [&](int r) -> const float* {
        const int g = key_idx(sel[r]);
        return g < a.n_pool ? a.actions + (size_t)g * hd : a.elites_cur + (size_t)(g - a.n_global) * hd;
    };
    const float invK = 1.f / (float)a.K;
    for (int e = tid; e < hd; e += MERGE_WG) {
        float s = 0.f;
        for (int r = 0; r < a.K; ++r) {
            const float x = src_row(r)[e];
            a.elites_next[(size_t)r * hd + e] = x;
            s += x;
        }
        const float m = s / (float)a.K;
        float v = 0.f;
        for (int r = 0; r < a.K; ++r) {
            const float dx = src_row(r)[e] - m;
            v = __builtin_fmaf(dx, dx, v);
        }
        const float sd = sqrtf(v / (float)a.K);
        const float nm = (1.f - a.alpha) * m + a.alpha * a.mean[e];
        const float ns = (1.f - a.alpha) * sd + a.alpha * a.std[e];
        if (!a.last) {
            a.mean[e] = nm;
            a.std[e] = ns;
        } else {
            new_mean[e] = nm;
        }
    }
    (void)invK;
    if (tid < a.K) a.elites_cost_next[tid] = key_cost(sel[tid]);
    if (a.last) {
        __syncthreads();
        for (int e = tid; e < hd; e += MERGE_WG) {
            const int j = e % a.d;
            a.mean[e] = (e + a.d < hd) ? new_mean[e + a.d] : new_mean[e];
            a.std[e] = (a.high[j] - a.low[j]) / 2.f * a.init_std;
        }
        if (tid < a.d) a.executed[tid] = src_row(0)[tid];
        if (tid == 0) a.best_cost[0] = key_cost(sel[0]);
    }
}

template <int H, int D, int O>
int launch_hdo(const FusedArgs& a, int kind, int rounds, int grid, hipStream_t st) {
    if (kind == 1) {
        if (rounds == 7)
            hipLaunchKernelGGL((fused_iter_kernel<H, D, O, 1, 7>), dim3(grid), dim3(FUSED_WG), 0, st, a);
        else
            hipLaunchKernelGGL((fused_iter_kernel<H, D, O, 1, 10>), dim3(grid), dim3(FUSED_WG), 0, st, a);
    } else {
        if (rounds == 7)
            hipLaunchKernelGGL((fused_iter_kernel<H, D, O, 0, 7>), dim3(grid), dim3(FUSED_WG), 0, st, a);
        else
            hipLaunchKernelGGL((fused_iter_kernel<H, D, O, 0, 10>), dim3(grid), dim3(FUSED_WG), 0, st, a);
    }
    return 0;
}

}  // namespace

// The compiled shape list (H, D, O): the benchmark / golden shapes.  Anything else runs on the
// unfused generic kernels.
#define ICEM_FUSED_SHAPES(X) X(30, 6, 17) X(30, 6, 18) X(12, 6, 17) X(13, 4, 17) X(10, 3, 17) X(30, 17, 24)

bool fused_supported(int O, int d, int h, int K) {
    if (K > 32) return false;
#define X(HH, DD, OO) \
    if (h == HH && d == DD && O == OO) return true;
    ICEM_FUSED_SHAPES(X)
#undef X
    return false;
}

int fused_tile_traj(int h, int d) { return tile_traj(h, d); }
int fused_tile_stride(int h, int d) { return tile_stride(h, d); }

int launch_fused_iter(const FusedArgs& a, int O, int kind, int rounds, int grid, hipStream_t st) {
#define X(HH, DD, OO) \
    if (a.h == HH && a.d == DD && O == OO) return launch_hdo<HH, DD, OO>(a, kind, rounds, grid, st);
    ICEM_FUSED_SHAPES(X)
#undef X
    return 1;
}

void launch_merge_single(const MergeSingleArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(merge_single_kernel, dim3(1), dim3(MERGE_WG), (size_t)a.h * a.d * sizeof(float), st, a);
}

}  // namespace icem
