// Test program (tests/test_gpu_select_units.py compiles and runs it on the GPU box): the one-wave selections of the merge
// prologues -- merge_select<12> (registers), merge_select_stream (O(1) registers), merge_select_shallow<3> (the noise-ahead
// launch's) -- on crafted candidate lists: they must return the same K keys, ascending.  Cases: random lists; the K best
// clustered in ONE list / in a few lists (survivors deeper than the depths held in registers); many ties at the threshold
// (<= 64 and > 64 survivors); fewer than K finite keys; kept elites that win / lose; short launches (fewer than 64 lists).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
#include "fused_dev.h"
using namespace icem;

template <int WHICH>
__global__ __launch_bounds__(64) void select_kernel(MergeSingleArgs a, unsigned long long* out) {
    __shared__ unsigned long long cand[64];
    __shared__ unsigned long long sel[64];
    const int lane = threadIdx.x;
    sel[lane] = 0ull;
    __syncthreads();
    if (WHICH == 0) merge_select<12>(a, lane, cand, sel);
    if (WHICH == 1) merge_select<12, true>(a, lane, cand, sel);
    if (WHICH == 2) merge_select_stream(a, lane, cand, sel);
    if (WHICH == 3) merge_select_shallow<3>(a, lane, cand, sel);
    __syncthreads();
    if (lane < a.K) out[lane] = sel[lane];
}

static unsigned long long host_key(float c, int idx) {
    if (c != c) c = INFINITY;
    c = c + 0.0f;
    unsigned u;
    __builtin_memcpy(&u, &c, 4);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned)idx;
}

int main() {
    std::mt19937 rng(7);
    const int K = 10;
    int bad = 0, cases = 0;
    unsigned long long *d_part, *d_out;
    float* d_keep;
    (void)hipMalloc(&d_part, sizeof(unsigned long long) * K * 256);
    (void)hipMalloc(&d_out, sizeof(unsigned long long) * 64);
    (void)hipMalloc(&d_keep, sizeof(float) * 64);
    for (int c = 0; c < 400; ++c) {
        const int kind = c % 8;
        const int n_lists = (kind == 7) ? 1 + (int)(rng() % 63) : (c % 3 == 0 ? 256 : 64 + (int)(rng() % 193));
        const int rows_per_list = 64;
        const int n_pool = n_lists * rows_per_list;
        const int n_keep = (c % 2) ? 3 : 0;
        std::vector<float> cost(n_pool);
        std::uniform_real_distribution<float> U(0.f, 1.f);
        for (auto& x : cost) x = 10.f + U(rng);
        if (kind == 1) {   // the K best clustered in one list
            const int l = rng() % n_lists;
            for (int r = 0; r < K + 2; ++r) cost[l * rows_per_list + r] = 1.f + 0.01f * r;
        } else if (kind == 2) {   // ... in three lists, deep
            for (int j = 0; j < 3; ++j) {
                const int l = rng() % n_lists;
                for (int r = 0; r < 5; ++r) cost[l * rows_per_list + 7 * r] = 1.f + U(rng);
            }
        } else if (kind == 3) {   // 40 ties at the best cost
            for (int r = 0; r < 40; ++r) cost[rng() % n_pool] = 2.f;
        } else if (kind == 4) {   // everything tied
            for (auto& x : cost) x = 3.f;
        } else if (kind == 5) {   // all but six rows NaN / inf
            for (auto& x : cost) x = (rng() % 2) ? NAN : INFINITY;
            for (int r = 0; r < 6; ++r) cost[rng() % n_pool] = U(rng);
        } else if (kind == 6) {   // 200 ties in the first list's neighbourhood + a few better
            for (int r = 0; r < 200 && r < n_pool; ++r) cost[r] = 2.f;
            for (int r = 0; r < 4; ++r) cost[rng() % n_pool] = 1.f;
        }
        // per-list sorted top-K keys, [K][n_lists]
        std::vector<unsigned long long> part((size_t)K * n_lists);
        for (int l = 0; l < n_lists; ++l) {
            std::vector<unsigned long long> ks(rows_per_list);
            for (int r = 0; r < rows_per_list; ++r) ks[r] = host_key(cost[l * rows_per_list + r], l * rows_per_list + r);
            std::sort(ks.begin(), ks.end());
            for (int i = 0; i < K; ++i) part[(size_t)i * n_lists + l] = ks[i];
        }
        float keep[64] = {0};
        for (int e = 0; e < n_keep; ++e) keep[e] = (e == 0 && c % 4 == 1) ? 0.5f : 10.5f + e;
        // reference: the K smallest of all keys + kept
        std::vector<unsigned long long> all;
        for (int i = 0; i < n_pool; ++i) all.push_back(host_key(cost[i], i));
        for (int e = 0; e < n_keep; ++e) all.push_back(host_key(keep[e], n_pool + e));
        std::sort(all.begin(), all.end());
        (void)hipMemcpy(d_part, part.data(), part.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemcpy(d_keep, keep, sizeof(keep), hipMemcpyHostToDevice);
        MergeSingleArgs a{};
        a.n_lists = n_lists;
        a.n_keep = n_keep;
        a.keep_base = -1;
        a.n_pool = n_pool;
        a.n_global = n_pool;
        a.K = K;
        a.h = 30;
        a.d = 6;
        a.part_k = d_part;
        a.elites_cost_cur = d_keep;
        for (int which = 0; which < 4; ++which) {
            (void)hipMemset(d_out, 0, 64 * 8);
            if (which == 0) select_kernel<0><<<1, 64>>>(a, d_out);
            if (which == 1) select_kernel<1><<<1, 64>>>(a, d_out);
            if (which == 2) select_kernel<2><<<1, 64>>>(a, d_out);
            if (which == 3) select_kernel<3><<<1, 64>>>(a, d_out);
            unsigned long long got[64];
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
            (void)hipMemcpy(got, d_out, sizeof(got), hipMemcpyDeviceToHost);
            bool ok = true;
            for (int i = 0; i < K; ++i) ok = ok && got[i] == all[i];
            ++cases;
            if (!ok) {
                ++bad;
                if (bad <= 5) printf("mismatch: case %d kind %d form %d n_lists %d n_keep %d\n", c, kind, which, n_lists, n_keep);
            }
        }
    }
    printf("select_equiv: %d cases, %d mismatching\n", cases, bad);
    return bad ? 1 : 0;
}
