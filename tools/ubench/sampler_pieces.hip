// What a (trajectory, dim) row of the colored-noise sampler costs, piece by piece (EXPERIMENTS R5.3): 512-thread workgroups, one
// row per thread as noise_rows_kernel / the noise role of iter_ahead_kernel run it, n = 65 536 trajectories x 6 dims.
//   0  row_normals only (Philox seed + 30 xoshiro words + 15 Box-Muller pairs), result reduced to one store per thread
//   1  + row_synth with the table as wave-uniform scalar operands (what ships), one store per thread
//   2  the same + the LDS tile and the coalesced copy-out (= noise_rows_kernel)
//   3  row_synth alone on constant draws (no generator)
//   4  row_synth with the table in LDS (ds_read broadcast) instead of scalar loads
// build on the GPU box: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I icem_amd/csrc tools/ubench/sampler_pieces.hip -o /tmp/sp && /tmp/sp
#include "fused_dev.h"
#include <cstdio>
#include <vector>
using namespace icem;
constexpr int H = 30, D = 6, HD = H * D, NT = 512, TPW = NT / D;
template <int MODE> __global__ __launch_bounds__(NT) void k(const float* W, float* out, int n, unsigned off) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, n_base = blockIdx.x * TPW, nl = tid / D, j = tid - nl * D;
    const bool has = tid < TPW * D && n_base + nl < n;
    float g[HMAX];
    float acc = 0.f;
    if (MODE == 4) {
        for (int e = tid; e < 16 * HMAX; e += NT) smem[e] = W[e];
        __syncthreads();
    }
    if (has) {
        if (MODE != 3) row_normals<H, 10>((unsigned)(n_base + nl), (unsigned)j, off, 0u, 1234u, 0u, g);
        else
#pragma unroll
            for (int m = 0; m < HMAX; ++m) g[m] = 0.25f * m + tid * 1e-3f;
        if (MODE == 0) {
#pragma unroll
            for (int m = 0; m < H; ++m) acc += g[m];
        } else if (MODE == 1 || MODE == 3) {
            row_synth<H>(W, g, [&](int t, float y) { acc += y; }, false);
        } else if (MODE == 4) {
            row_synth<H>(smem, g, [&](int t, float y) { acc += y; }, false);
        } else {
            float* trow = smem + nl * HD + j;
            row_synth<H>(W, g, [&](int t, float y) { trow[t * D] = y; }, false);
        }
    }
    if (MODE == 2) {
        __syncthreads();
        const int n_here = cmin(TPW, n - n_base);
        const float4* t4 = reinterpret_cast<const float4*>(smem);
        float4* g4 = reinterpret_cast<float4*>(out + (size_t)n_base * HD);
        for (int e = tid; e < n_here * HD / 4; e += NT) g4[e] = t4[e];
    } else if (has) {
        out[(size_t)(n_base + nl) * D + j] = acc;
    }
}
template <int MODE> void run(const char* name, const float* W, float* out, int n) {
    const int grid = (n + TPW - 1) / TPW;
    const size_t lds = MODE == 2 ? TPW * HD * 4 : (MODE == 4 ? 16 * HMAX * 4 : 0);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int r = 0; r < 3; ++r) k<MODE><<<grid, NT, lds>>>(W, out, n, r);
    (void)hipEventRecord(a);
    const int reps = 50;
    for (int r = 0; r < reps; ++r) k<MODE><<<grid, NT, lds>>>(W, out, n, r);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / reps, wave_rows = (double)n * D / 64.0 / 1024.0;
    printf("%-60s %7.2f us per launch (back to back) = %5.2f us per wave-row and SIMD\n", name, us, us / wave_rows);
}
int main() {
    const int n = 65536;
    std::vector<float> Wh(H * HMAX);
    for (size_t i = 0; i < Wh.size(); ++i) Wh[i] = 0.01f * (float)(i % 37) - 0.1f;
    float *W, *out; (void)hipMalloc(&W, Wh.size() * 4); (void)hipMalloc(&out, (size_t)n * HD * 4 + 4096);
    (void)hipMemcpy(W, Wh.data(), Wh.size() * 4, hipMemcpyHostToDevice);
    run<0>("0 generator + Box-Muller", W, out, n);
    run<3>("3 synthesis alone (scalar table)", W, out, n);
    run<4>("4 generator + Box-Muller + synthesis (LDS table)", W, out, n);
    run<1>("1 generator + Box-Muller + synthesis (scalar table)", W, out, n);
    run<2>("2 ... + LDS tile + coalesced copy-out (noise_rows_kernel)", W, out, n);
    return 0;
}
