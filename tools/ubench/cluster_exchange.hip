// What would it cost to spread ONE 16-trajectory tile of the RSSM rollout over a cluster of C workgroups (each streaming
// 1/C of the weights), i.e. an activation exchange + cluster barrier per layer boundary?  C workgroups exchange 1.6 KB
// slices through global memory and meet at a counter, `rounds` times; cluster members are blockIdx b, b+8, b+16, ...
// (the same XCD if workgroups are dealt round-robin to the XCDs -- the XCC_ID register is read to check).
//   mode 0: plain stores / loads + agent-scope fences around the counter (the memory model's way, any placement)
//   mode 1: written-through stores + agent-scope (cache-bypassing) loads, no fences (any placement)
//   mode 2: plain stores, workgroup-scope atomic loads, no fences (only valid inside one XCD's L2 -- if they bypass L1)
//   mode 3: plain stores, nontemporal loads, no fences;  mode 4: plain stores, plain loads, no fences (expected WRONG: stale L1)
// every value read is checked
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int SL = 400;  // floats per slice
template <int MODE>
__global__ __launch_bounds__(512) void exch(float* scratch, unsigned* cnt, int C, int rounds, long long* t, int* xcc, float* sink, unsigned* bad) {
    const int b = blockIdx.x, cl = b % 8 + 8 * (b / (8 * C)), me = (b / 8) % C;
    unsigned* my_cnt = cnt + cl * 32;
    float* base = scratch + (size_t)cl * 2 * C * SL;
    if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID[3:0]
    float acc = threadIdx.x;
    __syncthreads();
    const long long w0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        float* mine = base + ((size_t)(r & 1) * C + me) * SL;
        if (threadIdx.x < SL) {
            const float v = (float)(r * 131 + me * 17) + (float)threadIdx.x * 0.25f;   // checkable
            if (MODE == 1) __hip_atomic_store(mine + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else mine[threadIdx.x] = v;
        }
        if (MODE == 0) __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(my_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * C;
            while (__hip_atomic_load(my_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (MODE == 0) __threadfence();
        float s = 0.f;
        for (int m = 0; m < C; ++m) {
            if (m == me) continue;
            const float* other = base + ((size_t)(r & 1) * C + m) * SL;
            if (threadIdx.x < SL) {
                float got;
                if (MODE == 1) got = __hip_atomic_load(other + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 2) got = __hip_atomic_load(other + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 3) got = __builtin_nontemporal_load(other + threadIdx.x);
                else got = other[threadIdx.x];
                if (got != (float)(r * 131 + m * 17) + (float)threadIdx.x * 0.25f) atomicAdd(bad, 1u);
                s += got;
            }
        }
        acc = acc * 0.5f + s * 1e-3f;
    }
    const long long w1 = wall_clock64();
    if (threadIdx.x == 0) t[b] = w1 - w0;
    sink[b * 512 + threadIdx.x] = acc;
}
int main() {
    float *scratch, *sink; unsigned *cnt, *bad; long long* t; int* xcc;
    (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&scratch, 256 * 2 * SL * 4 * 2); (void)hipMalloc(&cnt, 64 * 32 * 4 * 4); (void)hipMalloc(&t, 256 * 8);
    (void)hipMalloc(&xcc, 256 * 4); (void)hipMalloc(&sink, 256 * 512 * 4);
    const int rounds = 2000;
    for (int C : {2, 4}) {
        for (int mode = 1; mode < 5; ++mode) {
            (void)hipMemset(cnt, 0, 64 * 32 * 4 * 4); (void)hipMemset(bad, 0, 4);
            const int grid = 64 * C;  // 64 tiles (N = 1024) spread over C workgroups each
            if (mode == 1) exch<1><<<grid, 512>>>(scratch, cnt, C, rounds, t, xcc, sink, bad);
            if (mode == 2) exch<2><<<grid, 512>>>(scratch, cnt, C, rounds, t, xcc, sink, bad);
            if (mode == 3) exch<3><<<grid, 512>>>(scratch, cnt, C, rounds, t, xcc, sink, bad);
            if (mode == 4) exch<4><<<grid, 512>>>(scratch, cnt, C, rounds, t, xcc, sink, bad);
            (void)hipDeviceSynchronize();
            std::vector<long long> h(grid); std::vector<int> x(grid);
            (void)hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(x.data(), xcc, grid * 4, hipMemcpyDeviceToHost);
            long long mx = 0; int same = 0, clusters = 0;
            for (int b = 0; b < grid; ++b) mx = h[b] > mx ? h[b] : mx;
            for (int b = 0; b < grid; ++b) if ((b / 8) % C == 0) { ++clusters; bool ok = true; for (int m = 1; m < C; ++m) ok = ok && x[b + 8 * m] == x[b]; same += ok; }
            unsigned hb = 0; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("C=%d mode %d: %.2f us per exchange (slowest workgroup), %u wrong values; clusters on one XCD: %d of %d; xcc of wg 0..9:", C, mode,
                   mx / 100.0 / rounds, hb, same, clusters);
            for (int b = 0; b < 10; ++b) printf(" %d", x[b]);
            printf("\n");
        }
    }
    return 0;
}
