// Latency of a software grid barrier (one resident workgroup per CU, agent-scope atomics through L2/fabric) against
// the gap between two dependent kernel launches: is a persistent whole-MPC-step kernel worth it?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(128) void barriers(unsigned* cnt, int rounds, float* sink, long long* t) {
    long long w0 = wall_clock64();
    float acc = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        acc = acc * 1.0001f + 1.f;
        sink[blockIdx.x * 128 + threadIdx.x] = acc;  // some global traffic to release
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * gridDim.x;
            while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = wall_clock64() - w0;
}
__global__ __launch_bounds__(128) void tiny(float* sink) { sink[blockIdx.x * 128 + threadIdx.x] += 1.f; }
int main() {
    unsigned* cnt; float* sink; long long* t;
    (void)hipMalloc(&cnt, 4); (void)hipMalloc(&sink, 256 * 128 * 4 * 4); (void)hipMalloc(&t, 8);
    for (int grid : {64, 256}) {
        for (int rounds : {1, 101}) {
            (void)hipMemset(cnt, 0, 4);
            barriers<<<grid, 128>>>(cnt, rounds, sink, t); (void)hipDeviceSynchronize();
            long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
            printf("grid %3d rounds %3d: %.2f us total in-kernel -> %.2f us per barrier\n", grid, rounds, h / 100.0, h / 100.0 / rounds);
        }
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) tiny<<<256, 128>>>(sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) tiny<<<256, 128>>>(sink);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("200 dependent tiny kernels: %.2f us each\n", ms * 1e3 / 200);
    return 0;
}
