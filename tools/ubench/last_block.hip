// Cost of the "last workgroup done does the epilogue" pattern on 8 XCDs: every workgroup writes a 46 KB tile and K
// candidate keys, releases (agent-scope fence + atomic ticket); the workgroup that draws the last ticket acquires
// and reads all the keys.  Compared with the same kernel without the ticket followed by a second tiny kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TILE = 46 * 1024 / 4;
__global__ __launch_bounds__(192) void work(float* tiles, unsigned long long* keys, unsigned* ticket, unsigned long long* out,
                                            int mode) {
    float* t = tiles + (size_t)blockIdx.x * TILE;
    for (int e = threadIdx.x; e < TILE; e += 192) t[e] = e * 0.5f + blockIdx.x;
    if (threadIdx.x < 10) keys[threadIdx.x * gridDim.x + blockIdx.x] = ((unsigned long long)blockIdx.x << 32) | threadIdx.x;
    if (mode == 0) return;
    __shared__ unsigned last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = tk == gridDim.x - 1;
        if (last) *ticket = 0;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    unsigned long long acc = 0;
    for (int e = threadIdx.x; e < 10 * (int)gridDim.x; e += 192) acc += __builtin_nontemporal_load(keys + e);
    atomicAdd(out, acc);
}
__global__ __launch_bounds__(192) void epilogue(const unsigned long long* keys, unsigned long long* out, int n) {
    unsigned long long acc = 0;
    for (int e = threadIdx.x; e < n; e += 192) acc += keys[e];
    atomicAdd(out, acc);
}
int main() {
    float* tiles; unsigned long long *keys, *out; unsigned* ticket;
    (void)hipMalloc(&tiles, (size_t)256 * TILE * 4); (void)hipMalloc(&keys, 256 * 10 * 8); (void)hipMalloc(&out, 8);
    (void)hipMalloc(&ticket, 4); (void)hipMemset(ticket, 0, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode : {0, 1, 0, 1}) {
        (void)hipMemset(out, 0, 8);
        for (int i = 0; i < 20; ++i) { work<<<256, 192>>>(tiles, keys, ticket, out, mode); if (!mode) epilogue<<<1, 192>>>(keys, out, 2560); }
        (void)hipDeviceSynchronize();
        (void)hipMemset(out, 0, 8);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) { work<<<256, 192>>>(tiles, keys, ticket, out, mode); if (!mode) epilogue<<<1, 192>>>(keys, out, 2560); }
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("%s: %.2f us per iteration (checksum %llu)\n", mode ? "last-workgroup epilogue (1 launch) " : "separate epilogue kernel (2 launches)", ms * 1e3 / 200, h);
    }
    return 0;
}
