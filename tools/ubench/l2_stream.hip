// How fast can ONE CU pull a read-only buffer that sits in L2?  (The learned-dynamics rollout streams 0.65 MB of weights
// per model step through every CU that hosts a recurrence workgroup; this is the floor of that design.)
// grid workgroups of 512 threads; every wave reads its own 1/8 of `bytes` with 16-byte loads per lane, DEPTH loads in
// flight, `steps` times over; one workgroup per CU when grid <= 256.  Also: 8-byte loads, and the same with all
// workgroups reading DIFFERENT copies (no sharing of L2 lines between CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4i* gp4;
template <int DEPTH>
__global__ __launch_bounds__(512) void stream(const v4i* buf, size_t bytes, int steps, int copies, int* sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t per_wave = bytes / 8 / 16;   // v4i elements per wave
    gp4 base = (gp4)buf + (size_t)(blockIdx.x % copies) * (bytes / 16) + (size_t)w * per_wave + lane;
    v4i acc = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
        gp4 p = base;
        asm volatile("" : "+v"(p));
        for (size_t i = 0; i < per_wave; i += 64 * DEPTH) {
            v4i r[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) r[d] = p[i + 64 * d];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= r[d];
        }
    }
    if (acc[0] == 0x12345678) sink[0] = acc[1] + acc[2] + acc[3];
}
int main() {
    const size_t bytes = 8 * 64 * 16 * 80;   // 655 360 bytes: 80 loads per lane per step
    const int max_copies = 64;
    v4i* buf; int* sink;
    (void)hipMalloc(&buf, bytes * max_copies); (void)hipMemset(buf, 1, bytes * max_copies); (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int steps = 200;
    for (int copies : {1, 64}) {
        for (int grid : {8, 32, 64, 128, 256}) {
            for (int depth : {8, 20, 40}) {
                for (int rep = 0; rep < 2; ++rep) {
                    (void)hipEventRecord(e0);
                    if (depth == 8) stream<8><<<grid, 512>>>(buf, bytes, steps, copies, sink);
                    if (depth == 20) stream<20><<<grid, 512>>>(buf, bytes, steps, copies, sink);
                    if (depth == 40) stream<40><<<grid, 512>>>(buf, bytes, steps, copies, sink);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                }
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                const double us_step = ms * 1e3 / steps;
                printf("copies %2d grid %3d depth %2d: %.2f us per 655 KB step per CU = %.1f GB/s per CU (%.1f B/clk at 2.4 GHz), %.2f TB/s total\n", copies,
                       grid, depth, us_step, bytes / us_step / 1e3, bytes / us_step / 1e3 / 2.4, bytes / us_step / 1e6 * grid);
            }
        }
    }
    return 0;
}
