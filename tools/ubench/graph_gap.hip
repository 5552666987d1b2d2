// Dependent-kernel gap: plain stream launches vs the same chain captured in a hipGraph.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(128) void tiny(float* sink) { sink[blockIdx.x * 128 + threadIdx.x] += 1.f; }
int main() {
    float* sink; (void)hipMalloc(&sink, 256 * 128 * 4);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) tiny<<<256, 128, 0, st>>>(sink);
    (void)hipStreamSynchronize(st);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < 200; ++i) tiny<<<256, 128, 0, st>>>(sink);
    (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("stream: %.2f us per dependent kernel\n", ms * 1e3 / 200);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 16; ++i) tiny<<<256, 128, 0, st>>>(sink);
    (void)hipStreamEndCapture(st, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipStreamSynchronize(st);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(ge, st);
    (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("graph of 16 kernels: %.2f us per graph launch -> %.2f us per kernel\n", ms * 1e3 / 20, ms * 1e3 / 20 / 16);
    return 0;
}
