// Where should the ring that streams the learned-dynamics rollout's weight chunks live?  ("the streamed
// weight chunks through an LDS ring filled by direct-to-LDS loads ... or price it out".)  One workgroup of 8 waves per
// CU, 64 workgroups (the N = 1024 launch), every workgroup streams the SAME 40 chunks of 7 KB per "step" out of L2 (the
// recurrence's streamed set), wave w takes chunks w, w + 8, ...: five per step, each used as the B operands of seven
// v_mfma_f32_16x16x32_bf16 (what a chunk is in rssm_split_kernel).
//   mode R  the ring is SLOTS register slots per wave (28 VGPRs each): global_load_dwordx4 x 7 per chunk, used straight
//           from the registers -- rssm_split_kernel's form (ring of two);
//   mode L  the ring is SLOTS slots of LDS per wave: global_load_lds_dwordx4 x 7 per chunk (no destination registers),
//           counted s_waitcnt vmcnt, ds_read_b128 x 7, the same MFMAs.
// Prints us per step for SLOTS = 1..4 in both modes, and for mode L with a chunk used TWICE per fill (two tiles per workgroup).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_ring lds_ring.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int CHUNK_V4 = 7 * 64;          // 16-byte vectors per chunk (7 KB)
constexpr int NCHUNK = 40, PER_WAVE = 5;

template <int SLOTS>
__global__ __launch_bounds__(512) void ring_regs(const v4i* w, int steps, float* sink) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v4i slot[SLOTS][7];
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const v4i a = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    const int total = steps * PER_WAVE;
    auto src = [&](int i) { return w + (size_t)(wv + 8 * (i % PER_WAVE)) * CHUNK_V4 + lane; };
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int k = 0; k < 7; ++k) slot[s][k] = src(s)[64 * k];
    for (int i0 = 0; i0 < total; i0 += SLOTS) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int i = i0 + s;
#pragma unroll
            for (int k = 0; k < 7; ++k)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, slot[s][k]), acc, 0, 0, 0);
            const v4i* p = src(i + SLOTS);
            asm volatile("" : "+v"(p));
#pragma unroll
            for (int k = 0; k < 7; ++k) slot[s][k] = p[64 * k];
        }
    }
    if (acc[0] == 1234.5f) sink[0] = acc[1];
}

template <int SLOTS, int USES>
__global__ __launch_bounds__(512) void ring_lds(const v4i* w, int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v4i* ring = reinterpret_cast<v4i*>(smem) + (size_t)wv * SLOTS * CHUNK_V4;   // this wave's slots
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const v4i a = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};
    const int total = steps * PER_WAVE;
    auto fill = [&](int i, int s) {
        const v4i* g = w + (size_t)(wv + 8 * (i % PER_WAVE)) * CHUNK_V4 + lane;
#pragma unroll
        for (int k = 0; k < 7; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 64 * k),
                                             (__attribute__((address_space(3))) void*)(ring + s * CHUNK_V4 + 64 * k), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) fill(s, s);
    for (int i0 = 0; i0 < total; i0 += SLOTS) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            // the oldest fill has landed when at most 7 (SLOTS - 1) direct-to-LDS loads are still in flight
            if constexpr (SLOTS == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (SLOTS == 2) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            if constexpr (SLOTS == 3) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            if constexpr (SLOTS == 4) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
#pragma unroll
            for (int u = 0; u < USES; ++u) {
                v4i b[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) b[k] = ring[s * CHUNK_V4 + 64 * k + lane];
#pragma unroll
                for (int k = 0; k < 7; ++k)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b[k]), acc, 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's reads are done before it is refilled
            fill(i0 + s + SLOTS, s);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] == 1234.5f) sink[0] = acc[1];
}

int main() {
    v4i* w; float* sink;
    (void)hipMalloc(&w, (size_t)(NCHUNK + 8 * 8) * CHUNK_V4 * 16); (void)hipMemset(w, 0, (size_t)(NCHUNK + 8 * 8) * CHUNK_V4 * 16); (void)hipMalloc(&sink, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int steps = 240, grid = 64;
    auto time = [&](auto launch, const char* what) {
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-58s %.2f us per step (40 chunks = 280 KB per CU: %.0f GB/s per CU, %.1f B/clk at 2.4 GHz)%s\n", what, ms * 1e3 / steps,
               280.0 * 1024 / (ms * 1e3 / steps) / 1e3, 280.0 * 1024 / (ms * 1e3 / steps) / 1e3 / 2.4, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
    };
#define LDS_BYTES(S) ((size_t)8 * (S) * CHUNK_V4 * 16)
#define RUN_L(S, U)                                                                                                         \
    {                                                                                                                        \
        (void)hipFuncSetAttribute((const void*)ring_lds<S, U>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES(S)); \
        time([&] { ring_lds<S, U><<<grid, 512, LDS_BYTES(S)>>>(w, steps, sink); },                                           \
             "LDS ring, " #S " slot(s) per wave (" #U " use(s) per fill):");                                                 \
    }
    time([&] { ring_regs<1><<<grid, 512>>>(w, steps, sink); }, "register ring, 1 slot per wave:");
    time([&] { ring_regs<2><<<grid, 512>>>(w, steps, sink); }, "register ring, 2 slots per wave (as shipped):");
    time([&] { ring_regs<3><<<grid, 512>>>(w, steps, sink); }, "register ring, 3 slots per wave:");
    time([&] { ring_regs<4><<<grid, 512>>>(w, steps, sink); }, "register ring, 4 slots per wave:");
    RUN_L(1, 1) RUN_L(2, 1) RUN_L(1, 2) RUN_L(2, 2)
    printf("(three and four LDS slots per wave are 168 / 224 KB: more than a CU has)\n");
    return 0;
}
