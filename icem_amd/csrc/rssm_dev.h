// Device helpers shared by the recurrent state-space model kernels (icem_rssm.hip, icem_rssm_split.hip): operand types,
// bf16 packing, the A-operand requests, the 16x16x32 MFMA chains, the LDS bias table.
#pragma once
#include "icem_rssm.h"
#include "options.h"

namespace icem {
namespace rssm_dev {
using namespace rssm;
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int RS = 232;   // bf16 row stride of the 224-wide activation rows (208 used + zero padding to the K blocks)
constexpr int ZS = 72;    // ... of the [z (32) | a (32)] row
constexpr int HS = 212;   // f32 row stride of the recurrent state

__device__ __forceinline__ unsigned short to_bf16(float x) {   // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ v4s pack4(float a, float b, float c, float d) {
    v4s r;
    r[0] = (short)to_bf16(a); r[1] = (short)to_bf16(b); r[2] = (short)to_bf16(c); r[3] = (short)to_bf16(d);
    return r;
}
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return __builtin_fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }  // ~1e-6: far inside bf16
// One GRU output: h' = (1 - u) n + u h with r = sigma(ir + hr), u = sigma(iu + hu), n = tanh(in + r hn).  The fused
// multiply-adds are spelled out: left to the compiler's contraction the two kernels that share this line could round
// differently (an ulp of h', now and then a different bf16 operand), and they are tested to agree bit for bit.
__device__ __forceinline__ float gru_out(float ir, float iu, float in, float hr, float hu, float hn, float h) {
    const float rg = sigmoidf_(ir + hr);
    const float ug = sigmoidf_(iu + hu);
    const float ng = tanhf_(__builtin_fmaf(rg, hn, in));
    return __builtin_fmaf(ug, h, (1.f - ug) * ng);
}

// A wave owns output blocks w, w+WAVES, ... of a 13-block layer (waves without a last block redo block 12 and drop
// the result -- cheaper than a divergent trip count, the matrix pipe is not the limit).
// The weights come straight from L2, so what matters is how many loads are in flight: a layer first REQUESTS all of
// the wave's A-operand blocks (NOB x KB 16-byte loads per lane), then runs the MFMAs.
constexpr int WAVES = 8;                 // wavefronts per workgroup (two per SIMD: one's loads under the other's MFMAs)
constexpr int NOB = (13 + WAVES - 1) / WAVES;
constexpr int NTHR = 64 * WAVES;
__device__ __forceinline__ int own_block(int w, int i) { const int ob = w + WAVES * i; return ob < 13 ? ob : 12; }

// The parameter buffer as a GLOBAL-address-space pointer.  The kernels launder the pointer through an empty asm every
// model step (so that the optimizer does not carry parameters across steps in registers); a generic pointer coming out
// of that asm turns every weight load into a flat_load, which counts on lgkmcnt as well -- and then every wait for an
// LDS operand waits for all the weight requests in flight, i.e. nothing is ever requested ahead.
typedef const __attribute__((address_space(1))) unsigned short* gptr;
typedef const __attribute__((address_space(1))) v4i* gptr_v4i;
template <int KB>
__device__ __forceinline__ void request(gptr W, v4i (&A)[KB]) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) A[kb] = *reinterpret_cast<gptr_v4i>(W + (size_t)kb * BLK);
}
template <int KB>
__device__ __forceinline__ void request(const unsigned short* __restrict__ W, v4i (&A)[KB]) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) A[kb] = *reinterpret_cast<const v4i*>(W + (size_t)kb * BLK);
}
// X: the lane's 8 bf16 of k-block 0 (row j, column 8 * g)
template <int KB>
__device__ __forceinline__ v4f mma(const v4i (&A)[KB], const unsigned short* X, v4f acc) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, A[kb]),
                                                      __builtin_bit_cast(v8bf, *reinterpret_cast<const v4i*>(X + kb * 32)), acc, 0, 0, 0);
    return acc;
}
// The biases (2 128 floats) are copied to LDS once: read from global right in front of a layer's MFMAs they would put
// an L2 round trip on every layer's critical path.
constexpr int NBIAS = 16 * (HIDB + 3 * DETB + 3 * DETB + HIDB + STB + HIDB + HIDB + 1);
__host__ __device__ constexpr int bias_slot(size_t off) {
    return off == B1 ? 0 : off == BGI ? 16 * HIDB : off == BGH ? 16 * (HIDB + 3 * DETB) : off == B4 ? 16 * (HIDB + 6 * DETB)
         : off == B5 ? 16 * (2 * HIDB + 6 * DETB) : off == B6 ? 16 * (2 * HIDB + 6 * DETB + STB)
         : off == B7 ? 16 * (3 * HIDB + 6 * DETB + STB) : 16 * (4 * HIDB + 6 * DETB + STB);
}
constexpr int bias_len(size_t off) {
    return off == B1 || off == B4 || off == B6 || off == B7 ? 16 * HIDB : off == BGI || off == BGH ? 48 * DETB : off == B5 ? 16 * STB : 16;
}
__device__ __forceinline__ v4f bias4(const float* bs, size_t off, int idx) {
    return *reinterpret_cast<const v4f*>(bs + bias_slot(off) + idx);
}
__device__ __forceinline__ v4s relu_pack(v4f a) { return pack4(fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f)); }

// A 13-block layer in two halves so that its weight requests can be issued early (they depend on nothing but the
// parameters, so they may also sit in front of the barrier that ends the previous phase): req_own asks for the wave's
// NOB blocks; fin_dense runs the MFMAs for TT tiles of 16 trajectories that share them and stores relu(W X + b).
// X / Y: the lane's row pointers in tile 0, xts: X's element stride between tiles.
template <int KB>
__device__ __forceinline__ void req_own(gptr Plane, size_t woff, int w, v4i (&A)[NOB][KB]) {
    __builtin_amdgcn_sched_barrier(0);   // requests stay where they are written
#pragma unroll
    for (int i = 0; i < NOB; ++i) request<KB>(Plane + woff + (size_t)own_block(w, i) * KB * BLK, A[i]);
    __builtin_amdgcn_sched_barrier(0);
}
template <int KB, int TT>
__device__ __forceinline__ void fin_dense(const float* bs, size_t boff, const v4i (&A)[NOB][KB], const unsigned short* X, int xts,
                                          unsigned short* Y, int w, int g) {
#pragma unroll
    for (int i = 0; i < NOB; ++i) {
        const int ob = own_block(w, i);
        const v4f b = bias4(bs, boff, ob * 16 + 4 * g);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const v4f a = mma<KB>(A[i], X + tt * xts, b);
            if (w + WAVES * i < 13) *reinterpret_cast<v4s*>(Y + tt * 16 * RS + ob * 16) = relu_pack(a);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
}  // namespace rssm_dev
}  // namespace icem
