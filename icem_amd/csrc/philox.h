// Philox4x32 counter RNG (Salmon et al., SC'11) + Box-Muller for gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icem {

struct U4 {
    uint32_t x, y, z, w;
};

// One v_mad_u64_u32 per 32x32->64 product; R rounds (10 = Random123 default, 7 = the
// smallest Crush-resistant count).
template <int R>
__device__ __forceinline__ U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                         uint32_t k1) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// Per-row stream: ONE Philox4x32-R call keyed by (seed, call offset, trajectory, action dim) seeds a
// xoshiro128++ state (Blackman & Vigna); the h words of the row come from that generator: 10
// full-rate integer ops per word instead of 5 quarter-rate 32x32->64 multiplies.  Counter based where
// it matters (any row of any call is reproducible on any shard), cheap where it is hot.
struct Xoshiro128pp {
    uint32_t s0, s1, s2, s3;
    __device__ __forceinline__ uint32_t next() {
        const uint32_t sum = s0 + s3;
        const uint32_t result = ((sum << 7) | (sum >> 25)) + s0;
        const uint32_t t = s1 << 9;
        s2 ^= s0;
        s3 ^= s1;
        s1 ^= s2;
        s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 11) | (s3 >> 21);
        return result;
    }
};

template <int R>
__device__ __forceinline__ Xoshiro128pp row_stream(uint32_t traj, uint32_t dim, uint32_t off_lo, uint32_t off_hi,
                                                   uint32_t seed_lo, uint32_t seed_hi) {
    const U4 r = philox4x32<R>(traj, dim << 16, off_lo, off_hi, seed_lo, seed_hi);
    return Xoshiro128pp{r.x, r.y, r.z, r.w};
}

// Two normals from two words.  f32: hardware log2 / sqrt / sin / cos (the angle is fed in
// revolutions, which is what v_sin_f32 / v_cos_f32 take).  f64: libm-grade.
__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, float& g0, float& g1) {
    const float u1 = __builtin_fmaf((float)xa, 0x1p-32f, 0x1p-33f);
    const float v = (float)xb * 0x1p-32f;
    // -2 ln(u1) = -2 ln2 * log2(u1)
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // raw v_sqrt_f32 (1 ulp)
    g0 = r * __builtin_amdgcn_cosf(v);
    g1 = r * __builtin_amdgcn_sinf(v);
}

__device__ __forceinline__ void box_muller(uint32_t xa, uint32_t xb, double& g0, double& g1) {
    const double u1 = ((double)xa + 0.5) * 0x1p-32;
    const double v = (double)xb * 0x1p-32;
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(6.283185307179586476925286766559 * v, &s, &c);
    g0 = r * c;
    g1 = r * s;
}

}  // namespace icem
