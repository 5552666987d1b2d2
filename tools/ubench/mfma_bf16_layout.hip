// Operand layout of v_mfma_f32_16x16x32_bf16 on gfx950: is element (row i, k) of A in lane i + 16*(k/8), slot k%8
// (and B likewise with the column), result D[i][j] in lane j + 16*(i/4), register i%4?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A [16][32], B [16][32] (row = output column j), D [16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    v8bf a, b;
    for (int v = 0; v < 8; ++v) { a[v] = (__bf16)A[r * 32 + 8 * g + v]; b[v] = (__bf16)B[r * 32 + 8 * g + v]; }
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int q = 0; q < 4; ++q) D[(4 * g + q) * 16 + r] = c[q];
}
int main() {
    float hA[512], hB[512], hD[256], *A, *B, *D;
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
    (void)hipMalloc(&A, 2048); (void)hipMalloc(&B, 2048); (void)hipMalloc(&D, 1024);
    (void)hipMemcpy(A, hA, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(A, B, D); (void)hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[j * 32 + kk];
        err = fmax(err, fabs(s - hD[i * 16 + j]));
    }
    printf("max |D - A B^T| with the assumed layout: %g\n", err);
    return 0;
}
