// Accuracy of a K = 128 dot product per output element of a 16 x 16 tile, against float64 on the host:
//   f32   : 32 x v_mfma_f32_16x16x4_f32
//   x3    : bf16 head / remainder splits, 3 products per 32 k  (x_h w_h + x_h w_l + x_l w_h)
//   x6    : three bf16 pieces per operand (exact: 8 + 8 + 8 bits), 6 products per 32 k
// prints max |err| / max |exact| and the rms ratio.   hipcc --offload-arch=gfx950 -O3 -o split_mfma split_mfma.hip && ./split_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short bf(float a) {
    const f32x2 v = {a, 0.f};
    return (unsigned short)(__builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf2)) & 0xffffu);
}
__device__ __forceinline__ float up(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// A [16][128] row-major (rows = i), B [16][128] row-major (rows = j): D[i][j] = sum_k A[i][k] B[j][k]
__global__ void k_all(const float* A, const float* B, float* Df, float* D3, float* D6) {
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    f32x4 af = {0, 0, 0, 0}, a3 = af, a6 = af;
    for (int ks = 0; ks < 32; ++ks) af = __builtin_amdgcn_mfma_f32_16x16x4f32(A[li * 128 + ks * 4 + lk], B[li * 128 + ks * 4 + lk], af, 0, 0, 0);
    for (int kc = 0; kc < 4; ++kc) {
        bf16x8 a1, a2, a3p, b1, b2, b3p;
        for (int t = 0; t < 8; ++t) {
            const float x = A[li * 128 + kc * 32 + lk * 8 + t], y = B[li * 128 + kc * 32 + lk * 8 + t];
            const unsigned short x1 = bf(x), x2 = bf(x - up(x1)), x3 = bf((x - up(x1)) - up(x2));
            const unsigned short y1 = bf(y), y2 = bf(y - up(y1)), y3 = bf((y - up(y1)) - up(y2));
            a1[t] = x1; a2[t] = x2; a3p[t] = x3; b1[t] = y1; b2[t] = y2; b3p[t] = y3;
        }
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, a3, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3p, b1, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3p, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, a6, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, a6, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) {
        const int i = 4 * lk + r, j = li;
        Df[i * 16 + j] = af[r]; D3[i * 16 + j] = a3[r]; D6[i * 16 + j] = a6[r];
    }
}

int main() {
    std::vector<float> A(16 * 128), B(16 * 128);
    srand(3);
    double worst[3] = {0, 0, 0}, rms[3] = {0, 0, 0}, scale = 0;
    float *dA, *dB, *dF, *d3, *d6;
    hipMalloc(&dA, 8192); hipMalloc(&dB, 8192); hipMalloc(&dF, 1024); hipMalloc(&d3, 1024); hipMalloc(&d6, 1024);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& v : A) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0);
        for (auto& v : B) v = (float)((rand() / (double)RAND_MAX - 0.5) * 2.0) * (trial % 2 ? 1.f : 37.5f);
        hipMemcpy(dA, A.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 8192, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_all, dim3(1), dim3(64), 0, 0, dA, dB, dF, d3, d6);
        float F[256], T3[256], T6[256];
        hipMemcpy(F, dF, 1024, hipMemcpyDeviceToHost); hipMemcpy(T3, d3, 1024, hipMemcpyDeviceToHost); hipMemcpy(T6, d6, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double e = 0, m = 0;
                for (int k = 0; k < 128; ++k) { e += (double)A[i * 128 + k] * B[j * 128 + k]; m += fabs((double)A[i * 128 + k] * B[j * 128 + k]); }
                const float* R[3] = {F, T3, T6};
                for (int q = 0; q < 3; ++q) {
                    const double err = fabs(R[q][i * 16 + j] - e) / m;              // relative to sum |terms|
                    worst[q] = fmax(worst[q], err); rms[q] += err * err;
                }
                scale += 1;
            }
    }
    const char* nm[3] = {"f32 16x16x4 ", "bf16 x3     ", "bf16 x6     "};
    for (int q = 0; q < 3; ++q) printf("%s  max err / sum|terms| %.3e   rms %.3e\n", nm[q], worst[q], sqrt(rms[q] / scale));
    return 0;
}
