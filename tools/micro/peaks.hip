// Microbenchmarks behind the peaks quoted in DESIGN.md / bench.py (SURVEY 8d asks for both to be verified
// on the box): HBM stream copy / read / write bandwidth and the f32 / bf16 MFMA issue rate.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void copy_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_k(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = a[i]; s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ void write_k(float4* __restrict__ b, size_t n) {
    const float4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_f32_k(float* out, int iters) {
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void mfma_bf16_k(float* out, int iters) {
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(threadIdx.x * 1e-3f + k); b[k] = (__bf16)(1.0f + k * 0.01f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const size_t bytes = (size_t)4 << 30;                     // 4 GiB per buffer (far beyond the 256 MB MALL)
    const size_t n = bytes / 16;
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&out, 4096 * 256 * 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int grid : {2048, 8192, 32768}) {
        hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy  grid %6d : %7.1f GB/s (read + write)\n", grid, 2.0 * bytes * 5 / ms * 1e-6);
        hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, out, n);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, out, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read  grid %6d : %7.1f GB/s\n", grid, 1.0 * bytes * 5 / ms * 1e-6);
        hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, b, n);
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, b, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("write grid %6d : %7.1f GB/s\n", grid, 1.0 * bytes * 5 / ms * 1e-6);
    }
    const int iters = 20000, blocks = 256 * 8;                // 8 workgroups of 4 waves per CU
    hipLaunchKernelGGL(mfma_f32_k, dim3(blocks), dim3(256), 0, 0, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_f32_k, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mfma f32 16x16x4   : %7.1f TFLOP/s\n", 2.0 * 16 * 16 * 4 * 8.0 * iters * blocks * 4 / ms * 1e-9);
    hipLaunchKernelGGL(mfma_bf16_k, dim3(blocks), dim3(256), 0, 0, out, 10);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mfma_bf16_k, dim3(blocks), dim3(256), 0, 0, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mfma bf16 16x16x32 : %7.1f TFLOP/s\n", 2.0 * 16 * 16 * 32 * 8.0 * iters * blocks * 4 / ms * 1e-9);
    return 0;
}
