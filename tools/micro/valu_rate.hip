// Issue rate of the f32 VALU forms the pair loops are written in, measured on the box (round 5): does a packed
// v_pk_fma_f32 (two fma per lane) issue in the time of one v_fma_f32, i.e. is the 157.3 TF "vector peak" reachable by
// packing, or does it take two passes?  Also v_pk_mul / v_pk_add, v_rcp_f32, v_rndne_f32 + v_med3_f32 (the minimum image),
// ds_read_b96 / ds_read_b128 gathers at random 12 / 16-byte slots (the column-tile candidates).
// Each kernel: 16 independent accumulator chains per lane (no dependency stalls), `iters` x 16 instructions, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int NCH = 16;

__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b) {
    float acc[NCH];
    for (int k = 0; k < NCH; ++k) acc[k] = threadIdx.x * 1e-3f + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
    }
    float s = 0.f;
    for (int k = 0; k < NCH; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk_fma(float* out, int iters, float a, float b) {
    f32x2 acc[NCH];
    const f32x2 av = {a, a * 1.01f}, bv = {b, b * 0.99f};
    for (int k = 0; k < NCH; ++k) acc[k] = f32x2{threadIdx.x * 1e-3f + k, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(av), "v"(bv));
    }
    float s = 0.f;
    for (int k = 0; k < NCH; ++k) s += acc[k].x + acc[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk_mul(float* out, int iters, float a, float b) {
    f32x2 acc[NCH];
    const f32x2 av = {a, a * 1.01f};
    for (int k = 0; k < NCH; ++k) acc[k] = f32x2{threadIdx.x * 1e-3f + k, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[k]) : "v"(av));
    }
    float s = 0.f;
    for (int k = 0; k < NCH; ++k) s += acc[k].x + acc[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_pk_add(float* out, int iters, float a, float b) {
    f32x2 acc[NCH];
    const f32x2 av = {a, a * 1.01f};
    for (int k = 0; k < NCH; ++k) acc[k] = f32x2{threadIdx.x * 1e-3f + k, 1.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(acc[k]) : "v"(av));
    }
    float s = 0.f;
    for (int k = 0; k < NCH; ++k) s += acc[k].x + acc[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_rcp(float* out, int iters, float a, float b) {
    float acc[NCH];
    for (int k = 0; k < NCH; ++k) acc[k] = threadIdx.x * 1e-3f + k + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[k]));
    }
    float s = 0.f;
    for (int k = 0; k < NCH; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_rndne_med3(float* out, int iters, float a, float b) {
    float acc[NCH];
    for (int k = 0; k < NCH; ++k) acc[k] = threadIdx.x * 1e-3f + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NCH; k += 2) {
            asm volatile("v_rndne_f32 %0, %0" : "+v"(acc[k]));
            asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(acc[k + 1]) : "v"(a), "v"(b));
        }
    }
    float s = 0.f;
    for (int k = 0; k < NCH; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// LDS gathers: every lane reads `iters` x 8 slots at pseudo-random positions of a 1 536-slot table (the tile size of lj4096)
template <int BYTES>
__global__ __launch_bounds__(256) void k_lds_gather(float* out, int iters, int nslot) {
    extern __shared__ float tab[];
    for (int t = threadIdx.x; t < nslot * (BYTES / 4); t += blockDim.x) tab[t] = t * 1e-3f;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x = x * 1664525u + 1013904223u;
            const int slot = (int)((x >> 8) % (unsigned)nslot);
            if (BYTES == 12) {
                struct __attribute__((packed, aligned(4))) R3 { float a, b, c; };
                const R3 r = *reinterpret_cast<const R3*>(tab + 3 * slot);
                s += r.a + r.b + r.c;
            } else {
                const float4 r = *reinterpret_cast<const float4*>(tab + 4 * slot);
                s += r.x + r.y + r.z + r.w;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float timed(F&& launch, hipEvent_t e0, hipEvent_t e1) {
    launch();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 5.f;
}

int main() {
    float* out;
    const int grid = 256 * 8, block = 256;                       // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    CK(hipMalloc(&out, sizeof(float) * grid * block));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    const double insts = (double)grid * (block / 64) * iters * NCH;          // wave-instructions per launch
    const double simd_cycles_per_s = 1024.0 * 2.4e9;
#define RUN(NAME, KERN, FLOP_PER_LANE)                                                                           \
    do {                                                                                                             \
        const float ms = timed([&] { hipLaunchKernelGGL(KERN, dim3(grid), dim3(block), 0, 0, out, iters, 1.0001f, 0.9999f); }, e0, e1); \
        printf("%-22s %8.3f ms  %6.2f cycles per wave-instruction and SIMD  %7.1f TFLOP/s\n", NAME, ms,         \
               ms * 1e-3 * simd_cycles_per_s / insts, insts * 64.0 * (FLOP_PER_LANE) / (ms * 1e-3) / 1e12);          \
    } while (0)
    RUN("v_fma_f32", k_fma, 2);
    RUN("v_pk_fma_f32", k_pk_fma, 4);
    RUN("v_pk_mul_f32", k_pk_mul, 2);
    RUN("v_pk_add_f32", k_pk_add, 2);
    RUN("v_rcp_f32", k_rcp, 1);
    RUN("v_rndne + v_med3", k_rndne_med3, 1);
    for (int bytes : {12, 16}) {
        const int nslot = 1536, it2 = 2048;
        const double reads = (double)grid * (block / 64) * it2 * 8;
        const float ms = bytes == 12
            ? timed([&] { hipLaunchKernelGGL(k_lds_gather<12>, dim3(grid), dim3(block), nslot * 12, 0, out, it2, nslot); }, e0, e1)
            : timed([&] { hipLaunchKernelGGL(k_lds_gather<16>, dim3(grid), dim3(block), nslot * 16, 0, out, it2, nslot); }, e0, e1);
        printf("ds_read_b%-3d random     %8.3f ms  %6.2f CU-cycles per wave-instruction (incl. ~6 VALU of index arithmetic each)  %6.1f TB/s\n",
               bytes * 8, ms, ms * 1e-3 * 256.0 * 2.4e9 / reads, reads * 64.0 * bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
