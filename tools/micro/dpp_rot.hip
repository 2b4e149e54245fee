// Microbenchmark: cost of rotating a register across the 64 lanes of a wave on gfx950 -- DPP wave_ror:1
// (one instruction, whole wave), DPP row_ror:1 (16-lane rows), ds_bpermute_b32, v_permlane32_swap, and a plain
// v_pk_fma_f32 as the yardstick.  Also prints what wave_ror:1 does to the lane index (direction check).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// MODE 0: pk_fma only; 1: + wave_ror:1 per fma; 2: + row_ror:1; 3: + ds_bpermute; 4: wave_ror fused in v_add (dpp operand)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float r[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) r[u] = threadIdx.x * 0.001f + u;
    f32x2 acc[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) acc[u] = f32x2{0.f, 0.f};
    const int src = ((threadIdx.x + 1) & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            if (MODE == 1) r[u] = dpp<0x13C>(r[u]);            // wave_ror:1
            if (MODE == 2) r[u] = dpp<0x121>(r[u]);            // row_ror:1
            if (MODE == 3) r[u] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, r[u])));
            if (MODE == 5) r[u] = dpp<0x134>(r[u]);            // wave_rol:1
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            f32x2 a = {r[2 * u], r[2 * u + 1]};
            acc[u] = __builtin_elementwise_fma(a, a, acc[u]);
            acc[u] = __builtin_elementwise_fma(a, acc[u], a);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) s += acc[u].x + acc[u].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void direction(int* out) {
    int v = threadIdx.x;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false);
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x121, 0xf, 0xf, false);
}

template <int MODE>
void run(const char* name) {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000, blocks = 4096;
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: blocks*4 waves / 1024 SIMDs, each iters * (12 pk_fma + 12 rot)
    const double wave_iters = (double)blocks * 4 / 1024.0 * iters;
    printf("%-28s %8.3f ms   %.1f cycles per (12 pk_fma + 12 rotations) at 2.4 GHz\n", name, ms,
           ms * 1e-3 * 2.4e9 / wave_iters);
    hipFree(out);
}

int main() {
    int* d; hipMalloc(&d, 192 * 4);
    hipLaunchKernelGGL(direction, dim3(1), dim3(64), 0, 0, d);
    int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_ror:1 lane0<-%d lane1<-%d lane63<-%d | wave_rol:1 lane0<-%d lane63<-%d | row_ror:1 lane0<-%d lane1<-%d lane16<-%d\n",
           h[0], h[1], h[63], h[64], h[127], h[128], h[129], h[144]);
    run<0>("pk_fma only");
    run<1>("+ dpp wave_ror:1");
    run<5>("+ dpp wave_rol:1");
    run<2>("+ dpp row_ror:1");
    run<3>("+ ds_bpermute");
    return 0;
}
