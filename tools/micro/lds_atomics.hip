// Microbenchmark: LDS add throughput on gfx950 -- float atomic (ds_add_f32), u32 atomic (ds_add_u32),
// plain read-add-write; lane-private addresses vs random addresses in a small table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE, bool RANDOM>
__global__ __launch_bounds__(256) void k(float* out, int iters, int table) {
    extern __shared__ float sm[];
    for (int i = threadIdx.x; i < table; i += blockDim.x) sm[i] = 0.f;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    float v = 1.0f + threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            int idx = RANDOM ? (int)((s >> 8) % (uint32_t)table) : (int)(((it * 8 + u) * 256 + threadIdx.x) % table);
            if (MODE == 0) __hip_atomic_fetch_add(&sm[idx], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) __hip_atomic_fetch_add((unsigned int*)&sm[idx], (unsigned int)(v * 1024.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else sm[idx] += v;
        }
    }
    __syncthreads();
    float acc = 0.f;
    for (int i = threadIdx.x; i < table; i += blockDim.x) acc += sm[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, bool RANDOM>
void run(const char* name, int table) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 1024;
    hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(256), table * 4, 0, out, 10, table);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(blocks), dim3(256), table * 4, 0, out, iters, table);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)blocks * 256 * iters * 8;
    printf("%-34s table %5d : %8.3f ms  %7.1f G lane-adds/s  (%.2f ns per wave-instr per CU)\n", name, table, ms,
           ops / ms * 1e-6, ms * 1e6 / (ops / 64 / 256));
    hipFree(out);
}

int main() {
    for (int table : {256, 2048, 16384}) {
        run<0, false>("ds_add_f32 lane-strided", table);
        run<0, true>("ds_add_f32 random", table);
        run<1, false>("ds_add_u32 lane-strided", table);
        run<1, true>("ds_add_u32 random", table);
        run<2, false>("read-add-write lane-strided", table);
        run<2, true>("read-add-write random (racy)", table);
    }
    return 0;
}
