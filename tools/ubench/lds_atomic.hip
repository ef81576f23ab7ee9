// micro-benchmark: LDS update throughput on gfx950 -- ds_add_u32 vs ds_add_f32 vs plain read+write,
// random addresses inside a 2048-word window per wave (the K3 access pattern).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(64) void k(const int *idx, int iters, float *out)
{
    __shared__ float acc[2048];
    __shared__ unsigned int acci[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) { acc[i] = 0; acci[i] = 0; }
    __syncthreads();
    int a[16];
    for (int j = 0; j < 16; ++j) a[j] = idx[(blockIdx.x * 16 + j) * 64 + lane] & 2047;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = (a[j] + it * 67) & 2047;
            if (MODE == 0) atomicAdd(&acci[p], 3u);
            else if (MODE == 1) atomicAdd(&acc[p], 1.5f);
            else if (MODE == 2) { acc[p] = acc[p] + 1.5f; __builtin_amdgcn_wave_barrier(); }
            else if (MODE == 3) { atomicMax(&acci[p], (unsigned)it); }
        }
    }
    __syncthreads();
    float s = 0;
    for (int i = lane; i < 2048; i += 64) s += acc[i] + (float)acci[i];
    if (s == 12345.f) out[0] = s;
}

int main()
{
    const int blocks = 256 * 16, iters = 2000;
    int *d_idx; float *d_out;
    int *h = (int *)malloc(blocks * 16 * 64 * sizeof(int));
    uint32_t x = 12345;
    for (int i = 0; i < blocks * 16 * 64; ++i) { x = x * 1664525u + 1013904223u; h[i] = x >> 8; }
    hipMalloc(&d_idx, blocks * 16 * 64 * sizeof(int));
    hipMalloc(&d_out, 4);
    hipMemcpy(d_idx, h, blocks * 16 * 64 * sizeof(int), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"ds_add_u32", "ds_add_f32", "read+fadd+write", "ds_max_u32"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d_idx, iters, d_out);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d_idx, iters, d_out);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d_idx, iters, d_out);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d_idx, iters, d_out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double updates = (double)blocks * iters * 16 * 64;
            if (rep) printf("%-16s %8.3f ms  %.3e lane-updates/s  %.2f lanes/clk/CU @2.4GHz\n", names[mode], ms,
                            updates / (ms * 1e-3), updates / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
