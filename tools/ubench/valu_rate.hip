// Issue rate of plain VALU instructions on gfx950: independent chains of v_and_b32 / v_add_u32 / v_xad_u32 / v_or_b32
// (K4's recurrence) and of v_fma_f32, 8 waves per SIMD.  Prints wave-instructions per cycle per SIMD-equivalents as
// lane-ops/s.   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t v[8], n[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = seed + threadIdx.x * 7 + i;
        n[i] = ~(seed * (i + 3)) ^ threadIdx.x;
        f[i] = (float)(threadIdx.x + i) * 1e-3f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) {           // and, xad, or  (3 ops)
                    uint32_t t, s;
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "v"(v[i]), "v"(n[i]));
                    asm volatile("v_xad_u32 %0, %1, %2, %1" : "=v"(s) : "v"(v[i]), "v"(t));
                    asm volatile("v_or_b32 %0, %1, %2" : "=v"(v[i]) : "v"(s), "v"(t));
                }
                else if (KIND == 1) {      // and, add, xor, or (4 ops)
                    uint32_t t, s, x;
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "v"(v[i]), "v"(n[i]));
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(s) : "v"(v[i]), "v"(t));
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(v[i]), "v"(t));
                    asm volatile("v_or_b32 %0, %1, %2" : "=v"(v[i]) : "v"(s), "v"(x));
                }
                else {                     // fma x3
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 2) & 7]));
                    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 3) & 7]));
                }
            }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= v[i] ^ __float_as_uint(f[i]);
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int KIND>
static void run(const char *name, int ops_per_body)
{
    uint32_t *out;
    const int grid = 256 * 8, iters = 2000;
    hipMalloc(&out, grid * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<KIND><<<grid, 256>>>(out, 10, 1u);
    hipEventRecord(a);
    k<KIND><<<grid, 256>>>(out, iters, 1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = (double)grid * 4 * iters * 64 * ops_per_body;      // 8 x 8 bodies per iteration
    printf("%-28s %8.3f ms  %7.2f T lane-ops/s  = %.2f cycles per wave64 instruction per SIMD (2.4 GHz, 1024 SIMDs)\n", name, ms,
           wave_instr * 64 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / wave_instr);
    hipFree(out);
}

int main()
{
    run<0>("and + xad + or", 3);
    run<1>("and + add + xor + or", 4);
    run<2>("fma_f32 x 3", 3);
    return 0;
}
