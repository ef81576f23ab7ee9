// micro-benchmark: what does rocprofv3's FETCH_SIZE report for a known byte count read with 8-byte-per-lane
// loads (K3's posting loads: global_load_dwordx2, 16 lanes per aligned 128-byte line) vs 16-byte-per-lane loads?
// The buffer (1 GiB) is far larger than L2 + Infinity Cache, every byte is read exactly once per launch.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib.bin
// run  : rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o calib -- tools/ubench/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void read_x2(const int2 *__restrict__ p, int64_t n, int *out)
{
    int acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int2 v = p[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x12345678) out[0] = acc;
}

__global__ __launch_bounds__(256) void read_x4(const int4 *__restrict__ p, int64_t n, int *out)
{
    int acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678) out[0] = acc;
}

int main()
{
    const size_t bytes = (size_t)1 << 30;
    void *buf;
    int *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_x2, dim3(256 * 16), dim3(256), 0, 0, (const int2 *)buf, (int64_t)(bytes / 8), out);
        hipLaunchKernelGGL(read_x4, dim3(256 * 16), dim3(256), 0, 0, (const int4 *)buf, (int64_t)(bytes / 16), out);
    }
    hipDeviceSynchronize();
    printf("each launch reads %zu bytes (= %.1f KB)\n", bytes, bytes / 1024.0);
    return 0;
}
