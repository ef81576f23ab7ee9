// does global_load_dwordx4 work at 8-byte (not 16-byte) alignment on gfx950 / this ROCm?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const int *src, int *dst, int shift)
{
    const int4 v = *(const int4 *)(src + shift + threadIdx.x * 4);   // shift = 2 -> 8-byte aligned only
    dst[threadIdx.x] = v.x + v.y * 3 + v.z * 5 + v.w * 7;
}
int main()
{
    int *s, *d; int h[1024], r[64];
    for (int i = 0; i < 1024; ++i) h[i] = i * 11 + 1;
    hipMalloc(&s, sizeof(h)); hipMalloc(&d, sizeof(r));
    hipMemcpy(s, h, sizeof(h), hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, shift);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(r, d, sizeof(r), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 64; ++t) { const int *p = h + shift + t * 4; bad += r[t] != p[0] + p[1] * 3 + p[2] * 5 + p[3] * 7; }
        printf("shift %d (byte offset %d): %s, %d wrong\n", shift, shift * 4, hipGetErrorString(e), bad);
    }
    return 0;
}
