// does v << row_newbcast(sh) (folded by the compiler into v_lshlrev_b32_dpp) do what k3_pair.hip needs?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int S> __device__ inline int row_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + S, 0xf, 0xf, false); }
__global__ void k(const int *vin, const int *shin, int *out)
{
    const int lane = threadIdx.x;
    int v = vin[lane], sh = shin[lane];
    int r5 = (int)((uint32_t)v << row_bcast_i<5>(sh));
    int r0 = (int)((uint32_t)v << row_bcast_i<0>(sh));
    float f = __int_as_float(row_bcast_i<5>(__float_as_int((float)sh))) * (float)v;
    out[lane] = r5; out[64 + lane] = r0; out[128 + lane] = (int)f;
}
// result on MI355X, ROCm 7.2: every lane gets bcast(VALUE) << own count -- the fold is wrong; k3_core.h multiplies instead
int main()
{
    int hv[64], hs[64], ho[192], *dv, *ds, *dout;
    for (int i = 0; i < 64; ++i) { hv[i] = i * 3 + 1; hs[i] = ((i & 15) == 5) ? ((i >> 4) & 1 ? 16 : 0) : 7; }
    hipMalloc(&dv, 256); hipMalloc(&ds, 256); hipMalloc(&dout, 768);
    hipMemcpy(dv, hv, 256, hipMemcpyHostToDevice); hipMemcpy(ds, hs, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dv, ds, dout);
    hipMemcpy(ho, dout, 768, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int sh5 = hs[(i & ~15) + 5], sh0 = hs[(i & ~15)];
        const int e5 = (int)((uint32_t)hv[i] << sh5), e0 = (int)((uint32_t)hv[i] << sh0);
        if (ho[i] != e5 || ho[64 + i] != e0) { if (bad < 8) printf("lane %d: v %d got %d / %d expected %d / %d (mul %d)\n", i, hv[i], ho[i], ho[64 + i], e5, e0, ho[128 + i]); ++bad; }
    }
    printf("dpp shift: %d lanes wrong\n", bad);
    return 0;
}
