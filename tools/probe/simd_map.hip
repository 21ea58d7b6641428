// Scratch probe: which SIMD does wave i of a 512-thread workgroup land on?  (HW_ID bits 5:4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) k(int* out)
{
    const int w = threadIdx.x >> 6;
    const int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
    const int cu = __builtin_amdgcn_s_getreg((3 << 11) | (8 << 6) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + w] = simd | (cu << 8);
}
int main()
{
    int* d;
    hipMalloc(&d, 64 * 8 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(64), dim3(512), 70000, 0, d);
    int h[64 * 8];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; ++b) {
        printf("wg %d:", b);
        for (int w = 0; w < 8; ++w) printf(" w%d->simd%d(cu%d)", w, h[b * 8 + w] & 3, h[b * 8 + w] >> 8);
        printf("\n");
    }
    return 0;
}
