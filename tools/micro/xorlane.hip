// v[lane ^ J] without the LDS crossbar: DPP quad_perm / row_half_mirror / row_mirror for J = 1, 2, 4, 8 and gfx950's
// v_permlane16_swap / v_permlane32_swap for J = 16, 32, checked here against __shfl_xor (ds_bpermute).
// hipcc --offload-arch=gfx950 -O3 -o xorlane xorlane.hip ; ./xorlane
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int J>
__device__ __forceinline__ int xor_lane(int v)
{
    const int lane = threadIdx.x & 63;
    if (J == 1) return dpp<0xB1>(v);                  // quad_perm [1,0,3,2]
    if (J == 2) return dpp<0x4E>(v);                  // quad_perm [2,3,0,1]
    if (J == 4) return dpp<0x1B>(dpp<0x141>(v));      // row_half_mirror (^7) then quad_perm [3,2,1,0] (^3)
    if (J == 8) return dpp<0x141>(dpp<0x140>(v));     // row_mirror (^15) then row_half_mirror (^7)
    if (J == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
}
__global__ void k(int* out)
{
    const int v = 1000 + 7 * (int)threadIdx.x;
    int bad = 0;
    bad |= (xor_lane<1>(v) != __shfl_xor(v, 1)) << 0;
    bad |= (xor_lane<2>(v) != __shfl_xor(v, 2)) << 1;
    bad |= (xor_lane<4>(v) != __shfl_xor(v, 4)) << 2;
    bad |= (xor_lane<8>(v) != __shfl_xor(v, 8)) << 3;
    bad |= (xor_lane<16>(v) != __shfl_xor(v, 16)) << 4;
    bad |= (xor_lane<32>(v) != __shfl_xor(v, 32)) << 5;
    out[threadIdx.x] = bad;
}
int main()
{
    int* d;
    int h[64];
    if (hipMalloc(&d, sizeof h) != hipSuccess) return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    int all = 0;
    for (int i = 0; i < 64; ++i) all |= h[i];
    printf("xor-lane mismatch mask over all lanes: 0x%x (bit k = J = 2^k wrong)\n", all);
    for (int i = 0; i < 64; ++i)
        if (h[i]) printf(" lane %d: 0x%x\n", i, h[i]);
    return all ? 3 : 0;
}
