// Does the f32 matrix pipe (v_mfma_f32_16x16x4_f32) run beside the vector ALU on gfx950?
// One workgroup of 8 waves per CU (waves w and w+4 share SIMD w&3); each role loops N times over
//   MFMA: 8 independent accumulators, one MFMA each        (8 x 32 cycles per iteration per wave if the pipe is serial)
//   VALU: 64 independent fmaf chains, one step each        (64 x 4 cycles ... v_fma_f32: 1 per cycle per SIMD = 4 cycles/wave-instruction)
// modes: 0 waves 0-3 MFMA, 4-7 idle | 1 waves 0-3 idle, 4-7 VALU | 2 waves 0-3 MFMA + 4-7 VALU | 3 every wave both,
// interleaved in one loop | 4 all 8 waves MFMA | 5 all 8 waves VALU
// hipcc --offload-arch=gfx950 -O3 -o coexec coexec.hip ; ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool MF, bool VA>
__device__ __forceinline__ void body(int n, float a, float b, float* out)
{
    f32x4 acc[8];
    float v[64];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){a, b, a, b};
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = a + i;
    for (int it = 0; it < n; ++it) {
        if (MF) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        if (VA) {
#pragma unroll
            for (int i = 0; i < 64; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        }
    }
    float s = 0.f;
    if (MF)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (VA)
#pragma unroll
        for (int i = 0; i < 64; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;   // never true: keeps the work alive
}

__global__ void __launch_bounds__(512) k(int mode, int n, float a, float b, float* out)
{
    const int wave = threadIdx.x >> 6;
    const bool lo = wave < 4;
    switch (mode) {
    case 0: if (lo) body<true, false>(n, a, b, out); break;
    case 1: if (!lo) body<false, true>(n, a, b, out); break;
    case 2: if (lo) body<true, false>(n, a, b, out); else body<false, true>(n, a, b, out); break;
    case 3: body<true, true>(n, a, b, out); break;
    case 4: body<true, false>(n, a, b, out); break;
    case 5: body<false, true>(n, a, b, out); break;
    }
}

int main()
{
    float* out;
    if (hipMalloc(&out, 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n = 20000, grid = 256;
    const char* names[] = {"4 waves MFMA (1 per SIMD), 4 idle", "4 idle, 4 waves VALU (1 per SIMD)", "4 waves MFMA + 4 waves VALU (one of each per SIMD)",
                           "8 waves, MFMA and VALU interleaved in each", "8 waves MFMA (2 per SIMD)", "8 waves VALU (2 per SIMD)"};
    for (int mode = 0; mode < 6; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, mode, n, 0.999f, 0.001f, out);
            hipEventRecord(e1, 0);
            if (hipEventSynchronize(e1) != hipSuccess) return 2;
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // per SIMD and iteration: MFMA waves issue 8 MFMAs, VALU waves 64 FMAs
        printf("mode %d  %-52s %8.3f ms   %7.1f ns per iteration\n", mode, names[mode], best, best * 1e6 / n);
    }
    hipFree(out);
    return 0;
}
