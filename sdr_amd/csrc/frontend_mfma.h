// Channel filter for SMALL decimations (fp32-FMA bound: 4*ntaps/D flop per input
// sample, SURVEY.md 8d "fp32 vector FMA for a2 at D=1 (shape R)") on the f32
// matrix cores.  v_mfma_f32_16x16x4_f32 is bit for bit a k-ordered fmaf chain
// (one rounding per product, no wider accumulation), so the FIR written as a
// banded Toeplitz product keeps the oracle's numeric contract exactly:
//
//   Y[i][g] = sum_k  A[i][k] * B[k][g]      i = output inside a group of 16
//   A[i][k] = rtaps[k - i*D]  (0 outside the taps)        g = group of 16 outputs
//   B[k][g] = x[16*D*g + k]   (re and im as two products sharing A)
//
// Row i's non-zero terms appear with ascending tap index = oldest sample first,
// the zero terms are fma(0, x, acc) = acc.  K = 15*D + ntaps, i.e. a fraction
// ntaps/(15*D+ntaps) of the MFMA work is useful (89 % for 121 taps, D = 1).
// A non-finite input sample (Inf, NaN) would also spoil the <= 15 outputs of its
// group whose window does not contain it (0*Inf = NaN in the zero-padded
// products).  Those come out as NaN, so they are found on the accumulators:
// a wavefront that holds a non-finite result recomputes its 256 outputs as the
// plain fmaf chain over the taps alone (mfma_exact_redo) -- what the oracle
// and the vector kernels do -- and the spoiling stays local to the ntaps
// outputs whose window holds the sample (round 3; tests/test_gpu_mfma.py).
//
// One wavefront = 256 outputs (16 groups x 16); a workgroup of nw <= 4 waves
// stages (256*nw-1)*D + ntaps samples in LDS.  Group g of a wave reads its
// samples with ds_read_b64 at stride 16*D samples: one 16-B pad after every
// 16*D samples makes the 32 lanes of a half-wave hit 32 distinct bank pairs
// ((8D+1) is odd).  The staging DMA (global_load_lds, 16 B per lane) writes the
// padded image directly: LDS chunk c takes source chunk c - c/(8D+1).
// After the K loop the tile's outputs go through LDS into output order and the
// epilogue (NCO, demod, gain, hysteresis scan, slicer, keep_one_in_n) runs with
// 4 consecutive outputs per thread.
#pragma once
#include "frontend_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Hysteresis + slicer for N consecutive outputs per thread (see slice_tile).  i0 = index inside the tile
// of the thread's first output (o_base = output index of the tile's index 0): everything per lane is
// 32-bit, the 64-bit parts are wave-uniform.
// FULL (uniform, chosen by the caller): every output of the tile exists, keep_n == 1 and the tile's bytes are
// a contiguous run of the row -- then only valid[0] is looked at (the discriminator's tile index 0 belongs to
// the previous tile) and the bytes leave through `pack`, an LDS area of 4 * (64*N + 16) bytes that nothing
// else uses after this function's barrier: each wavefront lays its 64*N bytes out at the global address's
// offset inside a dword and writes whole dwords, coalesced (N byte stores per lane at stride N touched every
// cache line of the run N times: 0.034 of 0.238 ms on the TFA discriminator shape).
// per-lane event masks of N outputs: bit e of H = output e sets the state (v > hi), of K = it clears it (v < lo)
template <int N, bool FULL>
__device__ __forceinline__ void slice_masks_xn(const bool* valid, const float* v, float lo, float hi, unsigned& H, unsigned& K)
{
    H = 0;
    K = 0;
    if (FULL) {
        // two instructions per output and mask: the compare's lane bit enters the mask as the carry of mask + mask
        // (outputs walked from the last to the first); the compiler's form -- compare, select 1 << e, or -- took ~7
        // per output for both masks.  "not above hi" and valid[0] are applied to the finished masks.
#pragma unroll
        for (int e = N - 1; e >= 0; --e) {
            asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(H) : "s"(hi), "v"(v[e]) : "vcc");
            asm("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(K) : "s"(lo), "v"(v[e]) : "vcc");
        }
        const unsigned keep = valid[0] ? ~0u : ~1u;
        H &= keep;
        K &= ~H & keep;
        return;
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const bool ok = valid[e];
        const bool up = v[e] > hi;
        H |= (unsigned)(ok && up) << e;
        K |= (unsigned)(ok && !up && v[e] < lo) << e;
    }
}

// The slicer proper, from the masks (a caller that knows the class of an output without its value -- the
// threshold-only discriminator -- builds them itself).
// The threshold_ff recurrence  state_e = H_e | (state_{e-1} & ~K_e)  is the carry chain of an addition (generate = H,
// propagate = neither event), so the N states of a lane come from ONE add once its carry-in is known, and "has an
// event", "state behind the last event" and "index of the first event" are bit scans -- the per-output select chains
// of rounds 1-2 were ~15 vector instructions per output (~190 of the ~1050 per wave on the TFA shape).
template <int N, bool FULL = false>
__device__ __forceinline__ void slice_tile_hk(const bool* valid, const unsigned H, const unsigned K, int i0, long long o_base,
                                              int keep_n, long long keep_c0,
                                              int8_t* __restrict__ bits, TileSum* __restrict__ ts_out,
                                              int* __restrict__ sh /* 12 ints */, unsigned char* pack = nullptr)
{
    static_assert(N <= 30, "the carry out of bit N-1 must fit the word");
    const unsigned E = H | K;
    const int has = E != 0;
    const int last = has ? (int)((H >> (31 - __builtin_clz(E | (E == 0)))) & 1u) : 0;
    const int firstk = has ? __builtin_ctz(E) : 0;
    const unsigned long long any = __ballot(has);
    const unsigned long long sset = __ballot(has && last);
    const int lane = lane_id(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    const int fk = __shfl(firstk, any ? __builtin_ctzll(any) : 0);
    if (lane == 0) {
        sh[wave] = any != 0;
        sh[4 + wave] = any ? (int)((sset >> (63 - __builtin_clzll(any))) & 1) : 0;
        sh[8 + wave] = any ? ((wave * 64 + __builtin_ctzll(any)) * N + fk) : -1;
    }
    __syncthreads();
    const unsigned long long below = any & ((1ull << lane) - 1ull);
    int st = 0;
    if (below) {
        st = (int)((sset >> (63 - __builtin_clzll(below))) & 1);
    } else {
        for (int w = wave - 1; w >= 0; --w)
            if (sh[w]) {
                st = sh[4 + w];
                break;
            }
    }
    if (threadIdx.x == 0) {
        int first = -1, lst = 0;
        for (int w = 0; w < nw; ++w)
            if (sh[w]) {
                if (first < 0) first = sh[8 + w];
                lst = sh[4 + w];
            }
        ts_out->first_evt = first < 0 ? -1 : (int32_t)(o_base + first);
        ts_out->last_state = lst;
    }
    // all N states of the lane: carries of (H | hold) + H + carry-in; bit e+1 of the carry vector = state behind output e
    constexpr unsigned ALLN = (1u << N) - 1u;
    const unsigned X = (H | ~E) & ALLN;
    const unsigned states = (((X + H + (unsigned)st) ^ X ^ H) >> 1) & ALLN;
    const long long ub = o_base - keep_c0;   // uniform: emitted-byte index of the tile's index 0 (keep_n == 1)
    if (FULL) {
        // wave w owns tile indices [64*N*w, 64*N*(w+1)); index 0 is not emitted when valid[0] of thread 0 is false
        const int skip0 = __shfl(valid[0] ? 0 : 1, 0) & (wave == 0 ? 1 : 0);   // uniform per wave
        const int f = wave * 64 * N + skip0, cnt = 64 * N - skip0;
        int8_t* const G = bits + ub + f;                                        // first byte of this wave's run
        const int a = (int)(reinterpret_cast<uintptr_t>(G) & 3);
        unsigned char* const wb = pack + wave * (64 * N + 16);
        // the lane's N bytes: four state bits become four bytes by one multiplication ((x & 15) * 0x204081 puts bit k
        // of x at bit 8k, the shifted copies do not overlap); outputs 1.. leave as unaligned 8-/4-/1-byte LDS writes
        const int pos0 = a + (i0 - f);
        if (pos0 >= a) wb[pos0] = (unsigned char)(states & 1u);
        typedef unsigned long long u64u __attribute__((aligned(1)));
        typedef unsigned u32u __attribute__((aligned(1)));
        auto spread4 = [](unsigned x) -> unsigned { return ((x & 15u) * 0x00204081u) & 0x01010101u; };
        constexpr int N8 = (N - 1) / 8, N4 = ((N - 1) - 8 * N8) / 4, N1 = (N - 1) - 8 * N8 - 4 * N4;
#pragma unroll
        for (int g = 0; g < N8; ++g) {
            const int e = 1 + 8 * g;
            *reinterpret_cast<u64u*>(wb + pos0 + e) = (unsigned long long)spread4(states >> e) | ((unsigned long long)spread4(states >> (e + 4)) << 32);
        }
#pragma unroll
        for (int g = 0; g < N4; ++g) {
            const int e = 1 + 8 * N8 + 4 * g;
            *reinterpret_cast<u32u*>(wb + pos0 + e) = spread4(states >> e);
        }
#pragma unroll
        for (int g = 0; g < N1; ++g) {
            const int e = 1 + 8 * N8 + 4 * N4 + g;
            wb[pos0 + e] = (unsigned char)((states >> e) & 1u);
        }
        // whole dwords [d0, d1) of the image are this wave's alone; the ragged ends go as bytes
        const int d0 = (a + 3) >> 2, d1 = (a + cnt) >> 2;
        const unsigned* const wd = reinterpret_cast<const unsigned*>(wb);
        unsigned* const gd = reinterpret_cast<unsigned*>(G - a);
        for (int d = d0 + lane; d < d1; d += 64) gd[d] = wd[d];
        const int nhead = 4 * d0 - a, t0 = 4 * d1, ntail = a + cnt - t0;
        if (lane < nhead) G[lane] = (int8_t)wb[a + lane];
        if (lane < ntail) G[t0 - a + lane] = (int8_t)wb[t0 + lane];
        return;
    }
    if (keep_n == 1 && ub >= 0) {           // every tile but a stream's first: no per-lane 64-bit arithmetic
        int8_t* __restrict__ pb = bits + ub;
#pragma unroll
        for (int e = 0; e < N; ++e)
            if (valid[e]) pb[i0 + e] = (int8_t)((states >> e) & 1u);
        return;
    }
    // keep_one_in_n: one division per thread (32-bit when the index allows), then
    // the remainder is stepped along the thread's N outputs
    const long long r0 = ub + i0;
    long long quo = 0;
    int rem = 0;
    if (keep_n > 1 && r0 + (N - 1) >= 0) {
        const long long rb = r0 < 0 ? 0 : r0;  // first non-negative index of this thread
        if (rb < 0x7fffffffLL) {
            quo = (unsigned)rb / (unsigned)keep_n;
            rem = (int)((unsigned)rb - (unsigned)quo * (unsigned)keep_n);
        } else {
            quo = rb / keep_n;
            rem = (int)(rb - quo * keep_n);
        }
    }
#pragma unroll
    for (int e = 0; e < N; ++e) {
        const int8_t se = (int8_t)((states >> e) & 1u);
        const long long r = r0 + e;
        if (r < 0) continue;
        if (keep_n == 1) {
            if (valid[e]) bits[r] = se;
        } else {
            if (valid[e] && rem == 0) bits[quo] = se;
            if (++rem == keep_n) {
                rem = 0;
                ++quo;
            }
        }
    }
}

template <int N, bool FULL = false>
__device__ __forceinline__ void slice_tile_xn(const bool* valid, const float* v, int i0, long long o_base,
                                              float lo, float hi, int keep_n, long long keep_c0,
                                              int8_t* __restrict__ bits, TileSum* __restrict__ ts_out,
                                              int* __restrict__ sh /* 12 ints */, unsigned char* pack = nullptr)
{
    unsigned H, K;
    slice_masks_xn<N, FULL>(valid, v, lo, hi, H, K);
    slice_tile_hk<N, FULL>(valid, H, K, i0, o_base, keep_n, keep_c0, bits, ts_out, sh, pack);
}

// The K loop of one wavefront: 256 outputs (16 groups x 16) from the padded sample image `smp` and the tap
// table at `ldsf` (tapL[15D + idx] = rtaps[idx]).  `odd` = the image starts one sample before the tile.
__device__ __forceinline__ void mfma_kloop(const float* ldsf, const float* smp, const int D, const int ksteps,
                                           const int odd, const int lane, const int wave, f32x4& accR, f32x4& accI)
{
    const int BK = 16 * D;
    const int g = lane & 15, kq = lane >> 4;
    const int G = wave * 16 + g;                       // group index inside the workgroup tile
    const float* ta = ldsf + (15 * D + kq - D * g);    // A[i=g][k]: tap index 4kk + kq - g*D
    const float* sb = smp + 2 * (odd + BK * G + kq) + 4 * G;
    // K steps in groups of 4 (= 16 samples; a pad block is D groups): constant
    // offsets inside a group, one pointer bump per group, the pad added per
    // block.  A lane that starts at offset 4 (odd tile start, kq = 3) crosses
    // into the next block one step early: `late` on the block's last step.
    const int late = (odd + kq >= 4) ? 4 : 0;
    const int ngrp = ksteps >> 2, tail = ksteps & 3;
    int gi = 0;
    float a[4], an[4];
    float2 b[4], bn[4];
    auto load = [&](float* av, float2* bv) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) av[s4] = ta[4 * s4];
#pragma unroll
        for (int s4 = 0; s4 < 3; ++s4) bv[s4] = *reinterpret_cast<const float2*>(sb + 8 * s4);
        bv[3] = *reinterpret_cast<const float2*>(sb + 24 + ((gi == D - 1) ? late : 0));
        ta += 16;
        sb += 32;
        if (++gi == D) {
            gi = 0;
            sb += 4;
        }
    };
    auto mfma8 = [&](const float* av, const float2* bv) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            accR = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4].x, accR, 0, 0, 0);
            accI = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4].y, accI, 0, 0, 0);
        }
    };
    auto mfma_tail = [&](const float* av, const float2* bv) {
#pragma unroll
        for (int s4 = 0; s4 < 3; ++s4)
            if (s4 < tail) {
                accR = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4].x, accR, 0, 0, 0);
                accI = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4].y, accI, 0, 0, 0);
            }
    };
    // Two operand sets used alternately (the loop is unrolled by two groups): the operands of the next group are
    // fetched while this group's 8 MFMAs run, and no register is copied -- twelve v_mov per group before, which
    // cost MFMA time one for one (the f32 MFMA shares the vector pipe).
    load(a, b);
    int grp = 0;
    for (; grp + 2 <= ngrp; grp += 2) {
        load(an, bn);
        mfma8(a, b);
        load(a, b);
        mfma8(an, bn);
    }
    if (grp < ngrp) {
        load(an, bn);
        mfma8(a, b);
        mfma_tail(an, bn);
    } else
        mfma_tail(a, b);
}

// true iff some of the lane's eight results is Inf or NaN (a finite sum that overflows is a harmless false alarm:
// the redo produces the same values)
__device__ __forceinline__ bool acc_nonfinite(const f32x4& r, const f32x4& i)
{
    const float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((i[0] + i[1]) + (i[2] + i[3]));
    return !(__builtin_fabsf(s) < __builtin_inff());
}
// The wave's 256 outputs once more without the zero-padded products: lane's row 4*(lane>>4) + r of column
// lane & 15, window = the ntaps samples from position odd + 16*D*G + row*D on (padded image: position p at
// float 2p + 4*(p / BK)), taps ascending = oldest sample first.  Rare (a capture that holds Inf / NaN): rolled.
__device__ __forceinline__ void mfma_exact_redo(const float* ldsf, const float* smp, const int D, const int ntaps,
                                             const int odd, const int lane, const int wave, f32x4& accR, f32x4& accI)
{
    const int BK = 16 * D, G = wave * 16 + (lane & 15);
    const float* const rt = ldsf + 15 * D;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const int p0 = odd + BK * G + (4 * (lane >> 4) + r) * D;
        float re = 0.f, im = 0.f;
#pragma unroll 1
        for (int j = 0; j < ntaps; ++j) {
            const int p = p0 + j;
            const float2 xv = *reinterpret_cast<const float2*>(smp + 2 * p + 4 * (p / BK));
            re = __builtin_fmaf(rt[j], xv.x, re);
            im = __builtin_fmaf(rt[j], xv.y, im);
        }
        accR[r] = re;
        accI[r] = im;
    }
}

#define MFMA_ROT_K 4   // 16-B loads in flight per lane in the rotating staging loop (8: 145 VGPRs, 3 waves per SIMD)
// blockDim.x = 64*nw, P.tile_out = 256*nw - q.  Dynamic LDS: tap table | padded samples (aliased by the output tile).
__global__ void __launch_bounds__(256) k_frontend_mfma(FrontParams P)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int sh[12];
    // discriminator: the 257-entry arctangent table sits in LDS (two dependent lookups per output
    // were two L1/L2 round trips each); filled here, first read after the barriers below
    __shared__ float s_atan[257];
    if (P.demod == SDR_DEMOD_QUAD)
        for (int i = threadIdx.x; i < 257; i += blockDim.x) s_atan[i] = P.atan_tab[i];
    float* const ldsf = reinterpret_cast<float*>(smem_raw);

    const int T = blockDim.x, nw = T >> 6, TB = nw * 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int tile = blockIdx.x, cap = blockIdx.y;
    const int D = P.D, ntaps = P.ntaps, BK = 16 * D;
    const int q = (P.demod == SDR_DEMOD_QUAD) ? 1 : 0;
    const long long o0 = (long long)tile * P.tile_out;
    const long long m_t0 = o0 - q;                         // FIR output index of the tile's first output
    const float2* __restrict__ x = P.x + (long long)cap * P.stride;
    const long long lo_idx = P.in0 + m_t0 * D - (ntaps - 1);  // first input sample of the tile
    const int ksteps = (15 * D + ntaps + 3) >> 2;
    const int ntl = (15 * D + 4 * (ksteps + 8) + 3) & ~3;  // tap table floats (+2 groups: operand prefetch), 16-B multiple
    const int npos = (TB - 1) * D + ntaps + 40;            // staged sample positions (K rounding, alignment, prefetch slack)
    float* const smp = ldsf + ntl;

    // ---- tap table: tapL[15D + idx] = rtaps[idx], zero outside
    for (int i = t; i < ntl; i += T) {
        const int idx = i - 15 * D;
        ldsf[i] = (idx >= 0 && idx < ntaps) ? P.rtaps[idx] : 0.f;
    }

    // ---- stage the samples in the padded layout: position p -> float 2p + 4*(p / BK)
    auto fetch = [&](long long idx) -> float2 {
        if (idx >= 0) return idx < P.n ? x[idx] : make_float2(0.f, 0.f);
        if (P.hist && idx >= -(long long)P.hist_len) return P.hist[P.hist_len + idx];
        return make_float2(0.f, 0.f);
    };
    const long long gidx = (long long)cap * P.stride + lo_idx;
    int odd = (int)(gidx & 1);
    const long long lo_al = lo_idx - odd;
    const int nsrc = (npos + odd + 1) >> 1;                // 16-B source chunks
    if (P.vec_ok && !P.pre_tab) {
        // chunks inside the capture by DMA; the few that touch its edges (zero
        // padding / streaming history) by guarded loads of their own lanes
        const int per = 8 * D + 1;                         // LDS chunks per pad block (last one is the pad)
        const int nlds = nsrc + nsrc / (8 * D) + 1;
        // (two loops: the DMA loop holds no other memory operation, see k_frontend)
        for (int c0 = (t & ~63); c0 < nlds; c0 += T) {
            const int c = c0 + lane;
            const int blk = c / per, r = c - blk * per;
            const int sc = blk * 8 * D + r;
            const long long s = lo_al + 2LL * sc;
            if (c < nlds && r < 8 * D && sc < nsrc && s >= 0 && s + 2 <= P.n)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void*)(x + s),
                    (__attribute__((address_space(3))) void*)(smp + 4 * c0), 16, 0, SDR_GLDS_AUX);
        }
        if (lo_al < 0 || lo_al + 2LL * nsrc > P.n) {   // uniform: only edge tiles have guarded chunks
            for (int c = t; c < nlds; c += T) {
                const int blk = c / per, r = c - blk * per;
                const int sc = blk * 8 * D + r;
                const long long s = lo_al + 2LL * sc;
                if (r < 8 * D && sc < nsrc && !(s >= 0 && s + 2 <= P.n)) {
                    *reinterpret_cast<float2*>(smp + 4 * c) = fetch(s);
                    *reinterpret_cast<float2*>(smp + 4 * c + 2) = fetch(s + 1);
                }
            }
        }
    } else if (P.vec_ok) {
        // The FM chain's first stage rotates every input sample on the way in (freq_xlating_fir_filter with
        // taps = [1] ahead of the resampler), so its tile goes through registers: 16-B loads, MFMA_ROT_K in flight
        // per lane before the first is used (one load -> rotate -> store per round made the staging a chain of
        // ~17 global-load latencies per workgroup: 0.33 ms for 2^26 samples), NCO phase by 64-bit adds
        // (phase = index * inc mod 2^64 exactly), the padded LDS position by a stepped quotient.
        const int per8 = 8 * D;                            // source chunks per pad block
        const bool interior = lo_al >= 0 && lo_al + 2LL * nsrc <= P.n;   // uniform
        unsigned long long ph = (P.pre_idx0 + (unsigned long long)(lo_al + 2LL * t)) * P.pre_inc;
        const unsigned long long dph = P.pre_inc * (unsigned long long)(2 * T);
        int quo = t / per8, rem = t - quo * per8;
        const int dq = T / per8, dr = T - dq * per8;
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + lo_al);
        for (int c0 = 0; c0 < nsrc; c0 += MFMA_ROT_K * T) {
            float4 rg[MFMA_ROT_K];
#pragma unroll
            for (int k = 0; k < MFMA_ROT_K; ++k) {
                const int sc = c0 + k * T + t;
                if (interior)
                    rg[k] = x4[sc < nsrc ? sc : nsrc - 1];   // unconditional (clamped): the loads stay back to back
                else {
                    rg[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (sc < nsrc) {
                        const float2 a = fetch(lo_al + 2LL * sc), b = fetch(lo_al + 2LL * sc + 1);
                        rg[k] = make_float4(a.x, a.y, b.x, b.y);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < MFMA_ROT_K; ++k) {
                const int sc = c0 + k * T + t;
                if (sc < nsrc) {
                    const float2 r0 = nco_eval(P.pre_tab, ph), r1 = nco_eval(P.pre_tab, ph + P.pre_inc);
                    float4 v;
                    v.x = __builtin_fmaf(rg[k].x, r0.x, -(rg[k].y * r0.y));
                    v.y = __builtin_fmaf(rg[k].x, r0.y, rg[k].y * r0.x);
                    v.z = __builtin_fmaf(rg[k].z, r1.x, -(rg[k].w * r1.y));
                    v.w = __builtin_fmaf(rg[k].z, r1.y, rg[k].w * r1.x);
                    *reinterpret_cast<float4*>(smp + 4 * (sc + quo)) = v;
                }
                ph += dph;
                quo += dq;
                rem += dr;
                if (rem >= per8) {
                    rem -= per8;
                    ++quo;
                }
            }
        }
    } else {
        // Register staging: unaligned buffers, and the FM chain's first stage, which rotates every input
        // sample on the way in (freq_xlating_fir_filter with taps = [1] ahead of the resampler).  Per
        // sample: one unguarded 8-B load on interior tiles (uniform test), the NCO phase by a 64-bit ADD
        // (phase = index * inc mod 2^64 exactly: one multiply per thread, then + inc*T per round), the
        // pad position p / BK by a stepped quotient instead of a division per sample.
        odd = 0;
        const bool rot = P.pre_tab != nullptr;
        unsigned long long ph = rot ? (P.pre_idx0 + (unsigned long long)(lo_idx + t)) * P.pre_inc : 0ull;
        const unsigned long long dph = P.pre_inc * (unsigned long long)T;
        int quo = t / BK, rem = t - quo * BK;
        const int dq = T / BK, dr = T - dq * BK;
        const bool interior = lo_idx >= 0 && lo_idx + npos <= P.n;   // uniform
        for (int p = t; p < npos; p += T) {
            float2 v = interior ? x[lo_idx + p] : fetch(lo_idx + p);
            if (rot) {
                const float2 r = nco_eval(P.pre_tab, ph);
                v = make_float2(__builtin_fmaf(v.x, r.x, -(v.y * r.y)), __builtin_fmaf(v.x, r.y, v.y * r.x));
                ph += dph;
            }
            *reinterpret_cast<float2*>(smp + 2 * p + 4 * quo) = v;
            quo += dq;
            rem += dr;
            if (rem >= BK) {
                rem -= BK;
                ++quo;
            }
        }
    }
    __syncthreads();

    // ---- banded Toeplitz product on the f32 matrix cores
    f32x4 accR = {0.f, 0.f, 0.f, 0.f}, accI = {0.f, 0.f, 0.f, 0.f};
    mfma_kloop(ldsf, smp, D, ksteps, odd, lane, wave, accR, accI);
    if (__ballot(acc_nonfinite(accR, accI)) != 0ull) mfma_exact_redo(ldsf, smp, D, ntaps, odd, lane, wave, accR, accI);
    if (P.demod == SDR_DEMOD_NONE && !P.nco_tab) {
        // complex output with nothing behind the filter (FM chain's resampler stage, squelch input): straight from
        // the accumulators, D[row = 4*(lane>>4) + r][col = lane&15] = four consecutive outputs per lane (32 B) --
        // no pass through LDS, no barriers
        if (P.tap) {
            const long long m = m_t0 + wave * 256 + 16 * (lane & 15) + 4 * (lane >> 4);
            float2* const o = reinterpret_cast<float2*>(P.tap + (long long)cap * P.tap_stride) + m;
            if (m + 4 <= P.nout && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
                reinterpret_cast<float4*>(o)[0] = make_float4(accR[0], accI[0], accR[1], accI[1]);
                reinterpret_cast<float4*>(o)[1] = make_float4(accR[2], accI[2], accR[3], accI[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < P.nout) o[r] = make_float2(accR[r], accI[r]);
            }
        }
        return;
    }
    __syncthreads();  // every wave is done with the samples: the output tile aliases them
    float2* const ybuf = reinterpret_cast<float2*>(smp);
    {
        // D[row = 4*(lane>>4) + r][col = lane&15]: output 16*g + row of this wave's 256
        const int base = wave * 256 + 16 * (lane & 15) + 4 * (lane >> 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) ybuf[base + r] = make_float2(accR[r], accI[r]);
    }
    __syncthreads();

    // ---- epilogue: thread t owns outputs 4t .. 4t+3 of the tile
    float ar[4], ai[4];
    bool valid[4];
    const long long m0 = m_t0 + 4LL * t;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long m = m0 + e;
        const bool have_y = (m >= 0 || P.hist != nullptr) && (m < P.nout);
        float2 y = ybuf[4 * t + e];
        if (!have_y) y = make_float2(0.f, 0.f);
        if (have_y && P.nco_tab) {
            const float2 r = nco_eval(P.nco_tab, (P.nco_m0 + (unsigned long long)m) * P.nco_inc);
            y = make_float2(__builtin_fmaf(y.x, r.x, -(y.y * r.y)), __builtin_fmaf(y.x, r.y, y.y * r.x));
        }
        ar[e] = y.x;
        ai[e] = y.y;
        valid[e] = (4 * t + e >= q) && (m < P.nout);
    }
    if (P.demod == SDR_DEMOD_NONE) {
        if (P.tap) {
            float2* dst = reinterpret_cast<float2*>(P.tap + (long long)cap * P.tap_stride);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (valid[e]) dst[m0 + e] = make_float2(ar[e], ai[e]);
        }
        return;
    }
    float v[4];
    if (P.demod == SDR_DEMOD_MAG2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(ar[e], ar[e], ai[e] * ai[e]);
    } else {
        // the previous output of e = 0 belongs to the neighbouring thread: pass the
        // (rotated, zero-filled) outputs through LDS once more
        __syncthreads();
        ybuf[4 * t + 3] = make_float2(ar[3], ai[3]);
        __syncthreads();
        const float2 pv0 = t ? ybuf[4 * t - 1] : make_float2(0.f, 0.f);   // thread 0's e = 0 is never valid (q = 1)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // every lane and output (no exec-masked block per output); valid[] drops what is not emitted
            const float2 pv = (e == 0) ? pv0 : make_float2(ar[e - 1], ai[e - 1]);
            const float tr = __builtin_fmaf(ar[e], pv.x, ai[e] * pv.y);
            const float ti = __builtin_fmaf(ai[e], pv.x, -(ar[e] * pv.y));
            const float a = fast_atan2f_dev(s_atan, ti, tr);
            v[e] = valid[e] ? a : 0.f;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = P.gain * v[e];
    if (P.tap) {
        float* dst = P.tap + (long long)cap * P.tap_stride;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (valid[e]) dst[m0 + e] = v[e];
    }
    if (!P.do_slice) return;
    slice_tile_xn<4>(valid, v, 4 * t, m_t0, P.lo, P.hi, P.keep_n, P.keep_c0, P.bits + (long long)cap * P.bits_stride,
                  P.tsum + (long long)cap * P.tiles_per_capture + tile, sh);
}

// ---------------------------------------------------------------------------------------------------
// The same product for a stage WITHOUT demodulation (complex output: the FM chain's freq_xlating ->
// rational_resampler_ccc stage, a prefilter stage) with TWO TEAMS of four wavefronts per workgroup.
// k_frontend_mfma's waves all walk the same phases -- stage (wait on HBM, rotate), K loop, epilogue -- and the
// workgroups of a CU start together and take equally long, so they stay in step: the four waves of a SIMD
// queue for the matrix pipe during the K loop and leave it idle during the rest (SQ_VALU_MFMA_BUSY 44 % of the
// CU-busy cycles at 131 taps / 4; a wave spends 12 % of its life in MFMAs, 13 % in vector instructions and
// the rest waiting).  Here the two teams of a workgroup run in ANTIPHASE, held there by one barrier per phase:
//
//   phase      0        1        2        3
//   team 0   commit   K loop   commit   K loop     commit = rotate the tile's samples (registers -> LDS) and
//   team 1     -      commit   K loop   commit              issue the loads of the team's next tile
//
// Wave w of team 0 and wave w of team 1 share a SIMD (8 waves over 4 SIMDs), so each SIMD always has one
// wave feeding the matrix pipe while the other does the vector and memory work.  Each team owns one sample
// buffer (its K loop and its next commit are a barrier apart) and keeps its next tile's 16-B chunks in
// registers while it runs the K loop, so the HBM latency is hidden too.  Outputs go straight from the
// accumulators to memory.  Same products in the same order as k_frontend_mfma: bit-identical.
// P.tile_out = 1024; blockDim.x = 512; grid.x = ceil(tiles / tpw); dynamic LDS mfma2_lds_bytes().
#define MFMA2_NCH 9   // 16-B chunks per lane per tile (host: the tile's chunks must fit, D <= 4 for ~130 taps)
__host__ __device__ inline int mfma2_npos(int D, int ntaps) { return 1023 * D + ntaps + 40; }
__host__ __device__ inline int mfma2_buf_floats(int D, int ntaps)
{
    const int npos = mfma2_npos(D, ntaps) + 2;
    return (2 * npos + 4 * (npos / (16 * D) + 2) + 3) & ~3;
}
__host__ __device__ inline int mfma2_tap_floats(int D, int ntaps)
{
    const int ksteps = (15 * D + ntaps + 3) >> 2;
    return (15 * D + 4 * (ksteps + 8) + 3) & ~3;
}
__host__ __device__ inline size_t mfma2_lds_bytes(int D, int ntaps)
{
    return ((size_t)mfma2_tap_floats(D, ntaps) + 2 * (size_t)mfma2_buf_floats(D, ntaps)) * sizeof(float) +
           SDR_NCO_TAB * sizeof(float2);   // + the NCO table
}

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) k_frontend_mfma2(FrontParams P, int tpw)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* const ldsf = reinterpret_cast<float*>(smem_raw);
    constexpr int T = 256, TB = 1024, NCH = MFMA2_NCH;
    const int team = threadIdx.x >> 8, t = threadIdx.x & 255, lane = t & 63, wave = t >> 6;
    const int cap = blockIdx.y;
    const int D = P.D, ntaps = P.ntaps;
    const int ksteps = (15 * D + ntaps + 3) >> 2;
    const int ntl = mfma2_tap_floats(D, ntaps);
    const int npos = mfma2_npos(D, ntaps);
    const int per8 = 8 * D;                                // source chunks per pad block
    float* const smp = ldsf + ntl + team * mfma2_buf_floats(D, ntaps);
    const float2* __restrict__ x = P.x + (long long)cap * P.stride;
    const int tile0 = blockIdx.x * tpw;
    const int ntile = (P.tiles_per_capture - tile0 < tpw) ? P.tiles_per_capture - tile0 : tpw;
    const int n_my = (ntile - team + 1) >> 1;              // this team's tiles: tile0 + team + 2j

    for (int i = threadIdx.x; i < ntl; i += 2 * T) {
        const int idx = i - 15 * D;
        ldsf[i] = (idx >= 0 && idx < ntaps) ? P.rtaps[idx] : 0.f;
    }
    // the 1024-entry NCO table in LDS: one lookup per input sample, scattered over the 8 KB table -- through the
    // L1 every lane of such a gather is a cache line of its own, 64 cycles of the CU's one texture path per wave
    // instruction: 0.13 ms per 2^26 samples, more than the whole K loop
    float2* const s_nco = reinterpret_cast<float2*>(ldsf + ntl + 2 * mfma2_buf_floats(D, ntaps));
    if (P.pre_tab)
        for (int i = threadIdx.x; i < SDR_NCO_TAB; i += 2 * T) s_nco[i] = P.pre_tab[i];
    auto fetch = [&](long long idx) -> float2 {
        if (idx >= 0) return idx < P.n ? x[idx] : make_float2(0.f, 0.f);
        if (P.hist && idx >= -(long long)P.hist_len) return P.hist[P.hist_len + idx];
        return make_float2(0.f, 0.f);
    };
    // TB*D is even: every tile of the capture starts with the same parity
    const int odd = P.vec_ok ? (int)(((long long)cap * P.stride + P.in0 - (ntaps - 1)) & 1) : 0;
    const int nsrc = (npos + odd + 1) >> 1;                // 16-B source chunks (two samples) per tile
    float4 rg[NCH];                                        // lane t holds chunks t, t + T, ... of the team's next tile
    auto tile_lo = [&](int tile) -> long long { return P.in0 + (long long)tile * TB * D - (ntaps - 1) - odd; };
    auto is_interior = [&](long long lo_al) -> bool { return P.vec_ok && lo_al >= 0 && lo_al + 2LL * nsrc <= P.n; };   // uniform
    auto issue = [&](int tile) {
        const long long lo_al = tile_lo(tile);
        if (!is_interior(lo_al)) return;                   // edge tiles are fetched, guarded, when they are committed
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + lo_al);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            // unconditional (index clamped): a load under `if (c < nsrc)` becomes an exec-masked block of its
            // own and the compiler waits for the previous load before each -- nine HBM round trips per tile
            const int c = k * T + t;
            rg[k] = x4[c < nsrc ? c : nsrc - 1];
        }
    };
    // Two samples, phases ph and ph + inc: nco_eval twice and the complex product twice, with the same IEEE
    // operations in the same order, but packed -- the polynomial for both phases at once, and every
    // (a*b - c*d, a*d + c*b) pair as v_pk_mul + v_pk_fma with the halves picked by op_sel and the sign by neg_lo
    // (29 vector instructions per chunk instead of ~50: they add to the MFMA time one for one).
    auto rotate2 = [&](float4 u, unsigned long long ph) -> float4 {
        typedef float v2 __attribute__((ext_vector_type(2)));
        const unsigned long long ph1 = ph + P.pre_inc;
        const unsigned int ua = (unsigned int)(ph >> 32), ub = (unsigned int)(ph1 >> 32);
        const v2 r = (v2){(float)(ua & 0x3fffffu), (float)(ub & 0x3fffffu)} * (v2){1.46291807926715968105e-9f, 1.46291807926715968105e-9f};
        const v2 r2 = r * r;
        const v2 one = (v2){1.0f, 1.0f};
        const v2 cr = __builtin_elementwise_fma(r2, __builtin_elementwise_fma(r2, (v2){4.16666666666666666667e-2f, 4.16666666666666666667e-2f}, (v2){-0.5f, -0.5f}), one);
        const v2 sr = r * __builtin_elementwise_fma(r2, __builtin_elementwise_fma(r2, (v2){8.33333333333333333333e-3f, 8.33333333333333333333e-3f}, (v2){-1.66666666666666666667e-1f, -1.66666666666666666667e-1f}), one);
        const float2 ta_ = s_nco[ua >> 22], tb_ = s_nco[ub >> 22];
        const v2 ta = (v2){ta_.x, ta_.y}, tb = (v2){tb_.x, tb_.y};
        v2 pa, pb, oa, ob;
        // o = (t.x*cr - t.y*sr, t.y*cr + t.x*sr): p = (t.y*s, t.x*s); o = t*c + (-p.lo, p.hi)      [s, c: lo = phase a, hi = phase b]
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(pa) : "v"(ta), "v"(sr));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(pb) : "v"(tb), "v"(sr));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(oa) : "v"(ta), "v"(cr), "v"(pa));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1]" : "=v"(ob) : "v"(tb), "v"(cr), "v"(pb));
        // v' = (v.x*o.x - v.y*o.y, v.x*o.y + v.y*o.x): q = (v.y*o.y, v.y*o.x); v' = v.x*o + (-q.lo, q.hi)
        const v2 va = (v2){u.x, u.y}, vb = (v2){u.z, u.w};
        v2 qa, qb, wa, wb;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(qa) : "v"(va), "v"(oa));
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(qb) : "v"(vb), "v"(ob));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(wa) : "v"(va), "v"(oa), "v"(qa));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(wb) : "v"(vb), "v"(ob), "v"(qb));
        return make_float4(wa.x, wa.y, wb.x, wb.y);
    };
    auto commit = [&](int tile) {
        const long long lo_al = tile_lo(tile);
        unsigned long long ph = 0ull, dph = 0ull;
        if (P.pre_tab) {   // freq_xlating_fir_filter with taps = [1] ahead of the resampler: rotate on the way in
            ph = (P.pre_idx0 + (unsigned long long)(lo_al + 2LL * t)) * P.pre_inc;
            dph = P.pre_inc * (unsigned long long)(2 * T);
        }
        int quo = t / per8, rem = t - quo * per8;
        const int dq = T / per8, dr = T - dq * per8;
        if (!is_interior(lo_al)) {
            // first / last tiles of a capture (history, zero padding) and unaligned buffers: chunk by chunk, rolled
#pragma unroll 1
            for (int c = t; c < nsrc; c += T) {
                const float2 a = fetch(lo_al + 2LL * c), b = fetch(lo_al + 2LL * c + 1);
                float4 v = make_float4(a.x, a.y, b.x, b.y);
                if (P.pre_tab) v = rotate2(v, ph);
                *reinterpret_cast<float4*>(smp + 4 * (c + quo)) = v;
                ph += dph;
                quo += dq;
                rem += dr;
                if (rem >= per8) {
                    rem -= per8;
                    ++quo;
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = k * T + t;
            if (c < nsrc) {
                float4 v = rg[k];
                if (P.pre_tab) v = rotate2(v, ph);
                *reinterpret_cast<float4*>(smp + 4 * (c + quo)) = v;
            }
            ph += dph;
            quo += dq;
            rem += dr;
            if (rem >= per8) {
                rem -= per8;
                ++quo;
            }
        }
    };

    float2* const dst = reinterpret_cast<float2*>(P.tap + (long long)cap * P.tap_stride);
    // the NCO table is read by team 0's first commit, in phase 0: every wave's share of it (and of the taps) has
    // to be in LDS before that.  (Without this barrier a wave that fell behind in the fill left stale entries to
    // the first tile of its workgroup: about one capture in twenty had a few thousand wrong outputs.)
    lds_only_barrier();
    if (n_my > 0) issue(tile0 + team);
    const int nphase = ntile + 1;                          // team 0: phases 0 .. 2*n0-1, team 1: 1 .. 2*n1
    for (int p = 0; p < nphase; ++p) {
        const int s = p - team;                            // this team's step: even = commit, odd = K loop
        if (s >= 0 && (s >> 1) < n_my) {
            const int j = s >> 1, tile = tile0 + team + 2 * j;
            if (!(s & 1)) {
                commit(tile);
                if (j + 1 < n_my) issue(tile + 2);
            } else {
                f32x4 accR = {0.f, 0.f, 0.f, 0.f}, accI = {0.f, 0.f, 0.f, 0.f};
                mfma_kloop(ldsf, smp, D, ksteps, odd, lane, wave, accR, accI);
                if (__ballot(acc_nonfinite(accR, accI)) != 0ull) mfma_exact_redo(ldsf, smp, D, ntaps, odd, lane, wave, accR, accI);
                // D[row = 4*(lane>>4) + r][col = lane&15]: outputs 16*col + row of this wave's 256 -- four in a row per lane
                const long long m = (long long)tile * TB + wave * 256 + 16 * (lane & 15) + 4 * (lane >> 4);
                float2* const o = dst + m;
                if (m + 4 <= P.nout && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
                    reinterpret_cast<float4*>(o)[0] = make_float4(accR[0], accI[0], accR[1], accI[1]);
                    reinterpret_cast<float4*>(o)[1] = make_float4(accR[2], accI[2], accR[3], accI[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (m + r < P.nout) o[r] = make_float2(accR[r], accI[r]);
                }
            }
        }
        // The team that committed may now run its K loop, the one that ran it may overwrite its buffer.  NOT
        // __syncthreads(): its fence also waits for the global loads just issued and the outputs just stored
        // (vmcnt(0)), which put one HBM round trip into every phase.  Only LDS traffic has to be complete.
        lds_only_barrier();
    }
}
