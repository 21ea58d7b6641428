// Polyphase channelizer, 64 sub-bands, critically sampled (BASELINE.json
// configs[4]): y_c[m] = sum_k taps[k] e^{+j 2 pi c k/64} x[64 m - k], i.e. what a
// bank of 64 freq_xlating_fir_filter_ccc(decim 64, taps, c*fs/64) blocks yields
// [GR-ext], computed as 64 polyphase branch sums + a 64-point DFT, with mag^2 /
// gain / threshold_ff / slicer / keep_one_in_n fused behind it.  HBM-bound: 8 B
// read per input sample, 1/keep_n B written (all 64 channels together produce
// one output per input sample).  No MFMA: the 64-point DFT is a 4 x 16 FFT in
// registers (30 flop per input sample), not a dense contraction.
//
// Numeric contract = oracle/dsp_oracle.c orc_pfb_channelize, op for op.
//
// Work split: a 256-thread workgroup owns 64 consecutive output times of one
// capture; its (P+63)*64 input samples are staged once in LDS by 16-B DMA.
// Wave w owns times 16w..16w+15 and, once the samples are consumed, a private
// 8.5 KB slab of the same LDS for three in-wave transposes:
//   phase 1  lane = branch p: 16 complex accumulators, the lane's column walked once
//            (conflict-free ds_read_b64, every sample reused for up to P outputs);
//   T1       V[time][p] -> quad layout: 4 lanes per output time, lane r of the quad
//            takes b = r mod 4;
//   phase 2a radix-4 butterflies over a (p = 16a + b) + twiddles, all in registers;
//   T2       U[time][e][b] (bank-swizzled) -> lane e of the quad takes all 16 b;
//   phase 2b 16-point FFT in registers (no cross-lane traffic, every butterfly
//            output is used);
//   T3       Y[channel][time] -> lane = channel c = e + 4d;
//   phase 3  hysteresis over the lane's 16 outputs in registers, carry across the
//            four waves through LDS, one TileSum per (channel, workgroup).
#pragma once
#include "sdr_common.h"
#include "frontend_kernels.h"

#define PFB_M 64
#define PFB_TM 16    // output times per wave
#ifndef PFB_NW
#define PFB_NW 4     // waves per workgroup
#endif
#define PFB_THREADS (64 * PFB_NW)
#define PFB_TILE (PFB_TM * PFB_NW)
#define PFB_VP 68     // row pitch (float2) of the V/U transposes: 16 rows per wave
#define PFB_YP 17     // row pitch of the Y transpose: 64 rows per wave
#define PFB_SLAB (PFB_TM * PFB_VP)   // == PFB_M * PFB_YP float2 per wave

struct PfbParams {
    const float2* x;        // capture 0, sample 0
    const float2* hist;     // streaming: hist_len samples that precede x[0]; nullptr = zeros (stream start)
    int hist_len;
    long long in0;          // input index (relative to x[0]) of output 0's newest sample; 0 for whole captures
    long long stride;       // samples between captures
    long long n;            // samples per capture
    long long nout;         // outputs per channel (n / 64)
    int vec_ok;
    const float* taps;      // prototype taps in natural order, zero-padded to P*64
    const float2* tw;       // 64 twiddles e^{+j 2 pi k/64}
    int demod;              // SDR_DEMOD_NONE | SDR_DEMOD_MAG2
    float gain, lo, hi;
    int keep_n;
    long long keep_c0;
    int8_t* bits;           // row = capture*64 + channel
    long long bits_stride;
    float* tap;
    long long tap_stride;
    TileSum* tsum;
    int tiles_per_row;
};

template <int P>
__global__ void __launch_bounds__(PFB_THREADS) k_pfb(PfbParams Q)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    __shared__ int sh_first[PFB_NW][PFB_M];   // first event of the wave's 16 outputs (0..15) or -1
    __shared__ int sh_last[PFB_NW][PFB_M];

    // XCD-aware tile order: workgroups go round-robin over the 8 XCDs (each with its
    // own L2), so blockIdx.x = 8*idx + k is mapped to the idx-th tile of the k-th
    // contiguous eighth of the capture.  Neighbouring tiles then share an L2: the
    // (P-1)*64-sample halo is an L2 hit and the 4 bytes a tile adds to each channel
    // row merge into full lines before they leave the L2.
    const int cap = blockIdx.y;
    int tile;
    {
        const int nt = gridDim.x, k = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int base = nt >> 3, rem = nt & 7;
        tile = k * base + (k < rem ? k : rem) + idx;
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: guards on it are scalar branches
    const long long m0 = (long long)tile * PFB_TILE;
    const float2* __restrict__ x = Q.x + (long long)cap * Q.stride;
    const long long lo_idx = Q.in0 + (m0 - (P - 1)) * PFB_M - (PFB_M - 1);
    constexpr int count = (P + PFB_TILE - 1) * PFB_M;

    auto fetch = [&](long long idx) -> float2 {
        if (idx >= 0) return idx < Q.n ? x[idx] : make_float2(0.f, 0.f);
        if (Q.hist && idx >= -(long long)Q.hist_len) return Q.hist[Q.hist_len + idx];
        return make_float2(0.f, 0.f);
    };

    // ---- stage the tile (same scheme as k_frontend) -------------------------------
    int lsh = 0;
    if (Q.vec_ok) {
        const long long gidx = (long long)cap * Q.stride + lo_idx;
        const int odd = (int)(gidx & 1);
        const long long lo_al = lo_idx - odd;
        const int pairs = (count + odd + 1) >> 1;
        lsh = odd;
        // (two loops: the DMA loop holds no other memory operation, see k_frontend)
        // chunk p holds samples lo_al + 2p, +1: inside the capture for p in [plo, phi) -- uniform bounds, so a lane
        // tests two 32-bit numbers and adds a 32-bit offset to a uniform base (as k_frontend_win)
        long long plo64 = lo_al < 0 ? (1 - lo_al) >> 1 : 0, phi64 = (Q.n - lo_al) >> 1;
        if (plo64 > pairs) plo64 = pairs;
        if (phi64 > pairs) phi64 = pairs;
        if (phi64 < plo64) phi64 = plo64;
        const int plo = (int)plo64, phi = (int)phi64;
        // scalar loop counter and LDS base (w is wave-uniform), one 64-bit pointer per lane bumped by a constant
        const char* __restrict__ gp = reinterpret_cast<const char*>(x + lo_al) + 16u * (unsigned)(64 * w + lane);
        int p0 = 64 * w;
        if (plo == 0 && phi == pairs)   // interior tile (uniform): whole waves of chunks need no lane mask
            for (; p0 + 64 <= pairs; p0 += PFB_THREADS, gp += 16 * PFB_THREADS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(lds + 2 * p0), 16, 0, SDR_GLDS_AUX);
        for (; p0 < pairs; p0 += PFB_THREADS, gp += 16 * PFB_THREADS) {
            const int p = p0 + lane;
            if (p >= plo && p < phi)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(lds + 2 * p0), 16, 0, SDR_GLDS_AUX);
        }
        if (lo_al < 0 || lo_al + 2LL * pairs > Q.n) {   // uniform: only edge tiles have guarded chunks
            for (int p = threadIdx.x; p < pairs; p += PFB_THREADS) {
                const long long s = lo_al + 2LL * p;
                if (!(s >= 0 && s + 2 <= Q.n)) {
                    lds[2 * p] = fetch(s);
                    lds[2 * p + 1] = fetch(s + 1);
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < count; i += PFB_THREADS) lds[i] = fetch(lo_idx + i);
    }

    // branch taps of this lane, twiddles of this lane for the six stages
    float hq[P];
#pragma unroll
    for (int q = 0; q < P; ++q) hq[q] = Q.taps[lane + PFB_M * q];
    typedef float f2v __attribute__((ext_vector_type(2)));
    const int qi = lane >> 2, r = lane & 3;   // quad layout: output time 16w + qi, quarter r
    f2v twq[3][4];                            // W^(e*b), e = 1..3, b = 4j + r
#pragma unroll
    for (int e = 1; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 t = Q.tw[(e * (4 * j + r)) & 63];
            twq[e - 1][j] = f2v{t.x, t.y};
        }
    __syncthreads();

    // ---- phase 1: polyphase branch sums, lane = p ------------------------------------
    // (re, im) pairs as 2-vectors: the fma / add / mul below are v_pk_*_f32, one
    // instruction for both components, same IEEE result per component.
    f2v acc[PFB_TM];
#pragma unroll
    for (int i = 0; i < PFB_TM; ++i) acc[i] = f2v{0.f, 0.f};
    {
        const f2v* col = reinterpret_cast<const f2v*>(lds + lsh + (long long)(PFB_TM * w) * PFB_M + (PFB_M - 1) - lane);
#pragma unroll
        for (int bb = 0; bb < PFB_TM + P - 1; ++bb) {
            const f2v s = col[bb * PFB_M];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int i = bb - (P - 1) + q;   // compile-time after unrolling
                if (i >= 0 && i < PFB_TM) acc[i] = __builtin_elementwise_fma(f2v{hq[q], hq[q]}, s, acc[i]);
            }
        }
    }
    auto cmul = [](f2v d, f2v wv) -> f2v {   // (dr*wr - di*wi, dr*wi + di*wr), the oracle's orc_cmul
        const f2v t = f2v{d.y, d.y} * f2v{-wv.y, wv.x};
        return __builtin_elementwise_fma(f2v{d.x, d.x}, wv, t);
    };
    __syncthreads();   // every wave is done with the samples: their LDS is reused below
    f2v* slab = reinterpret_cast<f2v*>(lds) + w * PFB_SLAB;

    // ---- T1: V[time][p] ------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < PFB_TM; ++i) slab[i * PFB_VP + lane] = acc[i];
    __syncthreads();

    // ---- phase 2a: radix-4 over a (p = 16a + b) + twiddle, b = 4j + r -----------------------
    f2v* vrow = slab + qi * PFB_VP;
    f2v z[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = 4 * j + r;
        const f2v v0 = vrow[b], v1 = vrow[16 + b], v2 = vrow[32 + b], v3 = vrow[48 + b];
        const f2v s02 = v0 + v2, d02 = v0 - v2, s13 = v1 + v3, d13 = v1 - v3;
        z[0][j] = s02 + s13;
        const f2v u1 = f2v{d02.x - d13.y, d02.y + d13.x};
        const f2v u2 = s02 - s13;
        const f2v u3 = f2v{d02.x + d13.y, d02.y - d13.x};
        z[1][j] = cmul(u1, twq[0][j]);
        z[2][j] = cmul(u2, twq[1][j]);
        z[3][j] = cmul(u3, twq[2][j]);
    }
    // ---- T2: U[time][e][b]; a lane reads and writes only positions = r mod 4 of its row,
    // so no barrier is needed between the reads above and these writes
    // Row layout pos(e, b) = 16e + 4(b>>2) + ((b + e) & 3): a Latin square in (e, b mod 4),
    // so both the writes (lanes differ in r = b mod 4) and the reads below (lanes differ
    // in e) spread over the banks; the plain 16e + b layout made the reads 4-way conflicts.
    // All T1 reads of the wave precede these writes in program order.
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) vrow[16 * e + 4 * j + ((r + e) & 3)] = z[e][j];
    __syncthreads();

    // ---- phase 2b: 16-point DFT over b in registers, lane r of the quad = e -------------------
    f2v t[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) t[b] = vrow[16 * r + 4 * (b >> 2) + ((b + r) & 3)];
#pragma unroll
    for (int s = 3; s >= 0; --s) {
        const int half = 1 << s;
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            if (l & half) continue;
            const f2v a = t[l], bb = t[l + half];
            t[l] = a + bb;
            const f2v d = a - bb;
            const int k = (l & (half - 1)) * (8 / half);
            if (k == 0)
                t[l + half] = d;
            else if (k == 4)
                t[l + half] = f2v{-d.y, d.x};
            else {
                const float2 wv = Q.tw[4 * k];
                t[l + half] = cmul(d, f2v{wv.x, wv.y});
            }
        }
    }
    __syncthreads();   // the wave has read its U rows: the slab becomes Y[channel][time]
    // ---- T3: position pos holds d = bitrev4(pos); channel c = e + 4d ----------------------------
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        const int d = ((pos & 1) << 3) | ((pos & 2) << 1) | ((pos & 4) >> 1) | ((pos & 8) >> 3);
        slab[(r + 4 * d) * PFB_YP + qi] = t[pos];
    }
    __syncthreads();
    float ar[PFB_TM], ai[PFB_TM];
#pragma unroll
    for (int i = 0; i < PFB_TM; ++i) {
        const f2v y = slab[lane * PFB_YP + i];
        ar[i] = y.x;
        ai[i] = y.y;
    }

    // ---- phase 3: lane = channel ------------------------------------------------------------
    const int ch = lane;
    const long long row = (long long)cap * PFB_M + ch;
    const long long o_w = m0 + PFB_TM * w;   // first output index of this wave
    if (Q.demod == SDR_DEMOD_NONE) {
        if (Q.tap) {
            float2* dst = reinterpret_cast<float2*>(Q.tap + row * Q.tap_stride);
#pragma unroll
            for (int i = 0; i < PFB_TM; ++i)
                if (o_w + i < Q.nout) dst[o_w + i] = make_float2(ar[i], ai[i]);
        }
        return;
    }
    // threshold_ff over the lane's 16 outputs as an adder's carry chain (see slice_tile_hk in frontend_mfma.h):
    // bit i of H / K = output i sets / clears the state; outputs past the row's end carry no event (uniform mask)
    // (masks built from the last output to the first: the compare's lane bit enters as the carry of mask + mask,
    // two instructions per output and mask; "not above hi" is applied to the finished K)
    unsigned H = 0, K = 0;
#pragma unroll
    for (int i = PFB_TM - 1; i >= 0; --i) {
        const float v = Q.gain * __builtin_fmaf(ar[i], ar[i], ai[i] * ai[i]);
        ar[i] = v;
        asm("v_cmp_lt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(H) : "s"(Q.hi), "v"(v) : "vcc");
        asm("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(K) : "s"(Q.lo), "v"(v) : "vcc");
    }
    {
        const long long left = Q.nout - o_w;
        const unsigned vm = left >= PFB_TM ? 0xffffu : (left <= 0 ? 0u : ((1u << (int)left) - 1u));
        H &= vm;
        K &= vm & ~H;
    }
    const unsigned E = H | K;
    const int first = E ? __builtin_ctz(E) : -1;
    const int st = E ? (int)((H >> (31 - __builtin_clz(E))) & 1u) : 0;
    const unsigned X = (H | ~E) & 0xffffu;
    unsigned states = (((X + H) ^ X ^ H) >> 1) & 0xffffu;   // bit i = threshold_ff output at time i, carry-in 0
    if (Q.tap) {
        float* dst = Q.tap + row * Q.tap_stride;
#pragma unroll
        for (int i = 0; i < PFB_TM; ++i)
            if (o_w + i < Q.nout) dst[o_w + i] = ar[i];
    }
    sh_first[w][lane] = first;
    sh_last[w][lane] = st;
    __syncthreads();
    int carry = 0;   // state left by the earlier waves of this workgroup (tile carry-in taken as 0)
    for (int v = 0; v < w; ++v)
        if (sh_first[v][lane] >= 0) carry = sh_last[v][lane];
    if (carry) states |= (first < 0) ? 0xffffu : ((1u << first) - 1u);
    if (w == 0) {
        int tf = -1, tl = 0;
        for (int v = 0; v < PFB_NW; ++v)
            if (sh_first[v][lane] >= 0) {
                if (tf < 0) tf = PFB_TM * v + sh_first[v][lane];
                tl = sh_last[v][lane];
            }
        TileSum t;
        t.first_evt = tf < 0 ? -1 : (int32_t)(m0 + tf);
        t.last_state = tl;
        Q.tsum[row * Q.tiles_per_row + tile] = t;
    }
    int8_t* brow = Q.bits + row * Q.bits_stride;
    if (Q.keep_n == 1) {   // keep_c0 == 0: the wave's 16 bytes are one 16-B store when the row position allows
        if (o_w + PFB_TM <= Q.nout && ((reinterpret_cast<uintptr_t>(brow + o_w) & 15) == 0)) {
            uint4 pk;
            unsigned* pw = reinterpret_cast<unsigned*>(&pk);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pw[k] = (((states >> (4 * k)) & 0xfu) * 0x00204081u) & 0x01010101u;   // bit j of the nibble -> bit 8j
            }
            *reinterpret_cast<uint4*>(brow + o_w) = pk;
        } else {
            for (int i = 0; i < PFB_TM && o_w + i < Q.nout; ++i) brow[o_w + i] = (int8_t)((states >> i) & 1u);
        }
    } else if (o_w < Q.nout) {
        // emitted outputs are those with (o - keep_c0) % keep_n == 0: one division per wave tile
        // (o < 2^31: nout is bounded by the int32 TileSum index)
        const unsigned kn = (unsigned)Q.keep_n;
        long long o = o_w < Q.keep_c0 ? Q.keep_c0 : o_w;
        const unsigned r = (unsigned)(o - Q.keep_c0);
        unsigned q = r / kn;
        if (q * kn != r) {
            ++q;
            o = Q.keep_c0 + (long long)q * kn;
        }
        const long long o_end = (o_w + PFB_TM < Q.nout) ? o_w + PFB_TM : Q.nout;
        for (; o < o_end; o += kn, ++q) brow[q] = (int8_t)((states >> (int)(o - o_w)) & 1u);
    }
}
