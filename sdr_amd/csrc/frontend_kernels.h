// Front-end kernels: NCO + decimating FIR + demod + gain + hysteresis + slicer
// + keep_one_in_n fused in one pass over the IQ samples (SURVEY.md 8a rows
// a1, a2, a4, a5, a8).  HBM-bound for D >~ 16: 8 B read per input sample,
// 1/(D*keep_n) B written.  Arithmetic order is the oracle's (oracle/dsp_oracle.c
// header): one fmaf per tap and lane, oldest sample first.
#pragma once
#include "sdr_common.h"

// Cache policy of the staging DMA (aux operand of global_load_lds).  2 = nt: every input
// byte is read once, so it is streamed through the L2 instead of displacing it -- measured
// 0.530 -> 0.505 ms on the headline batch (sc0/sc1 on top make no difference).
#ifndef SDR_GLDS_AUX
#define SDR_GLDS_AUX 2
#endif

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// rtlsdr_source's byte -> float map, (u8 - 127.4) * (1/128) per component: both ops
// are exact in float32 (|u8 - 127.4f| < 128 needs no more than 24 bits; 2^-7 scale)
__device__ __forceinline__ float2 cu8_to_f2(unsigned char i, unsigned char q)
{
    return make_float2(((float)i - 127.4f) * 0.0078125f, ((float)q - 127.4f) * 0.0078125f);
}

// exact-phase NCO: see orc nco64 in oracle/dsp_oracle.c (same op sequence)
__device__ __forceinline__ float2 nco_eval(const float2* __restrict__ tab, unsigned long long ph)
{
    unsigned int u = (unsigned int)(ph >> 32);
    unsigned int i = u >> 22;
    float r = (float)(u & 0x3fffffu) * 1.46291807926715968105e-9f;
    float r2 = r * r;
    float cr = __builtin_fmaf(r2, __builtin_fmaf(r2, 4.16666666666666666667e-2f, -0.5f), 1.0f);
    float sr = r * __builtin_fmaf(r2, __builtin_fmaf(r2, 8.33333333333333333333e-3f, -1.66666666666666666667e-1f), 1.0f);
    float2 t = tab[i];
    float2 o;
    o.x = __builtin_fmaf(t.x, cr, -(t.y * sr));
    o.y = __builtin_fmaf(t.y, cr, t.x * sr);
    return o;
}

// gnuradio fast_atan2f restated (table 257 floats): the oracle's arithmetic (one division, the table
// lerp as sub / mul / add, one add or subtract of the octant constant), written without branches --
// the if/else ladder of the original compiled to ~30 basic blocks of exec masking per call.  The
// selects reproduce the ladder exactly: its eight outcomes are +-(c + s*base) with c in {0, pi, pi/2}
// and s = +-1, and IEEE negation is exact (a - b == -(b - a), c + (-b) == c - b; none of these sums
// can be zero since base <= ~pi/4 < c, so no signed-zero case arises; 0 + base == base).
__device__ __forceinline__ float fast_atan2f_dev(const float* __restrict__ tab, float y, float x)
{
    const float y_abs = __builtin_fabsf(y), x_abs = __builtin_fabsf(x);
    const bool ylt = y_abs < x_abs;
    const float z = (ylt ? y_abs : x_abs) / (ylt ? x_abs : y_abs);
    float alpha = z * 255.0f;
    const int index = ((int)alpha) & 0xff;
    alpha -= (float)index;
    const float t0 = tab[index], t1 = tab[index + 1];
    float base = t0 + (t1 - t0) * alpha;
    base = (z < 0.003921569f) ? z : base;
    const bool xdom = x_abs > y_abs, xpos = x >= 0.0f;
    const float c = xdom ? (xpos ? 0.0f : 3.14159265358979323846f) : 1.57079632679489661923f;
    const float sb = (xdom != xpos) ? -base : base;    // xdom: minus iff x < 0;  otherwise: minus iff x >= 0
    const float r = c + sb;
    const float angle = (y >= 0.0f) ? r : -r;
    return ((y_abs > 0.0f) || (x_abs > 0.0f)) ? angle : 0.0f;
}

// Hysteresis (threshold_ff) inside one tile as a scan over {set, clear, hold}
// events: wavefront ballots + a <=4-entry cross-wave table.  Outputs before the
// tile's first event take the carry-in, assumed 0 here and patched by the
// back-end kernel once the carry of every tile is known.  Then the slicer
// (state - 0.5 >= 0 == state) and keep_one_in_n.
__device__ __forceinline__ void slice_tile(bool valid, float v, long long o, long long o_base, float lo, float hi,
                                           int keep_n, long long keep_c0, int8_t* __restrict__ bits,
                                           TileSum* __restrict__ ts_out, int* __restrict__ sh /* 12 ints */)
{
    int evt = valid ? (v > hi ? 1 : (v < lo ? 2 : 0)) : 0;
    unsigned long long sset = __ballot(evt == 1);
    unsigned long long sclr = __ballot(evt == 2);
    unsigned long long any = sset | sclr;
    int lane = lane_id(), wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if (lane == 0) {
        sh[wave] = any != 0;
        sh[4 + wave] = any ? (int)((sset >> (63 - __builtin_clzll(any))) & 1) : 0;
        sh[8 + wave] = any ? (wave * 64 + __builtin_ctzll(any)) : -1;
    }
    __syncthreads();
    unsigned long long below = any & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    int st = 0;
    if (below) {
        int idx = 63 - __builtin_clzll(below);
        st = (int)((sset >> idx) & 1);
    } else {
        for (int w = wave - 1; w >= 0; --w)
            if (sh[w]) {
                st = sh[4 + w];
                break;
            }
    }
    if (threadIdx.x == 0) {
        int first = -1, last = 0;
        for (int w = 0; w < nw; ++w)
            if (sh[w]) {
                if (first < 0) first = sh[8 + w];
                last = sh[4 + w];
            }
        // first_evt as output index: thread index -> o of thread 0 + index
        ts_out->first_evt = first < 0 ? -1 : (int32_t)(o_base + first);
        ts_out->last_state = last;
    }
    if (valid && o >= keep_c0) {
        long long r = o - keep_c0;
        if (keep_n == 1)
            bits[r] = (int8_t)st;
        else if (r < 0x7fffffffLL) {  // 32-bit division: the 64-bit one costs ~10x
            const unsigned qq = (unsigned)r / (unsigned)keep_n;
            if ((unsigned)r == qq * (unsigned)keep_n) bits[qq] = (int8_t)st;
        } else if (r % keep_n == 0)
            bits[r / keep_n] = (int8_t)st;
    }
}

// Everything behind the channel filter for thread t's output o = (ar, ai): demodulator, gain,
// optional tap, hysteresis scan + slicer.  SYNC_TILE: ysh aliases the LDS tile, so every wave must
// have finished reading it before the discriminator exchange overwrites it.
// QD = false: the chain's demodulator is known not to be the discriminator, whose arctangent is then not compiled
// into the kernel at all (round 3: the hot complex64 kernel ran 3 % slower at decimation 10 after a round-2 change
// to fast_atan2f_dev -- code it never executes for a mag^2 chain -- moved its identical tap loop in memory).
template <bool SYNC_TILE, bool QD = true>
__device__ __forceinline__ void frontend_finish(const FrontParams& P, const int tile, const int cap, const int t,
                                                const int q, const long long o, const float ar, const float ai,
                                                float2* ysh, int* sh)
{
    bool valid = (t >= q) && (o < P.nout) && (t - q < P.tile_out);
    float v = 0.f;
    if (P.demod == SDR_DEMOD_NONE) {
        if (valid && P.tap)
            reinterpret_cast<float2*>(P.tap + (long long)cap * P.tap_stride)[o] = make_float2(ar, ai);
        return;
    } else if (P.demod == SDR_DEMOD_MAG2) {
        v = __builtin_fmaf(ar, ar, ai * ai);
    } else if (QD) {
        if (SYNC_TILE) __syncthreads();  // every wave has finished reading the tile
        ysh[t] = make_float2(ar, ai);
        __syncthreads();
        if (valid) {
            float2 pv = ysh[t - 1];
            float tr = __builtin_fmaf(ar, pv.x, ai * pv.y);
            float ti = __builtin_fmaf(ai, pv.x, -(ar * pv.y));
            v = fast_atan2f_dev(P.atan_tab, ti, tr);
        }
    }
    v = P.gain * v;
    if (valid && P.tap) (P.tap + (long long)cap * P.tap_stride)[o] = v;
    if (!P.do_slice) return;
    // slice_tile wants the output index of thread 0 (= o0 - q)
    slice_tile(valid, v, o, o - t, P.lo, P.hi, P.keep_n, P.keep_c0,
               P.bits + (long long)cap * P.bits_stride,
               P.tsum + (long long)cap * P.tiles_per_capture + tile, sh);
}

// One thread = one FIR output.  Tile = blockDim.x consecutive FIR outputs whose
// (T-1)*D + ntaps input samples are staged once in LDS (coalesced 8-B loads),
// then every thread walks its ntaps-window with ds_read_b64 + 2 (4) fmaf.
// QUAD demod needs y[m-1]: thread t computes y[o0-1+t] and emits for t>=1.
template <bool USE_LDS, bool CU8, bool QD = true>
__device__ __forceinline__ void frontend_tile(const FrontParams& P, const FrontKTaps& KT, const int tile, const int cap,
                                              float2* lds, int* sh, float2* ysh_static, const int staged_lsh = -1)
{
    float2* const ysh = USE_LDS ? lds : ysh_static;
    const int T = blockDim.x;
    const int q = (QD && P.demod == SDR_DEMOD_QUAD) ? 1 : 0;
    const long long o0 = (long long)tile * P.tile_out;
    const long long m_t0 = o0 - q;                       // FIR output index of thread 0
    const float2* __restrict__ x = P.x + (long long)cap * P.stride;
    const int H = P.ntaps - 1;
    const long long lo_idx = P.in0 + m_t0 * P.D - H;     // first input sample of the tile
    const int count = (T - 1) * P.D + P.ntaps;

    // rtl-sdr bytes: the same two exact float ops as the oracle's cu8_to_fc32
    const uchar2* __restrict__ xb = reinterpret_cast<const uchar2*>(P.x) + (long long)cap * P.stride;
    auto fetch = [&](long long idx) -> float2 {
        if (CU8) {
            uchar2 u;
            if (idx >= 0) {
                if (idx >= P.n) return make_float2(0.f, 0.f);
                u = xb[idx];
            } else if (P.hist && idx >= -(long long)P.hist_len)
                u = reinterpret_cast<const uchar2*>(P.hist)[P.hist_len + idx];
            else
                return make_float2(0.f, 0.f);
            return cu8_to_f2(u.x, u.y);
        }
        if (idx >= 0) return idx < P.n ? x[idx] : make_float2(0.f, 0.f);
        if (P.hist && idx >= -(long long)P.hist_len) return P.hist[P.hist_len + idx];
        return make_float2(0.f, 0.f);
    };

    int lsh = 0;  // LDS index of sample lo_idx
    if (USE_LDS && staged_lsh >= 0) {
        lsh = staged_lsh;   // the caller (k_frontend_pipe) has staged the tile and synchronised
    } else if (USE_LDS) {
        // Interior tiles: 16-B direct-to-LDS loads (global_load_lds_dwordx4), no
        // VGPR round trip, one 1-KiB piece per wave-instruction.  The tile start is
        // moved down to a 16-B boundary of the capture buffer.
        const long long gidx = (long long)cap * P.stride + lo_idx;
        const int odd = (int)(gidx & 1);
        const long long lo_al = lo_idx - odd;
        const int pairs = (count + odd + 1) >> 1;
        if (!CU8 && P.vec_ok && !P.pre_tab) {
            // Every 16-B chunk that lies inside the capture goes by DMA; only the few
            // chunks that touch the capture's edges (zero padding / streaming history)
            // are fetched by their lanes with guarded loads.  (First and last tile of a
            // capture used to take a fully guarded, serialised staging loop.)
            // Two loops on purpose: with the guarded fetches in the same loop body the
            // compiler may merge the paths and wait (vmcnt(0)) after every DMA, which
            // serialises the staging (seen: 0.46 -> 1.24 ms).  The DMA loop holds no
            // other memory operation.
            lsh = odd;
            // chunk p holds samples lo_al + 2p, +1: inside the capture for p in [plo, phi) -- uniform 32-bit bounds;
            // the loop counter and the LDS base of every DMA (M0) are scalars (the wave's first chunk read with
            // readfirstlane), a lane keeps one 64-bit pointer and bumps it: 4 vector instructions per DMA where
            // the per-lane 64-bit index arithmetic took 12 (25 DMAs per lane and tile at D = 50)
            long long plo64 = lo_al < 0 ? (1 - lo_al) >> 1 : 0, phi64 = (P.n - lo_al) >> 1;
            if (plo64 > pairs) plo64 = pairs;
            if (phi64 > pairs) phi64 = pairs;
            if (phi64 < plo64) phi64 = plo64;
            const int plo = (int)plo64, phi = (int)phi64;
            const int wb = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
            const char* __restrict__ gp = reinterpret_cast<const char*>(x + lo_al) + 16u * (unsigned)(wb + lane_id());
            const unsigned gstep = 16u * (unsigned)T;
            int p0 = wb;
            if (plo == 0 && phi == pairs) {
                // interior tile (uniform): whole waves of chunks without a lane mask -- M0, the DMA, one pointer
                // bump and the scalar loop; the ragged last chunk falls through to the masked loop
                for (; p0 + 64 <= pairs; p0 += T, gp += gstep)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)gp,
                        (__attribute__((address_space(3))) void*)(lds + 2 * p0), 16, 0, SDR_GLDS_AUX);
            }
            for (; p0 < pairs; p0 += T, gp += gstep) {
                const int p = p0 + lane_id();
                if (p >= plo && p < phi)
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void*)gp,
                        (__attribute__((address_space(3))) void*)(lds + 2 * p0), 16, 0, SDR_GLDS_AUX);
            }
            if (lo_al < 0 || lo_al + 2LL * pairs > P.n) {   // uniform: only edge tiles have guarded chunks
                for (int p = threadIdx.x; p < pairs; p += T) {
                    const long long s = lo_al + 2LL * p;
                    if (!(s >= 0 && s + 2 <= P.n)) {
                        lds[2 * p] = fetch(s);
                        lds[2 * p + 1] = fetch(s + 1);
                    }
                }
            }
        } else if (P.pre_tab) {
            // freq_xlating_fir_filter with taps=[1] in front of the FIR: rotate
            // each input sample as it is staged (same ops as the stand-alone block)
            for (int i = threadIdx.x; i < count; i += T) {
                const long long idx = lo_idx + i;
                float2 v = fetch(idx);
                float2 r = nco_eval(P.pre_tab, (P.pre_idx0 + (unsigned long long)idx) * P.pre_inc);
                lds[i] = make_float2(__builtin_fmaf(v.x, r.x, -(v.y * r.y)), __builtin_fmaf(v.x, r.y, v.y * r.x));
            }
        } else {
            for (int i = threadIdx.x; i < count; i += T) lds[i] = fetch(lo_idx + i);
        }
        __syncthreads();
    }

    const int t = threadIdx.x;
    const long long m = m_t0 + t;
    const long long o = m;          // demod output index == FIR output index (emitted for t>=q)
    bool have_y = (m >= 0 || P.hist != nullptr) && (m < P.nout);
    // m == -1 in streaming mode: previous call's last output, recomputed from history
    float ar = 0.f, ai = 0.f;
    if (USE_LDS) {
        // Taps live in VGPRs, one tap per lane per 64-tap chunk (loaded with one
        // coalesced load, the next chunk prefetched), and are broadcast with
        // v_readlane: no memory access for taps inside the FIR loop.  All lanes run
        // the loop (readlane needs the whole wave); lanes without an output are
        // zeroed afterwards.
        const float* __restrict__ rt = P.rtaps;
        const float2* p = lds + lsh + (long long)t * P.D;
        const int lane = lane_id();
        const bool cplx = P.taps_complex != 0;
        if (P.ktaps_n > 0) {
            // <= 64 taps: they live in the kernel-argument segment, the compiler fetches them
            // with s_load (uniform address, constant memory) straight into SGPR operands of
            // the fmas: no VGPR, no v_readlane, no LDS traffic for taps.
            const int nt = P.ntaps;
            int j = 0;
            if (cplx && !(P.D & 1)) {
                // Even decimation: every lane's window starts at the same parity, so the samples can be
                // read two at a time as 16-B words (after one single tap when the tile starts odd).
                // ds_read_b128 is served in groups of 16 lanes with 4-dword granules: a lane stride of
                // 2*D dwords with D = 2 (mod 4) -- 10, 50 -- is conflict-free there, while the 8-B
                // reads of a 32-lane group leave half the banks idle (2-way conflicts).
                if (lsh & 1) {
                    const float2 s0 = p[0];
                    const float a = KT.t[0].x, b = KT.t[0].y;
                    ar = __builtin_fmaf(a, s0.x, ar);
                    ar = __builtin_fmaf(-b, s0.y, ar);
                    ai = __builtin_fmaf(a, s0.y, ai);
                    ai = __builtin_fmaf(b, s0.x, ai);
                    j = 1;
                }
                for (; j + 8 <= nt; j += 8) {
                    const float4* p4 = reinterpret_cast<const float4*>(p + j);
                    float4 qv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) qv[u] = p4[u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float a0 = KT.t[j + 2 * u].x, b0 = KT.t[j + 2 * u].y;
                        ar = __builtin_fmaf(a0, qv[u].x, ar);
                        ar = __builtin_fmaf(-b0, qv[u].y, ar);
                        ai = __builtin_fmaf(a0, qv[u].y, ai);
                        ai = __builtin_fmaf(b0, qv[u].x, ai);
                        const float a1 = KT.t[j + 2 * u + 1].x, b1 = KT.t[j + 2 * u + 1].y;
                        ar = __builtin_fmaf(a1, qv[u].z, ar);
                        ar = __builtin_fmaf(-b1, qv[u].w, ar);
                        ai = __builtin_fmaf(a1, qv[u].w, ai);
                        ai = __builtin_fmaf(b1, qv[u].z, ai);
                    }
                }
            }
            if (cplx) {
                for (; j + 8 <= nt; j += 8) {
                    float2 s[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s[u] = p[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float a = KT.t[j + u].x, b = KT.t[j + u].y;
                        ar = __builtin_fmaf(a, s[u].x, ar);
                        ar = __builtin_fmaf(-b, s[u].y, ar);
                        ai = __builtin_fmaf(a, s[u].y, ai);
                        ai = __builtin_fmaf(b, s[u].x, ai);
                    }
                }
                for (; j < nt; ++j) {
                    const float2 s = p[j];
                    const float a = KT.t[j].x, b = KT.t[j].y;
                    ar = __builtin_fmaf(a, s.x, ar);
                    ar = __builtin_fmaf(-b, s.y, ar);
                    ai = __builtin_fmaf(a, s.y, ai);
                    ai = __builtin_fmaf(b, s.x, ai);
                }
            } else {
                if (!(P.D & 1)) {   // real taps, even decimation: the same 16-B reads
                    if (lsh & 1) {
                        const float2 s0 = p[0];
                        ar = __builtin_fmaf(KT.t[0].x, s0.x, ar);
                        ai = __builtin_fmaf(KT.t[0].x, s0.y, ai);
                        j = 1;
                    }
                    for (; j + 8 <= nt; j += 8) {
                        const float4* p4 = reinterpret_cast<const float4*>(p + j);
                        float4 qv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) qv[u] = p4[u];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float a0 = KT.t[j + 2 * u].x, a1 = KT.t[j + 2 * u + 1].x;
                            ar = __builtin_fmaf(a0, qv[u].x, ar);
                            ai = __builtin_fmaf(a0, qv[u].y, ai);
                            ar = __builtin_fmaf(a1, qv[u].z, ar);
                            ai = __builtin_fmaf(a1, qv[u].w, ai);
                        }
                    }
                }
                for (; j + 8 <= nt; j += 8) {
                    float2 s[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s[u] = p[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float a = KT.t[j + u].x;
                        ar = __builtin_fmaf(a, s[u].x, ar);
                        ai = __builtin_fmaf(a, s[u].y, ai);
                    }
                }
                for (; j < nt; ++j) {
                    const float2 s = p[j];
                    const float a = KT.t[j].x;
                    ar = __builtin_fmaf(a, s.x, ar);
                    ai = __builtin_fmaf(a, s.y, ai);
                }
            }
        } else {
        float tr = 0.f, ti = 0.f;
        if (lane < P.ntaps) {
            if (cplx) {
                float2 tt = reinterpret_cast<const float2*>(rt)[lane];
                tr = tt.x;
                ti = tt.y;
            } else
                tr = rt[lane];
        }
        for (int c0 = 0; c0 < P.ntaps; c0 += 64) {
            float ntr = 0.f, nti = 0.f;
            const int jn = c0 + 64 + lane;
            if (jn < P.ntaps) {
                if (cplx) {
                    float2 tt = reinterpret_cast<const float2*>(rt)[jn];
                    ntr = tt.x;
                    nti = tt.y;
                } else
                    ntr = rt[jn];
            }
            const int lim = (P.ntaps - c0) < 64 ? (P.ntaps - c0) : 64;
            const float2* pc = p + c0;
            // hand-unrolled by 8 so that eight ds_read_b64 are in flight per wait
            auto rl = [](float v, int l) {
                return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
            };
            int j = 0;
            if (cplx) {
                for (; j + 8 <= lim; j += 8) {
                    float2 s[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s[u] = pc[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float a = rl(tr, j + u), b = rl(ti, j + u);
                        ar = __builtin_fmaf(a, s[u].x, ar);
                        ar = __builtin_fmaf(-b, s[u].y, ar);
                        ai = __builtin_fmaf(a, s[u].y, ai);
                        ai = __builtin_fmaf(b, s[u].x, ai);
                    }
                }
                for (; j < lim; ++j) {
                    const float2 s = pc[j];
                    const float a = rl(tr, j), b = rl(ti, j);
                    ar = __builtin_fmaf(a, s.x, ar);
                    ar = __builtin_fmaf(-b, s.y, ar);
                    ai = __builtin_fmaf(a, s.y, ai);
                    ai = __builtin_fmaf(b, s.x, ai);
                }
            } else {
                for (; j + 8 <= lim; j += 8) {
                    float2 s[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s[u] = pc[j + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float a = rl(tr, j + u);
                        ar = __builtin_fmaf(a, s[u].x, ar);
                        ai = __builtin_fmaf(a, s[u].y, ai);
                    }
                }
                for (; j < lim; ++j) {
                    const float2 s = pc[j];
                    const float a = rl(tr, j);
                    ar = __builtin_fmaf(a, s.x, ar);
                    ai = __builtin_fmaf(a, s.y, ai);
                }
            }
            tr = ntr;
            ti = nti;
        }
        }
        if (!have_y) {
            ar = 0.f;
            ai = 0.f;
        }
    }
    if (have_y) {
        const float* __restrict__ rt = P.rtaps;
        if (USE_LDS) {
        } else {
            long long base = lo_idx + (long long)t * P.D;
            for (int j = 0; j < P.ntaps; ++j) {
                float2 s = fetch(base + j);
                if (P.taps_complex) {
                    float tr = rt[2 * j], ti = rt[2 * j + 1];
                    ar = __builtin_fmaf(tr, s.x, ar);
                    ar = __builtin_fmaf(-ti, s.y, ar);
                    ai = __builtin_fmaf(tr, s.y, ai);
                    ai = __builtin_fmaf(ti, s.x, ai);
                } else {
                    float tp = rt[j];
                    ar = __builtin_fmaf(tp, s.x, ar);
                    ai = __builtin_fmaf(tp, s.y, ai);
                }
            }
        }
        if (P.nco_tab) {
            float2 r = nco_eval(P.nco_tab, (P.nco_m0 + (unsigned long long)m) * P.nco_inc);
            float yr = __builtin_fmaf(ar, r.x, -(ai * r.y));
            float yi = __builtin_fmaf(ar, r.y, ai * r.x);
            ar = yr;
            ai = yi;
        }
    }

    frontend_finish<USE_LDS, QD>(P, tile, cap, t, q, o, ar, ai, ysh, sh);
}

// CU8 = the input is rtl-sdr bytes (sdr_set_input_format).  Aligned cu8 batches and calls
// run on k_frontend_pipe below; this instantiation is the general path (unaligned buffers,
// pre-rotation), staging through guarded per-sample fetches.
template <bool USE_LDS, bool CU8 = false, bool QD = true>
__global__ void __launch_bounds__(256) k_frontend(FrontParams P, FrontKTaps KT)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int sh[12];
    // discriminator history exchange: aliases the (by then consumed) tile when there is one
    __shared__ float2 ysh_static[USE_LDS ? 1 : 256];
    frontend_tile<USE_LDS, CU8, QD>(P, KT, blockIdx.x, blockIdx.y, reinterpret_cast<float2*>(smem_raw), sh, ysh_static);
}

// Tap loop of k_frontend_raw8: samples are (I, Q) byte pairs in LDS, converted as they are read.
template <bool CPLX>
__device__ __forceinline__ void raw8_fir(const FrontKTaps& KT, const unsigned short* pu, const int nt, const bool edge,
                                         const bool even_d, const bool lead_odd, const long long s0, const long long vlo, const long long n,
                                         float& ar, float& ai)
{
    const float kS = 0.0078125f, kC = -127.4f * 0.0078125f;
    auto mac = [&](unsigned r, bool ok, int j) {
        float sx = __builtin_fmaf((float)(r & 0xffu), kS, kC);
        float sy = __builtin_fmaf((float)(r >> 8), kS, kC);
        if (!ok) sx = sy = 0.f;
        const float a = KT.t[j].x, b = KT.t[j].y;
        ar = __builtin_fmaf(a, sx, ar);
        if (CPLX) ar = __builtin_fmaf(-b, sy, ar);
        ai = __builtin_fmaf(a, sy, ai);
        if (CPLX) ai = __builtin_fmaf(b, sx, ai);
    };
    int j = 0;
    if (!edge && even_d) {
        // Even decimation: every lane's window starts at the same parity, so after one single tap when that parity is
        // odd the window is read as DWORDS (two samples each, 4-byte aligned: ds_read2_b32 at full LDS rate).  Read as
        // 16-bit items the compiler merged eight of them into one ds_read_b128 at a 4-byte-aligned address (lane
        // stride 2*D bytes), which the LDS serves at a third of its rate: 24 cycles per instruction by
        // SQ_LDS_IDX_ACTIVE, the LDS as busy as the vector ALUs (profiles/r03_i_raw8_sq.csv).
        if (lead_odd) {   // uniform: LDS sample index of tap 0 = off + t*D, D even
            mac(pu[0], true, 0);
            j = 1;
        }
        for (; j + 8 <= nt; j += 8) {
            const unsigned* pw = reinterpret_cast<const unsigned*>(pu + j);
            unsigned w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = pw[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mac(w[u] & 0xffffu, true, j + 2 * u);
                mac(w[u] >> 16, true, j + 2 * u + 1);
            }
        }
    } else if (!edge) {
        for (; j + 8 <= nt; j += 8) {
            unsigned r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = pu[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) mac(r[u], true, j + u);
        }
    }
    for (; j < nt; ++j) {
        const long long idx = s0 + j;
        mac(pu[j], !edge || (idx >= vlo && idx < n), j);
    }
}

// rtl-sdr bytes kept as bytes: the tile is staged by the same 16-B direct-to-LDS DMA as the complex64
// kernel, but as the raw (I, Q) byte pairs -- 2 B per sample, a quarter of the float tile, so four times
// as many workgroups fit a CU -- and every lane converts its window's samples ((u - 127.4)/128, the two
// exact float ops of the oracle) on the way from LDS into the fmas.  With 61 taps at decimation 50 a
// sample is used 1.2 times, so converting at the read costs what converting at the staging did.
// Zero padding (capture edges without history) cannot be written as a byte: edge tiles (wave-uniform
// test) mask the samples outside [first valid, n) in the tap loop instead.  Needs the taps in the
// kernel arguments (<= 64), a 16-B aligned buffer and no pre-rotation; other cu8 calls take
// k_frontend_pipe / k_frontend<., true>.
__global__ void __launch_bounds__(256) k_frontend_raw8(FrontParams P, FrontKTaps KT)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int sh[12];
    __shared__ float2 ysh[256];
    const int T = blockDim.x, t = threadIdx.x, tile = blockIdx.x, cap = blockIdx.y;
    const int q = (P.demod == SDR_DEMOD_QUAD) ? 1 : 0;
    const int H = P.ntaps - 1;
    const int count = (T - 1) * P.D + P.ntaps;
    const long long m_t0 = (long long)tile * P.tile_out - q;
    const long long lo_idx = P.in0 + m_t0 * P.D - H;                       // first input sample of the tile
    const int off = (int)(((long long)cap * P.stride + lo_idx) & 7);       // 8 samples = one 16-B chunk
    const long long lo = lo_idx - off;
    const int nch = (count + off + 7) >> 3;
    const uchar2* __restrict__ xb = reinterpret_cast<const uchar2*>(P.x) + (long long)cap * P.stride;
    const long long vlo = P.hist ? -(long long)P.hist_len : 0;             // first sample that exists
    uchar2* lds2 = reinterpret_cast<uchar2*>(smem_raw);

    // chunks wholly inside the capture: DMA, nothing else in the loop
    // (chunk c holds samples lo + 8c .. +7: inside the capture for c in [clo, chi); scalar counters, one bumped
    // pointer per lane, no lane mask on whole waves of an interior tile -- as k_frontend)
    {
        long long clo64 = lo < 0 ? (7 - lo) >> 3 : 0, chi64 = (P.n - lo) >> 3;
        if (clo64 > nch) clo64 = nch;
        if (chi64 > nch) chi64 = nch;
        if (chi64 < clo64) chi64 = clo64;
        const int clo = (int)clo64, chi = (int)chi64;
        const int wb = __builtin_amdgcn_readfirstlane(t & ~63);
        const char* __restrict__ gp = reinterpret_cast<const char*>(xb + lo) + 16u * (unsigned)(wb + lane_id());
        const unsigned gstep = 16u * (unsigned)T;
        int c0 = wb;
        if (clo == 0 && chi == nch)
            for (; c0 + 64 <= nch; c0 += T, gp += gstep)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(smem_raw + 16 * c0), 16, 0,
                                                 SDR_GLDS_AUX);
        for (; c0 < nch; c0 += T, gp += gstep) {
            const int c = c0 + lane_id();
            if (c >= clo && c < chi)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(smem_raw + 16 * c0), 16, 0,
                                                 SDR_GLDS_AUX);
        }
    }
    const bool edge = lo < 0 || lo + 8LL * nch > P.n;                      // uniform
    if (edge) {
        for (int c = t; c < nch; c += T) {
            const long long s = lo + 8LL * c;
            if (!(s >= 0 && s + 8 <= P.n))
                for (int j = 0; j < 8; ++j) {
                    const long long idx = s + j;
                    uchar2 u = make_uchar2(0, 0);
                    if (idx >= 0) {
                        if (idx < P.n) u = xb[idx];
                    } else if (idx >= vlo)
                        u = reinterpret_cast<const uchar2*>(P.hist)[P.hist_len + idx];
                    lds2[8 * c + j] = u;
                }
        }
    }
    __syncthreads();

    const long long m = m_t0 + t;
    const bool have_y = (m >= 0 || P.hist != nullptr) && (m < P.nout);
    float ar = 0.f, ai = 0.f;
    {
        const unsigned short* pu = reinterpret_cast<const unsigned short*>(smem_raw) + off + t * P.D;
        const long long s0 = lo_idx + (long long)t * P.D;                  // sample index of tap 0
        if (P.taps_complex)
            raw8_fir<true>(KT, pu, P.ntaps, edge, !(P.D & 1), (off & 1) != 0, s0, vlo, P.n, ar, ai);
        else
            raw8_fir<false>(KT, pu, P.ntaps, edge, !(P.D & 1), (off & 1) != 0, s0, vlo, P.n, ar, ai);
        if (!have_y) ar = ai = 0.f;
    }
    if (have_y && P.nco_tab) {
        const float2 r = nco_eval(P.nco_tab, (P.nco_m0 + (unsigned long long)m) * P.nco_inc);
        const float yr = __builtin_fmaf(ar, r.x, -(ai * r.y));
        const float yi = __builtin_fmaf(ar, r.y, ai * r.x);
        ar = yr;
        ai = yi;
    }
    frontend_finish<false>(P, tile, cap, t, q, m, ar, ai, ysh, sh);
}

// Register-staged form, the default for cu8 input: the tile's bytes come in as plain
// 16-B global loads (NCH chunks per lane, register array with static indexing; chunk =
// 8 cu8 samples or 2 complex64 samples) and are converted on the way into the float
// tile -- no raw LDS area, so cu8 gets the same 128-output tile as complex64 input
// (the DMA variant needed 64 KB for it and fell back to 64 outputs).  A workgroup may
// walk `tpw` consecutive tiles with the next tile's loads in flight while the current
// one is filtered; measured, that never pays here (cu8 headline batch: tpw 1 -> 0.353 ms,
// 2 -> 0.414, 4 -> 0.447, 8 -> 0.543; complex64: 0.58 ms against 0.455 ms for the DMA
// kernel), so tpw defaults to 1 and complex64 stays on k_frontend's LDS DMA.  Chunks
// that touch the capture's edges (zero padding, streaming history) take guarded
// per-sample fetches in a loop of their own.
template <bool CU8, int NCH>
__global__ void __launch_bounds__(256) k_frontend_pipe(FrontParams P, FrontKTaps KT, int tpw)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float2* lds = reinterpret_cast<float2*>(smem_raw);
    __shared__ int sh[12];
    __shared__ float2 ysh_static[1];
    constexpr int SPC = CU8 ? 8 : 2;   // samples per 16-B chunk
    const int T = blockDim.x, tid = threadIdx.x, cap = blockIdx.y;
    const int q = (P.demod == SDR_DEMOD_QUAD) ? 1 : 0;
    const int H = P.ntaps - 1;
    const int count = (T - 1) * P.D + P.ntaps;
    const char* __restrict__ xbytes = reinterpret_cast<const char*>(P.x) + (long long)cap * P.stride * (CU8 ? 2 : 8);
    const float2* __restrict__ x = P.x + (long long)cap * P.stride;
    const uchar2* __restrict__ xb = reinterpret_cast<const uchar2*>(P.x) + (long long)cap * P.stride;
    auto fetch = [&](long long idx) -> float2 {
        if (CU8) {
            uchar2 u;
            if (idx >= 0) {
                if (idx >= P.n) return make_float2(0.f, 0.f);
                u = xb[idx];
            } else if (P.hist && idx >= -(long long)P.hist_len)
                u = reinterpret_cast<const uchar2*>(P.hist)[P.hist_len + idx];
            else
                return make_float2(0.f, 0.f);
            return cu8_to_f2(u.x, u.y);
        }
        if (idx >= 0) return idx < P.n ? x[idx] : make_float2(0.f, 0.f);
        if (P.hist && idx >= -(long long)P.hist_len) return P.hist[P.hist_len + idx];
        return make_float2(0.f, 0.f);
    };
    // Per tile (all wave-uniform): lo = first sample of the tile's chunk grid (chunk-aligned in
    // the capture buffer), off = lo_idx - lo, nch = chunks, [c_in0, c_in1) = chunks that lie
    // wholly inside the capture (they go through the registers; the rest are edge chunks).
    struct TileGeo {
        long long lo;
        int off, nch, c_in0, c_in1;
    };
    auto geo = [&](int tile) -> TileGeo {
        TileGeo g;
        const long long lo_idx = P.in0 + ((long long)tile * P.tile_out - q) * P.D - H;
        g.off = (int)(((long long)cap * P.stride + lo_idx) & (SPC - 1));
        g.lo = lo_idx - g.off;
        g.nch = (count + g.off + SPC - 1) / SPC;
        long long a = g.lo < 0 ? (-g.lo + SPC - 1) / SPC : 0;            // first chunk with s >= 0
        long long b = P.n - g.lo < 0 ? 0 : (P.n - g.lo) / SPC;            // chunks with s + SPC <= n
        if (a > g.nch) a = g.nch;
        if (b > g.nch) b = g.nch;
        if (b < a) b = a;
        g.c_in0 = (int)a;
        g.c_in1 = (int)b;
        return g;
    };
    const int tile0 = blockIdx.x * tpw;
    int tile1 = tile0 + tpw;
    if (tile1 > P.tiles_per_capture) tile1 = P.tiles_per_capture;

    uint4 r[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) r[k] = make_uint4(0u, 0u, 0u, 0u);
    auto issue = [&](const TileGeo& g) {
        const char* tb = xbytes + g.lo * (CU8 ? 2 : 8);   // uniform base, 32-bit lane offsets
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = k * T + tid;
            if (c >= g.c_in0 && c < g.c_in1) r[k] = *reinterpret_cast<const uint4*>(tb + 16u * (unsigned)c);
        }
    };
    TileGeo gcur;
    if (tile0 < tile1) {
        gcur = geo(tile0);
        issue(gcur);
    }
    for (int tile = tile0; tile < tile1; ++tile) {
        const TileGeo g = gcur;
        // registers -> float tile (sample s sits at lds[s - lo])
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = k * T + tid;
            if (c >= g.c_in0 && c < g.c_in1) {
                if (CU8) {
                    const float kS = 0.0078125f, kC = -127.4f * 0.0078125f;
                    const unsigned wd[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
                    float4* dst = reinterpret_cast<float4*>(lds + 8 * c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float4 o;
                        o.x = __builtin_fmaf((float)(wd[j] & 0xffu), kS, kC);
                        o.y = __builtin_fmaf((float)((wd[j] >> 8) & 0xffu), kS, kC);
                        o.z = __builtin_fmaf((float)((wd[j] >> 16) & 0xffu), kS, kC);
                        o.w = __builtin_fmaf((float)(wd[j] >> 24), kS, kC);
                        dst[j] = o;
                    }
                } else
                    *reinterpret_cast<uint4*>(lds + 2 * c) = r[k];
            }
        }
        if (g.c_in0 > 0 || g.c_in1 < g.nch) {   // uniform: edge tiles only
            for (int c = tid; c < g.nch; c += T)
                if (c < g.c_in0 || c >= g.c_in1)
                    for (int j = 0; j < SPC; ++j) lds[SPC * c + j] = fetch(g.lo + (long long)c * SPC + j);
        }
        if (tile + 1 < tile1) {   // next tile's bytes: in flight while this tile is filtered
            gcur = geo(tile + 1);
            issue(gcur);
        }
        __syncthreads();
        frontend_tile<true, CU8>(P, KT, tile, cap, lds, sh, ysh_static, g.off);
        __syncthreads();   // the tile and sh[] are reused by the next round
    }
}

// Real FIR (decimation 1) on demodulated floats + gain + slicer:
// TFA/analyzer2.grc:1075-1147 low_pass_filter (fff) between mag^2 and threshold.
__global__ void __launch_bounds__(256) k_post_fir_slice(PostParams P)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    __shared__ int sh[12];
    const int T = blockDim.x, tile = blockIdx.x, cap = blockIdx.y;
    const long long o0 = (long long)tile * P.tile_out;
    const float* __restrict__ v = P.v + (long long)cap * P.v_stride;
    const int H = P.ntaps - 1;
    const int count = T + H;
    for (int i = threadIdx.x; i < count; i += T) {
        long long idx = o0 - H + i;
        float s = 0.f;
        if (idx >= 0) {
            if (idx < P.nout) s = v[idx];
        } else if (P.hist)
            s = P.hist[H + idx];
        lds[i] = s;
    }
    __syncthreads();
    const int t = threadIdx.x;
    const long long o = o0 + t;
    bool valid = o < P.nout;
    float a = 0.f;
    if (valid)
        for (int j = 0; j < P.ntaps; ++j) a = __builtin_fmaf(P.rtaps[j], lds[t + j], a);
    if (valid && P.tap) (P.tap + (long long)cap * P.tap_stride)[o] = a;
    float g = P.gain * a;
    slice_tile(valid, g, o, o0, P.lo, P.hi, P.keep_n, P.keep_c0, P.bits + (long long)cap * P.bits_stride,
               P.tsum + (long long)cap * P.tiles_per_capture + tile, sh);
}
