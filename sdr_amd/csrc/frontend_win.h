// Channel filter at decimation 1, real taps or the complex band-pass taps of freq_xlating_fir_filter_ccc
// (win_tile<true>: two packed fmas per tap, the NCO in the epilogue) (fir_filter_ccf as the reference wires it:
// TFA/decoder.grc:1163-1234 27 taps, FM-Radio/FM-Radio.grc:674-746 33 taps, TS33C/live_decoder.grc:917-988
// 121 taps) on the vector ALUs with a SLIDING WINDOW: every lane owns WIN_R consecutive outputs and
// keeps the samples they share in registers, so one LDS read feeds WIN_R packed fmas instead of one.
//
//   acc[r] += tap[j] * x[lane*R + r + j]        r = 0..R-1, j ascending (oldest sample first: the
//                                               oracle's order, one fmaf per tap and component)
//
// The window is a ring of WIN_W = R + 6 registers (sample s of the lane sits in slot s mod W); the tap
// loop is unrolled by W so every slot index is static.  After sub-step u the slots u-1 and u are dead
// and take the samples W ahead (one aligned 16-B ds_read for two samples, in flight for five
// sub-steps = 50 packed fmas).  Lane stride = R samples = 5 granules of 16 B: a 16-lane group of a
// ds_read_b128 touches 16 different granules of the 64 banks -- conflict-free without padding.
// Taps sit in LDS (broadcast reads, in order with the sample reads on one counter).  A tile that
// starts on an odd sample is read from one sample earlier with the tap table shifted by one and the
// dummy first step skipped (no 0*x product is ever formed, so non-finite samples stay local).
// 4*ntaps/1 flop per input sample against 8 bytes: fp32-FMA bound from ~16 taps up.
#pragma once
#include "frontend_mfma.h"

#define WIN_R 10
#define WIN_W 16
// k_frontend_win's DM template argument: 0 = demodulator and outputs read from the parameters; SDR_DEMOD_MAG2 = mag^2
// chains only; SDR_DEMOD_QUAD | WIN_DM_SLICE_ONLY = discriminator chains that only slice (no float output)
#define WIN_DM_SLICE_ONLY 0x10
#define WIN_DM_QUAD_SLICE (SDR_DEMOD_QUAD | WIN_DM_SLICE_ONLY)

// LDS floats of the tap table (shifted copy + one block of slack), 16-B multiple
__host__ __device__ inline int win_tap_floats(int ntaps, bool cplx = false) { return ((ntaps + 1 + WIN_W + 3) & ~3) * (cplx ? 2 : 1); }
// staged sample positions: T*R outputs + taps rounded up to whole blocks + one block of look-ahead
__host__ __device__ inline int win_sample_slots(int T, int ntaps)
{
    return T * WIN_R + ((ntaps + 1 + WIN_W - 1) / WIN_W) * WIN_W + WIN_W + 2 + 8;   // + 8: cu8 tiles start on a 16-B chunk
}

typedef float win_v2f __attribute__((ext_vector_type(2)));

// Epilogue of one tile.  FULL: every output of the tile exists and is emitted (but tile index 0 of a
// discriminator tile, which is the previous tile's last output); if the chain slices, keep_n == 1.
// DM (compile time): 0 = the demodulator is read from P.demod; SDR_DEMOD_MAG2 = this instantiation serves mag^2
// chains only (no discriminator code, no arctangent table: fewer registers, one more wave per SIMD)
template <bool FULL, int DM>
__device__ __forceinline__ void win_epilogue(const FrontParams& P, const int tile, const int cap, win_v2f* acc,
                                             const long long m_t0, const int i_lo, const int i_hi, const int q, int* sh,
                                             const float* s_atan, float2* s_edge, unsigned char* pack)
{
    constexpr int R = WIN_R;
    typedef win_v2f v2f;
    const int demod = DM ? (DM & 3) : P.demod;
    constexpr bool SLICE_ONLY = (DM & WIN_DM_SLICE_ONLY) != 0;   // no float tap output in this instantiation
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long m0 = m_t0 + (long long)R * t;
    bool valid[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = R * t + r;
        if (FULL)
            valid[r] = r > 0 || i >= q;
        else {
            if (!(i >= i_lo && i < i_hi)) acc[r] = (v2f){0.f, 0.f};
            valid[r] = i >= q && i < i_hi;
        }
    }
    if (P.nco_tab) {   // freq_xlating_fir_filter_ccc: complex band-pass taps in front, the output rotated (uniform)
        unsigned long long ph = (P.nco_m0 + (unsigned long long)m0) * P.nco_inc;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float2 rot = nco_eval(P.nco_tab, ph);
            const v2f y = acc[r];
            acc[r] = (v2f){__builtin_fmaf(y.x, rot.x, -(y.y * rot.y)), __builtin_fmaf(y.x, rot.y, y.y * rot.x)};
            ph += P.nco_inc;
        }
    }
    if (demod == SDR_DEMOD_NONE) {
        if (P.tap) {
            float2* dst = reinterpret_cast<float2*>(P.tap + (long long)cap * P.tap_stride);
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (valid[r]) dst[m0 + r] = make_float2(acc[r].x, acc[r].y);
        }
        return;
    }
    float v[R];
    if (demod == SDR_DEMOD_MAG2) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = __builtin_fmaf(acc[r].x, acc[r].x, acc[r].y * acc[r].y);
    } else {
        // the output before the thread's first one belongs to the neighbouring lane (wave)
        float2 pv;
        pv.x = __shfl_up(acc[R - 1].x, 1);
        pv.y = __shfl_up(acc[R - 1].y, 1);
        if (lane == 63) s_edge[wave] = make_float2(acc[R - 1].x, acc[R - 1].y);
        __syncthreads();
        if (lane == 0) pv = wave ? s_edge[wave - 1] : make_float2(0.f, 0.f);
        if (FULL && P.quad_fast && (SLICE_ONLY || !P.tap)) {
            // Threshold-only form on an interior tile: the classes go into the slicer's event masks directly (outputs
            // walked from the last to the first, mask = mask + mask + class bit by one add-with-carry whose carry-in is
            // the class's lane mask) -- no stand-in values, no second compare.  Same tests as the general form below.
            const bool hi_up = P.q_s_hi >= 0.f, lo_up = P.q_s_lo > 0.f;
            unsigned Hm = 0, Km = 0;
            bool inband = false;
#pragma unroll
            for (int r = R - 1; r >= 0; --r) {
                const float2 pr = r ? make_float2(acc[r - 1].x, acc[r - 1].y) : pv;
                const float tr = __builtin_fmaf(acc[r].x, pr.x, acc[r].y * pr.y);
                const float ti = __builtin_fmaf(acc[r].y, pr.x, -(acc[r].x * pr.y));
                const float ch = __builtin_fmaf(ti, P.q_c_hi, -(tr * P.q_s_hi));
                const float cl = __builtin_fmaf(ti, P.q_c_lo, -(tr * P.q_s_lo));
                const float lim = (__builtin_fabsf(tr) + __builtin_fabsf(ti)) * P.q_band;
                inband |= !(__builtin_fabsf(ch) > lim) || !(__builtin_fabsf(cl) > lim);
                const unsigned long long m_up = __ballot(ti >= 0.f), m_ch = __ballot(ch > 0.f), m_cl = __ballot(cl < 0.f);
                const unsigned long long gt_hi = hi_up ? (m_up & m_ch) : (m_up | m_ch);
                const unsigned long long lt_lo = lo_up ? (~m_up | m_cl) : (~m_up & m_cl);
                asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(Hm) : "s"(gt_hi) : "vcc");
                asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(Km) : "s"(lt_lo) : "vcc");
            }
            if (__ballot(inband) != 0ull) {   // some lane of this wave has a sample inside a band: the wave's tile the exact way
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float2 pr = r ? make_float2(acc[r - 1].x, acc[r - 1].y) : pv;
                    const float tr = __builtin_fmaf(acc[r].x, pr.x, acc[r].y * pr.y);
                    const float ti = __builtin_fmaf(acc[r].y, pr.x, -(acc[r].x * pr.y));
                    v[r] = P.gain * fast_atan2f_dev(s_atan, ti, tr);
                }
                slice_masks_xn<R, true>(valid, v, P.lo, P.hi, Hm, Km);
            } else {
                const unsigned keep = valid[0] ? ~0u : ~1u;
                Hm &= keep;
                Km &= ~Hm & keep;
            }
            slice_tile_hk<R, true>(valid, Hm, Km, R * t, m_t0, P.keep_n, P.keep_c0, P.bits + (long long)cap * P.bits_stride,
                                   P.tsum + (long long)cap * P.tiles_per_capture + tile, sh, pack);
            return;
        } else if (P.quad_fast && (SLICE_ONLY || !P.tap)) {
            // Threshold-only form (round 3).  The slicer needs whether gain*fast_atan2f(ti, tr) is above hi, below lo
            // or between -- the arctangent itself (~41 vector instructions: a division, two dependent table reads,
            // the octant ladder) cost as much as the 27-tap filter of the TFA chain.  fast_atan2f is within 1.6e-6 rad
            // of the true angle (tests/test_oracle_dsp.py), so a sample whose direction is further than the guard
            // band (1e-4 rad) from both threshold directions has its class decided by the SIGN of two cross
            // products |t| sin(theta - phi) and of ti (the arctangent's own y >= 0 test); the few inside a band --
            // and anything non-finite or of zero length -- take the full arctangent, whole wave, so that every byte
            // equals the oracle's.  v[] then carries +inf / -inf / lo as stand-ins for "above hi" / "below lo" / "hold".
            const bool hi_up = P.q_s_hi >= 0.f, lo_up = P.q_s_lo > 0.f;   // threshold directions in the upper half plane (uniform)
            bool inband = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float2 pr = r ? make_float2(acc[r - 1].x, acc[r - 1].y) : pv;
                const float tr = __builtin_fmaf(acc[r].x, pr.x, acc[r].y * pr.y);
                const float ti = __builtin_fmaf(acc[r].y, pr.x, -(acc[r].x * pr.y));
                const float ch = __builtin_fmaf(ti, P.q_c_hi, -(tr * P.q_s_hi));
                const float cl = __builtin_fmaf(ti, P.q_c_lo, -(tr * P.q_s_lo));
                // |t| sin(guard) <= (|tr| + |ti|) * guard: the band test on the 1-norm is at most sqrt(2) wider
                const float lim = (__builtin_fabsf(tr) + __builtin_fabsf(ti)) * P.q_band;
                inband |= !(__builtin_fabsf(ch) > lim) || !(__builtin_fabsf(cl) > lim);   // also NaN / Inf / zero length
                const bool up = ti >= 0.f;
                const bool gt_hi = hi_up ? (up && ch > 0.f) : (up || ch > 0.f);
                const bool lt_lo = lo_up ? (!up || cl < 0.f) : (!up && cl < 0.f);
                v[r] = gt_hi ? __builtin_inff() : (lt_lo ? -__builtin_inff() : P.lo);
            }
            if (__ballot(inband) != 0ull) {   // some lane of this wave has a sample inside a band: the wave's tile the exact way
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float2 pr = r ? make_float2(acc[r - 1].x, acc[r - 1].y) : pv;
                    const float tr = __builtin_fmaf(acc[r].x, pr.x, acc[r].y * pr.y);
                    const float ti = __builtin_fmaf(acc[r].y, pr.x, -(acc[r].x * pr.y));
                    v[r] = P.gain * fast_atan2f_dev(s_atan, ti, tr);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // computed for every lane and output (no exec-masked block per output); outputs that do not
                // exist have zero accumulators, the ones not emitted are dropped by valid[] below
                const float2 pr = r ? make_float2(acc[r - 1].x, acc[r - 1].y) : pv;
                const float tr = __builtin_fmaf(acc[r].x, pr.x, acc[r].y * pr.y);
                const float ti = __builtin_fmaf(acc[r].y, pr.x, -(acc[r].x * pr.y));
                const float a = fast_atan2f_dev(s_atan, ti, tr);
                v[r] = P.gain * (valid[r] ? a : 0.f);
            }
        }
    }
    if (demod == SDR_DEMOD_MAG2) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = P.gain * v[r];
    }
    if (!SLICE_ONLY && P.tap) {
        float* dst = P.tap + (long long)cap * P.tap_stride;
        if (FULL) {
            // a lane's R floats in a row are R stores at stride R floats, each touching every cache line of the
            // wave's run (0.026 of 0.10 ms on the FM chain's 33-tap stage): transposed through the tile's sample
            // area -- free once every wave is past its tap loop -- they leave as R coalesced 256-B stores per wave
            float* const lf = reinterpret_cast<float*>(pack) + wave * 64 * R;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) lf[lane * R + r] = v[r];
            float* const o = dst + m_t0 + wave * 64 * R;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int idx = k * 64 + lane;
                if (wave * 64 * R + idx >= q) o[idx] = lf[idx];
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (valid[r]) dst[m0 + r] = v[r];
        }
    }
    if (!SLICE_ONLY && !P.do_slice) return;
    slice_tile_xn<R, FULL>(valid, v, R * t, m_t0, P.lo, P.hi, P.keep_n, P.keep_c0, P.bits + (long long)cap * P.bits_stride,
                           P.tsum + (long long)cap * P.tiles_per_capture + tile, sh, pack);
}


// One tile: blockDim.x = T (64 or 256) threads, T*WIN_R - q outputs.
// CPLX: complex taps (rtaps = interleaved re,im): per tap and output two packed fmas in the oracle's order,
//   (re, im) += a * (x.re, x.im);   (re, im) += (-b, b) * (x.im, x.re)        tap = a + jb
// the halves of the tap pair and the swapped sample picked by op_sel, the sign by neg_lo: no extra moves.
template <bool CPLX, int DM>
__device__ __forceinline__ void win_tile(const FrontParams& P, const int tile, const int cap, unsigned char* smem_raw,
                                         int* sh, const float* s_atan, float2* s_edge)
{
    const int demod = DM ? (DM & 3) : P.demod;
    constexpr int R = WIN_R, W = WIN_W;
    const int T = blockDim.x, t = threadIdx.x, lane = t & 63;
    const int nt = P.ntaps, H = nt - 1;
    const int q = (demod == SDR_DEMOD_QUAD) ? 1 : 0;
    const int ntl = win_tap_floats(nt, CPLX);
    float* const tapL = reinterpret_cast<float*>(smem_raw);
    float2* const lds = reinterpret_cast<float2*>(tapL + ntl);

    const long long o0 = (long long)tile * P.tile_out;
    const long long m_t0 = o0 - q;                         // FIR output index of thread 0's first output
    const float2* __restrict__ x = P.x + (long long)cap * P.stride;
    const long long lo_idx = P.in0 + m_t0 - H;             // first input sample of the tile (D == 1)
    const int count = T * R + H;

    auto fetch = [&](long long idx) -> float2 {
        if (idx >= 0) return idx < P.n ? x[idx] : make_float2(0.f, 0.f);
        if (P.hist && idx >= -(long long)P.hist_len) return P.hist[P.hist_len + idx];
        return make_float2(0.f, 0.f);
    };
    int lsh = 0;   // LDS index of sample lo_idx
    if (P.in_fmt == SDR_IN_CU8) {
        // rtl-sdr bytes ((u8 - 127.4) / 128 per component, the oracle's two exact ops): converted on the way in.
        // Interior tiles of an aligned buffer: 16-B chunks of eight samples from the chunk boundary below the
        // tile, four 16-B LDS writes each; edge tiles and unaligned buffers sample by sample.
        const uchar2* __restrict__ xb = reinterpret_cast<const uchar2*>(P.x) + (long long)cap * P.stride;
        auto fetch8 = [&](long long idx) -> float2 {
            uchar2 u;
            if (idx >= 0) {
                if (idx >= P.n) return make_float2(0.f, 0.f);
                u = xb[idx];
            } else if (P.hist && idx >= -(long long)P.hist_len)
                u = reinterpret_cast<const uchar2*>(P.hist)[P.hist_len + idx];
            else
                return make_float2(0.f, 0.f);
            return cu8_to_f2(u.x, u.y);
        };
        const long long gidx = (long long)cap * P.stride + lo_idx;
        const int s8 = P.vec_ok ? (int)(gidx & 7) : 0;
        const long long lo_al = lo_idx - s8;
        const int nch = (count + s8 + 7) >> 3;
        lsh = s8;
        if (P.vec_ok && lo_al >= 0 && lo_al + 8LL * nch <= P.n) {   // uniform
            const uint4* __restrict__ src = reinterpret_cast<const uint4*>(xb + lo_al);
            for (int c0 = 0; c0 < nch; c0 += 2 * T) {
                uint4 rg[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int c = c0 + k * T + t;
                    rg[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (c < nch) rg[k] = src[c];
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int c = c0 + k * T + t;
                    if (c < nch) {
                        const unsigned wd[4] = {rg[k].x, rg[k].y, rg[k].z, rg[k].w};
                        float4* dst = reinterpret_cast<float4*>(lds + 8 * c);
#pragma unroll
                        for (int d = 0; d < 4; ++d) {   // one dword = two samples (I, Q, I, Q)
                            const float2 a = cu8_to_f2((unsigned char)(wd[d] & 0xffu), (unsigned char)((wd[d] >> 8) & 0xffu));
                            const float2 b = cu8_to_f2((unsigned char)((wd[d] >> 16) & 0xffu), (unsigned char)(wd[d] >> 24));
                            dst[d] = make_float4(a.x, a.y, b.x, b.y);
                        }
                    }
                }
            }
        } else {
            for (int i = t; i < count + s8; i += T) lds[i] = fetch8(lo_al + i);
        }
    } else if (P.vec_ok && !P.pre_tab) {
        // 16-B chunks inside the capture by DMA (a loop that holds nothing else), the few that touch
        // its edges by guarded loads -- as k_frontend
        const long long gidx = (long long)cap * P.stride + lo_idx;
        const int odd = (int)(gidx & 1);
        const long long lo_al = lo_idx - odd;
        const int pairs = (count + odd + 1) >> 1;
        lsh = odd;
        // chunk p holds samples lo_al + 2p, +1: inside the capture for p in [plo, phi) -- uniform bounds,
        // so a lane tests two 32-bit numbers and adds a 32-bit offset to a uniform base
        long long plo64 = lo_al < 0 ? (1 - lo_al) >> 1 : 0, phi64 = (P.n - lo_al) >> 1;
        if (plo64 > pairs) plo64 = pairs;
        if (phi64 > pairs) phi64 = pairs;
        if (phi64 < plo64) phi64 = plo64;
        const int plo = (int)plo64, phi = (int)phi64;
        // the wave's first chunk as a scalar: the LDS base of every DMA (M0) and the loop counter stay in scalar
        // registers, a lane keeps one 64-bit pointer and bumps it (11 vector instructions per DMA before, 4 now)
        const int wb = __builtin_amdgcn_readfirstlane(t & ~63);
        const char* __restrict__ gp = reinterpret_cast<const char*>(x + lo_al) + 16u * (unsigned)(wb + lane);
        const unsigned gstep = 16u * (unsigned)T;
        int p0 = wb;
        if (plo == 0 && phi == pairs)   // interior tile (uniform): whole waves of chunks need no lane mask
            for (; p0 + 64 <= pairs; p0 += T, gp += gstep)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(lds + 2 * p0), 16, 0,
                                                 SDR_GLDS_AUX);
        for (; p0 < pairs; p0 += T, gp += gstep) {
            const int p = p0 + lane;
            if (p >= plo && p < phi)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                                 (__attribute__((address_space(3))) void*)(lds + 2 * p0), 16, 0,
                                                 SDR_GLDS_AUX);
        }
        if (lo_al < 0 || lo_al + 2LL * pairs > P.n) {   // uniform: only edge tiles have guarded chunks
            for (int p = t; p < pairs; p += T) {
                const long long s = lo_al + 2LL * p;
                if (!(s >= 0 && s + 2 <= P.n)) {
                    lds[2 * p] = fetch(s);
                    lds[2 * p + 1] = fetch(s + 1);
                }
            }
        }
    } else {
        for (int i = t; i < count; i += T) {
            const long long idx = lo_idx + i;
            float2 v = fetch(idx);
            if (P.pre_tab) {   // freq_xlating_fir_filter with taps=[1] in front of the FIR (FM chain)
                const float2 r = nco_eval(P.pre_tab, (P.pre_idx0 + (unsigned long long)idx) * P.pre_inc);
                v = make_float2(__builtin_fmaf(v.x, r.x, -(v.y * r.y)), __builtin_fmaf(v.x, r.y, v.y * r.x));
            }
            lds[i] = v;
        }
    }
    // tap table, shifted by one when the tile starts odd: tapL[j] = rtaps[j - shift]
    const int shift = lsh & 1;
    if (CPLX) {
        float2* const tapC = reinterpret_cast<float2*>(tapL);
        const float2* __restrict__ rt2 = reinterpret_cast<const float2*>(P.rtaps);
        for (int i = t; i < (ntl >> 1); i += T) {
            const int idx = i - shift;
            tapC[i] = (idx >= 0 && idx < nt) ? rt2[idx] : make_float2(0.f, 0.f);
        }
    } else {
        for (int i = t; i < ntl; i += T) {
            const int idx = i - shift;
            tapL[i] = (idx >= 0 && idx < nt) ? P.rtaps[idx] : 0.f;
        }
    }
    __syncthreads();

    // ---- sliding-window FIR.  The packed fmas are written out (v_pk_fma_f32: one IEEE fma per
    // component, the same rounding as two fmaf calls) so that the tap can be taken from either half of
    // its register pair by op_sel -- the compiler's own selection moved every odd tap and every window
    // pair into fresh registers first (94 moves per 156 fmas).
    typedef win_v2f v2f;
    typedef float v4f __attribute__((ext_vector_type(4)));
    v2f acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = (v2f){0.f, 0.f};
    {
        const v4f* pw = reinterpret_cast<const v4f*>(lds + (lsh - shift) + t * R);   // 16-B aligned: two samples per read
        const v4f* pt = reinterpret_cast<const v4f*>(tapL);
        const int jlo = shift, jhi = nt + shift;
        v2f w[W];
#pragma unroll
        for (int k = 0; k < W / 2; ++k) {
            const v4f v = pw[k];
            w[2 * k] = v.xy;
            w[2 * k + 1] = v.zw;
        }
#define WIN_FMA_LO(A, TP, X) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(A) : "v"(TP), "v"(X))
#define WIN_FMA_HI(A, TP, X) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(A) : "v"(TP), "v"(X))
// complex tap TP = (a, b): second step, (lo, hi) += (-b * x.hi, b * x.lo)
#define WIN_CFMA_B(A, TP, X) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(A) : "v"(TP), "v"(X))
        for (int j0 = 0; j0 < jhi; j0 += W) {
            v2f tp[CPLX ? W : W / 2];   // real: tap pairs (t[2k], t[2k+1]); complex: one (re, im) per tap
#pragma unroll
            for (int k = 0; k < (CPLX ? W / 2 : W / 4); ++k) {
                const v4f v = pt[(CPLX ? (j0 >> 1) : (j0 >> 2)) + k];
                tp[2 * k] = v.xy;
                tp[2 * k + 1] = v.zw;
            }
            const v4f* pn = pw + ((j0 + W) >> 1);
            const bool full = j0 >= jlo && j0 + W <= jhi;   // uniform
#pragma unroll
            for (int u = 0; u < W; ++u) {
                if (full || (j0 + u >= jlo && j0 + u < jhi)) {
                    if (CPLX) {
#pragma unroll
                        for (int r = 0; r < R; ++r) WIN_FMA_LO(acc[r], tp[CPLX ? u : 0], w[(u + r) % W]);
#pragma unroll
                        for (int r = 0; r < R; ++r) WIN_CFMA_B(acc[r], tp[CPLX ? u : 0], w[(u + r) % W]);
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if (u & 1)
                                WIN_FMA_HI(acc[r], tp[u >> 1], w[(u + r) % W]);
                            else
                                WIN_FMA_LO(acc[r], tp[u >> 1], w[(u + r) % W]);
                        }
                    }
                }
                if (u & 1) {
                    const v4f v = pn[u >> 1];
                    w[u - 1] = v.xy;
                    w[u] = v.zw;
                }
            }
        }
#undef WIN_FMA_LO
#undef WIN_FMA_HI
#undef WIN_CFMA_B
    }

    // ---- epilogue: thread t owns outputs R*t .. R*t+R-1 of the tile (tile-relative index i: FIR output
    // m_t0 + i).  An output exists for i in [i_lo, i_hi) (m >= 0 or streaming history, m < nout) and is
    // emitted for i in [q, i_hi): uniform 32-bit limits instead of 64-bit compares per lane and output.
    const long long left = P.nout - m_t0;
    const int i_hi = left > (long long)T * R ? T * R : (left < 0 ? 0 : (int)left);
    const int i_lo = (P.hist != nullptr || m_t0 >= 0) ? 0 : (int)(-m_t0);
    // interior tile of a slicing chain that keeps every byte: no per-output masks, packed byte stores
    // ... or of a chain that only keeps the demodulated floats (FM stage B)
    const bool full = i_lo == 0 && i_hi == T * R && demod != SDR_DEMOD_NONE &&
                      (P.do_slice ? (P.keep_n == 1 && m_t0 - P.keep_c0 >= 0) : P.tap != nullptr);
    if (full)
        win_epilogue<true, DM>(P, tile, cap, acc, m_t0, i_lo, i_hi, q, sh, s_atan, s_edge, reinterpret_cast<unsigned char*>(lds));
    else
        win_epilogue<false, DM>(P, tile, cap, acc, m_t0, i_lo, i_hi, q, sh, s_atan, s_edge, nullptr);
}

// blockDim.x = T (64 or 256), P.D == 1, real taps, P.tile_out = T*WIN_R - q; one tile per workgroup.
// (A workgroup walking 2..16 consecutive tiles was measured twice and dropped.  Round 2a: no faster at any
// count.  Round 2b, with two sample buffers, the next tile's DMA in flight during the FIR and LDS-only barriers
// in the epilogue so that nothing waits for it: 0.339 ms against 0.285 on the TFA shape, 0.308 against 0.251 on
// the R shape -- two buffers are 43 KB of LDS, 3 workgroups per CU instead of 7, and this kernel lives on its
// 28 resident waves hiding each other's LDS and fma latencies, not on prefetch.)
template <bool CPLX, int DM>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DM == SDR_DEMOD_MAG2 || DM == WIN_DM_QUAD_SLICE ? 6 : 1))) k_frontend_win(FrontParams P)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    __shared__ int sh[12];
    __shared__ float s_atan[DM == SDR_DEMOD_MAG2 ? 1 : 257];
    __shared__ float2 s_edge[4];
    if (DM != SDR_DEMOD_MAG2 && (DM == WIN_DM_QUAD_SLICE || P.demod == SDR_DEMOD_QUAD))
        for (int i = threadIdx.x; i < 257; i += blockDim.x) s_atan[i] = P.atan_tab[i];
    win_tile<CPLX, DM>(P, blockIdx.x, blockIdx.y, smem_raw, sh, s_atan, s_edge);
}
