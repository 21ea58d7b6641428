// FM-Radio chain back half (FM-Radio/FM-Radio.grc:2903-2964):
// audio FIR (fir_filter_fff, decimating), fm_deemph (1-pole IIR, double state),
// rational_resampler_fff (polyphase L/M) * volume.  These run at <= quad_rate
// (a few 100 kS/s): a few MB per capture, L2-resident; the HBM-bound part of
// the chain is the first decimating FIR (k_frontend with input pre-rotation).
#pragma once
#include "sdr_common.h"

// Per-stream state of the FM chain's back half between sdr_fm_process calls (what the GNU Radio blocks keep:
// fir_filter / rational_resampler history, iir_filter_ffd delay line).
struct FmDeemphState {
    double xp, yp;   // x[n-1], y[n-1]
};

// y[m] = sum_k taps[k] v[in0 + m*D - k], oldest sample first (one fmaf per tap).  Samples before v[0] come
// from `hist` (the hist_len floats that precede v[0]; zeros before stream start).
__global__ void __launch_bounds__(256) k_fir_fff_decim(const float* __restrict__ v, const float* __restrict__ hist,
                                                       int hist_len, long long n_in, long long in0, long long nout,
                                                       const float* __restrict__ rtaps, int ntaps, int D,
                                                       float* __restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* lds = reinterpret_cast<float*>(smem_raw);
    const int T = blockDim.x;
    const long long o0 = (long long)blockIdx.x * T;
    const long long lo = in0 + o0 * D - (ntaps - 1);
    const int count = (T - 1) * D + ntaps;
    if (lo >= 0 && lo + count <= n_in) {
        // interior tile (uniform test): eight unguarded loads in flight per lane before the first LDS store
        // (one guarded load -> store per round was a chain of ~count/T load latencies per workgroup)
        const float* __restrict__ src = v + lo;
        for (int b = 0; b < count; b += 8 * T) {
            float r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = b + k * T + threadIdx.x;
                r[k] = src[i < count ? i : count - 1];   // unconditional (clamped): loads stay back to back
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = b + k * T + threadIdx.x;
                if (i < count) lds[i] = r[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < count; i += T) {
            long long idx = lo + i;
            float s = 0.f;
            if (idx >= 0) {
                if (idx < n_in) s = v[idx];
            } else if (hist && idx >= -(long long)hist_len)
                s = hist[hist_len + idx];
            lds[i] = s;
        }
    }
    __syncthreads();
    const long long o = o0 + threadIdx.x;
    if (o >= nout) return;
    const float* p = lds + (long long)threadIdx.x * D;
    float a = 0.f;
#pragma unroll 8
    for (int j = 0; j < ntaps; ++j) a = __builtin_fmaf(rtaps[j], p[j], a);
    out[o] = a;
}

// The same filter on the f32 matrix cores (v_mfma_f32_16x16x4_f32: an fmaf chain in k order, bit-identical
// to the loop above), as a banded Toeplitz product like k_frontend_mfma but real-valued:
//   Y[i][G] = sum_k A[i][k] B[k][G],  A[i][k] = rtaps[k - i*D] (0 outside),  B[k][G] = x[16*D*G + k]
// = output 16*G + i of the tile.  The wfm audio filter is ~771 taps at decimation 10: 771 LDS reads per output
// on the vector ALUs (LDS-bound, 0.148 ms for 1.7 M outputs) against (771 + 150)/4 K-steps of one MFMA each.
// blockDim.x = 64*nw; tile = 256*nw outputs.  Samples sit in LDS with 4 pad floats after every 16*D (lane
// stride 16*D + 4 floats = 4 * odd: the 16 groups x 4 k-lanes of a B read hit 64 different banks).
// Dynamic LDS: fir_mfma_lds_bytes(D, ntaps, nw).
#define FIR_MFMA_INFLIGHT 12
static inline size_t fir_mfma_lds_bytes(int D, int ntaps, int nw)
{
    const long long TB = 256LL * nw, BK = 16LL * D;
    const long long ksteps = (15LL * D + ntaps + 3) >> 2;
    const long long ntl = (15LL * D + 4 * (ksteps + 8) + 3) & ~3LL;
    const long long npos = (TB - 1) * D + ntaps + 40;
    return (size_t)(ntl + npos + 4 * (npos / BK + 2)) * sizeof(float);
}
__global__ void __launch_bounds__(256) k_fir_fff_mfma(const float* __restrict__ v, const float* __restrict__ hist,
                                                      int hist_len, long long n_in, long long in0, long long nout,
                                                      const float* __restrict__ rtaps, int ntaps, int D,
                                                      float* __restrict__ out)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float* const ldsf = reinterpret_cast<float*>(smem_raw);
    const int T = blockDim.x, nw = T >> 6, TB = nw * 256;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int BK = 16 * D;
    const long long o0 = (long long)blockIdx.x * TB;
    const long long lo = in0 + o0 * D - (ntaps - 1);       // first input sample of the tile
    const int ksteps = (15 * D + ntaps + 3) >> 2;
    const int ntl = (15 * D + 4 * (ksteps + 8) + 3) & ~3;
    const int npos = (TB - 1) * D + ntaps + 40;            // + K rounding and operand-prefetch slack
    float* const smp = ldsf + ntl;

    for (int i = t; i < ntl; i += T) {
        const int idx = i - 15 * D;
        ldsf[i] = (idx >= 0 && idx < ntaps) ? rtaps[idx] : 0.f;
    }
    {
        // position p (sample lo + p) -> smp[p + 4*(p / BK)].  Interior tiles: aligned 16-B loads from the four
        // samples' boundary below lo, up to FIR_MFMA_INFLIGHT per lane in flight before the first is used (the
        // staging is a latency chain, not a bandwidth one: 4-B loads eight at a time took six round trips per
        // tile and 3/4 of the kernel); the pad quotient stepped, not divided.
        const int odd = (int)(lo & 3);
        const long long lo_al = lo - odd;
        const int nch = (npos + odd + 3) >> 2;
        const bool interior = lo_al >= 0 && lo_al + 4LL * nch <= n_in && (reinterpret_cast<uintptr_t>(v) & 15) == 0;
        if (interior) {
            const float4* __restrict__ src4 = reinterpret_cast<const float4*>(v + lo_al);
            int quo = (4 * t) / BK, rem = 4 * t - quo * BK;          // of 4c, c = this lane's chunk
            const int dq = (4 * T) / BK, dr = 4 * T - dq * BK;
            for (int b = 0; b < nch; b += FIR_MFMA_INFLIGHT * T) {
                float4 r[FIR_MFMA_INFLIGHT];
#pragma unroll
                for (int k = 0; k < FIR_MFMA_INFLIGHT; ++k) {
                    // unconditional, index clamped: a load under `if` is an exec-masked block of its own and the
                    // compiler waits for the previous load before it (one HBM round trip per load)
                    const int c = b + k * T + t;
                    r[k] = src4[c < nch ? c : nch - 1];
                }
#pragma unroll
                for (int k = 0; k < FIR_MFMA_INFLIGHT; ++k) {
                    const int c = b + k * T + t;
                    if (c < nch) {
                        const float e4[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int pe = 4 * c + e - odd, re = rem + e - odd;
                            const int qe = quo + (re >= BK ? 1 : 0) - (re < 0 ? 1 : 0);
                            if (pe >= 0 && pe < npos) smp[pe + 4 * qe] = e4[e];
                        }
                    }
                    quo += dq;
                    rem += dr;
                    if (rem >= BK) {
                        rem -= BK;
                        ++quo;
                    }
                }
            }
        } else {
            // tiles at the ends of the call (history before v[0], zeros past the end), unaligned buffers
            int quo = t / BK, rem = t - quo * BK;
            const int dq = T / BK, dr = T - dq * BK;
            for (int p = t; p < npos; p += T) {
                const long long idx = lo + p;
                float s = 0.f;
                if (idx >= 0) {
                    if (idx < n_in) s = v[idx];
                } else if (hist && idx >= -(long long)hist_len)
                    s = hist[hist_len + idx];
                smp[p + 4 * quo] = s;
                quo += dq;
                rem += dr;
                if (rem >= BK) {
                    rem -= BK;
                    ++quo;
                }
            }
        }
    }
    __syncthreads();

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    {
        const int g = lane & 15, kq = lane >> 4;
        const int G = wave * 16 + g;
        const float* ta = ldsf + (15 * D + kq - D * g);    // A[i = g][k]: tap index k - g*D, k = 4*kk + kq
        const float* sb = smp + (BK + 4) * G + kq;         // B[k][col = g]: sample 16*D*G + k
        const int ngrp = ksteps >> 2, tail = ksteps & 3;
        int gi = 0;
        float a[4], an[4], b[4], bn[4];
        auto load = [&](float* av, float* bv) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                av[s4] = ta[4 * s4];
                bv[s4] = sb[4 * s4];
            }
            ta += 16;
            sb += 16;
            if (++gi == D) {                               // 16*D samples done: step over the pad
                gi = 0;
                sb += 4;
            }
        };
        auto mfma4 = [&](const float* av, const float* bv) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4], acc, 0, 0, 0);
        };
        auto mfma_tail = [&](const float* av, const float* bv) {
#pragma unroll
            for (int s4 = 0; s4 < 3; ++s4)
                if (s4 < tail) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s4], bv[s4], acc, 0, 0, 0);
        };
        // two operand sets used alternately: the next group's operands are fetched while this group's MFMAs run,
        // and no register is copied (the f32 MFMA shares the vector pipe: every v_mov costs MFMA time)
        load(a, b);
        int grp = 0;
        for (; grp + 2 <= ngrp; grp += 2) {
            load(an, bn);
            mfma4(a, b);
            load(a, b);
            mfma4(an, bn);
        }
        if (grp < ngrp) {
            load(an, bn);
            mfma4(a, b);
            mfma_tail(an, bn);
        } else
            mfma_tail(a, b);
    }
    // a non-finite input (0 * Inf in the zero-padded products) shows as NaN results: such a wave recomputes its 256
    // outputs over the taps alone, as k_fir_fff_decim does (see frontend_mfma.h: mfma_exact_redo)
    {
        const float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        if (__ballot(!(__builtin_fabsf(s) < __builtin_inff())) != 0ull) {
            const int G = wave * 16 + (lane & 15);
            const float* const rt = ldsf + 15 * D;
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const int p0 = BK * G + (4 * (lane >> 4) + r) * D;
                float a1 = 0.f;
#pragma unroll 1
                for (int j = 0; j < ntaps; ++j) {
                    const int p = p0 + j;
                    a1 = __builtin_fmaf(rt[j], smp[p + 4 * (p / BK)], a1);
                }
                acc[r] = a1;
            }
        }
    }
    // D[row = 4*(lane>>4) + r][col = lane&15] = output 16*col + row of this wave's 256: four consecutive per lane
    const long long o = o0 + wave * 256 + 16 * (lane & 15) + 4 * (lane >> 4);
    if (o + 4 <= nout)
        *reinterpret_cast<float4*>(out + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else
        for (int r = 0; r < 4; ++r)
            if (o + r < nout) out[o + r] = acc[r];
}

// fm_deemph: y[n] = b0 x[n] + b0 x[n-1] + p1 y[n-1] with double accumulators
// (gr::filter::iir_filter_ffd).  One workgroup; thread t owns a contiguous chunk.
// Pass 1 runs the recurrence from y=0 and records (p1^len, y_end); an inclusive
// scan of the affine maps gives every chunk's true start value; pass 2 re-runs
// the chunk with it.  The op order inside a chunk is the sequential one; the
// chunk hand-off differs from it by O(1e-16) relative, far below float32.
#define DEEMPH_THREADS 1024
__global__ void __launch_bounds__(DEEMPH_THREADS) k_deemph(const float* __restrict__ x, long long n, double b0,
                                                           double p1, float* __restrict__ y,
                                                           const FmDeemphState* __restrict__ st_in,
                                                           FmDeemphState* __restrict__ st_out)
{
    __shared__ double sA[DEEMPH_THREADS];  // multiplier of the incoming state
    __shared__ double sB[DEEMPH_THREADS];  // additive part
    const int t = threadIdx.x;
    const long long per = (n + DEEMPH_THREADS - 1) / DEEMPH_THREADS;
    const long long a0 = (long long)t * per, a1 = (a0 + per < n) ? a0 + per : n;
    double yp = 0.0, pw = 1.0;
    const double x_init = st_in->xp, y_init = st_in->yp;   // state left by the previous call (0, 0 at stream start)
    double xp = (a0 > 0 && a0 <= n) ? (double)x[a0 - 1] : x_init;
    for (long long i = a0; i < a1; ++i) {
        double xi = (double)x[i];
        double acc = b0 * xi;
        acc += b0 * xp;
        acc += p1 * yp;
        xp = xi;
        yp = acc;
        pw *= p1;
    }
    sA[t] = pw;
    sB[t] = yp;
    __syncthreads();
    // inclusive scan of y_out = A*y_in + B (composition, left to right)
    for (int off = 1; off < DEEMPH_THREADS; off <<= 1) {
        double A = sA[t], B = sB[t], A2 = 1.0, B2 = 0.0;
        if (t >= off) {
            A2 = sA[t - off];
            B2 = sB[t - off];
        }
        __syncthreads();
        if (t >= off) {
            sA[t] = A * A2;
            sB[t] = A * B2 + B;
        }
        __syncthreads();
    }
    double y0 = (t > 0) ? sA[t - 1] * y_init + sB[t - 1] : y_init;  // state entering this chunk
    yp = y0;
    xp = (a0 > 0 && a0 <= n) ? (double)x[a0 - 1] : x_init;
    for (long long i = a0; i < a1; ++i) {
        double xi = (double)x[i];
        double acc = b0 * xi;
        acc += b0 * xp;
        acc += p1 * yp;
        xp = xi;
        yp = acc;
        y[i] = (float)acc;
    }
    if (a0 < n && a1 == n) {   // owner of the last sample
        st_out->xp = xp;
        st_out->yp = yp;
    }
}

// Fast path of the same recurrence: the pole p1 < 1 forgets its state, so a
// thread that starts `warm` samples early from y = 0 reaches the true state to
// below double rounding (p1^warm < 2^-60) before its own chunk begins.  One
// thread = DEEMPH_CHUNK outputs; any number of workgroups.
#define DEEMPH_CHUNK 16
// Streaming: a chunk whose warm-up would reach before x[0] starts at x[0] from the EXACT state the
// previous call left (st_in; zeros at stream start), the owner of the last sample leaves the state for
// the next call in st_out (a different slot: other workgroups may still be reading st_in).
// A workgroup owns 256 consecutive chunks.  Its inputs (own span + warm-up + the sample before) are staged
// into LDS by coalesced loads and its outputs leave through LDS the same way: a thread walking global
// memory one float at a time paid one load latency per step (warm + 16 dependent steps: 0.047 ms for
// 1.7 M samples).  One pad float per 16 keeps the lanes' stride-16 walks on distinct banks.
// Dynamic LDS: deemph_lds_bytes(warm).
#define DEEMPH_PAD(j) ((j) + ((j) >> 4))
static inline size_t deemph_lds_bytes(int warm)
{
    const int span = 256 * DEEMPH_CHUNK, nst = span + warm + 1;
    return (size_t)(DEEMPH_PAD(nst) + 1 + DEEMPH_PAD(span) + 1) * sizeof(float);
}
__global__ void __launch_bounds__(256) k_deemph_warm(const float* __restrict__ x, long long n, double b0, double p1,
                                                     int warm, float* __restrict__ y,
                                                     const FmDeemphState* __restrict__ st_in,
                                                     FmDeemphState* __restrict__ st_out)
{
    extern __shared__ float dl[];
    const int span = 256 * DEEMPH_CHUNK, nst = span + warm + 1;
    const long long B0 = (long long)blockIdx.x * span;
    const long long g0 = B0 - warm - 1;                    // sample index of staged position 0
    float* const yo = dl + DEEMPH_PAD(nst) + 1;
    for (int j = threadIdx.x; j < nst; j += 256) {
        const long long g = g0 + j;
        dl[DEEMPH_PAD(j)] = (g >= 0 && g < n) ? x[g] : 0.f;
    }
    __syncthreads();
    const long long a0 = B0 + (long long)threadIdx.x * DEEMPH_CHUNK;
    if (a0 < n) {
        const long long a1 = (a0 + DEEMPH_CHUNK < n) ? a0 + DEEMPH_CHUNK : n;
        long long i = a0 - warm;
        double yp = 0.0, xp;
        if (i <= 0) {
            i = 0;
            yp = st_in->yp;
            xp = st_in->xp;
        } else {
            const int jp = (int)(i - 1 - g0);
            xp = (double)dl[DEEMPH_PAD(jp)];
        }
        int j = (int)(i - g0);
        for (; i < a0; ++i, ++j) {
            const double xi = (double)dl[DEEMPH_PAD(j)];
            double acc = b0 * xi;
            acc += b0 * xp;
            acc += p1 * yp;
            xp = xi;
            yp = acc;
        }
        int o = threadIdx.x * DEEMPH_CHUNK;
        for (; i < a1; ++i, ++j, ++o) {
            const double xi = (double)dl[DEEMPH_PAD(j)];
            double acc = b0 * xi;
            acc += b0 * xp;
            acc += p1 * yp;
            xp = xi;
            yp = acc;
            yo[DEEMPH_PAD(o)] = (float)acc;
        }
        if (a1 == n) {
            st_out->xp = xp;
            st_out->yp = yp;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < span; j += 256)
        if (B0 + j < n) y[B0 + j] = yo[DEEMPH_PAD(j)];
}

// rational_resampler_fff: output m = sum_k h[p + k L] x[i - k], p = (m M) mod L,
// i = (m M) / L (m, i = absolute indices since stream start); then multiply_const(volume).
// This call: outputs m0 .. m0+nout-1 from x = the inputs i_base .. i_base+n_in-1 and the hist_len
// inputs before them (zeros before stream start).
// LDS form: the block's input span and the polyphase tap table (one row per phase, steps oldest first, odd row
// pitch) are staged once, so the 35 steps of an output are LDS reads instead of 35 global loads + 35 gathers from
// the tap array (0.048 -> 0.020 ms for 1.6 M outputs).  Same products in the same order.  Dynamic LDS: resample_lds_bytes().
static inline size_t resample_lds_bytes(int ntaps, int L, int M)
{
    const long long nt = (ntaps + L - 1) / L, ntp = nt | 1;
    return (size_t)(L * ntp + 256LL * M / L + nt + 8) * sizeof(float);
}
__global__ void __launch_bounds__(256) k_resample_fff_lds(const float* __restrict__ x, const float* __restrict__ hist,
                                                          int hist_len, long long n_in, long long i_base, long long m0,
                                                          long long nout, const float* __restrict__ taps, int ntaps, int L,
                                                          int M, float volume, float* __restrict__ out)
{
    extern __shared__ float rl[];
    const int nt = (ntaps + L - 1) / L, ntp = nt | 1;
    float* const tapT = rl;              // tapT[p * ntp + j] = h[p + (nt-1-j) L] (0 past the filter)
    float* const xs = rl + L * ntp;
    const long long ml0 = (long long)blockIdx.x * 256;
    const long long m_first = m0 + ml0;
    const long long m_last = m0 + ((ml0 + 256 < nout) ? ml0 + 256 : nout) - 1;
    const long long i_lo = (m_first * (long long)M) / L - i_base - (nt - 1);   // oldest input of the block's first output
    const long long i_hi = (m_last * (long long)M) / L - i_base;               // newest input of its last output
    const int cnt = (int)(i_hi - i_lo + 1);
    for (int j = threadIdx.x; j < cnt; j += 256) {
        const long long idx = i_lo + j;
        float s = 0.0f;
        if (idx >= 0) {
            if (idx < n_in) s = x[idx];
        } else if (hist && idx >= -(long long)hist_len)
            s = hist[hist_len + idx];
        xs[j] = s;
    }
    for (int qd = threadIdx.x; qd < L * nt; qd += 256) {
        const int p = qd / nt, j = qd - p * nt;
        const int ti = p + (nt - 1 - j) * L;
        tapT[p * ntp + j] = ti < ntaps ? taps[ti] : 0.0f;
    }
    __syncthreads();
    const long long ml = ml0 + threadIdx.x;
    if (ml >= nout) return;
    const long long num = (m0 + ml) * (long long)M;
    const int base = (int)(num / L - i_base - (nt - 1) - i_lo);
    const float* tr = tapT + (int)(num % L) * ntp;
    const float* xr = xs + base;
    float a = 0.f;
    for (int j = 0; j < nt; ++j) a = __builtin_fmaf(tr[j], xr[j], a);   // oldest first
    out[ml] = volume * a;
}

__global__ void __launch_bounds__(256) k_resample_fff(const float* __restrict__ x, const float* __restrict__ hist,
                                                      int hist_len, long long n_in, long long i_base, long long m0,
                                                      long long nout, const float* __restrict__ taps, int ntaps, int L,
                                                      int M, float volume, float* __restrict__ out)
{
    const long long ml = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ml >= nout) return;
    const long long m = m0 + ml;
    const long long num = m * (long long)M;
    const long long i = num / L - i_base;
    const int p = (int)(num % L);
    const int nt = (ntaps + L - 1) / L;
    float a = 0.f;
    for (int j = 0; j < nt; ++j) {  // oldest first: tap k = nt-1-j
        const int k = nt - 1 - j;
        const int ti = p + k * L;
        const float t = ti < ntaps ? taps[ti] : 0.0f;
        const long long idx = i - k;
        float s = 0.0f;
        if (idx >= 0) {
            if (idx < n_in) s = x[idx];
        } else if (hist && idx >= -(long long)hist_len)
            s = hist[hist_len + idx];
        a = __builtin_fmaf(t, s, a);
    }
    out[ml] = volume * a;
}

// ---- de-emphasis and audio resampler in ONE launch (round 3) ---------------------------------------------------
// k_deemph_warm + k_resample_fff_lds walked the 1.7 M audio-rate floats twice through HBM with a launch each
// (0.022 + 0.020 ms of the 0.375 ms step; 410 + 6300 workgroups, the second one filling a 3.4-KB transposed tap
// table with a division per entry in every workgroup).  Here a workgroup of 256 threads computes FR_RANGE
// consecutive de-emphasised floats exactly as k_deemph_warm does (thread = 16 outputs, warmed up over `warm` earlier
// inputs, exact state at stream / call start), the first FR_HALO of which repeat the previous workgroup's last ones
// (the resampler's history; positions before the call come from `hist`, the resampler block's own memory), leaves its
// own FR_RANGE - FR_HALO of them in y, and then emits the resampler outputs whose newest input lies in its own span,
// reading the floats back from LDS.  tapT: the polyphase table already transposed by the host
// (tapT[p * ntp + j] = h[p + (nt-1-j) L], 0 past the filter; ntp = nt | 1).  Same operations in the same order
// as the two kernels it replaces.  Dynamic LDS: deemph_resample_lds_bytes(warm, nt, L).
#define FR_RANGE 4096
#define FR_HALO 48      // >= taps per phase - 1 (host checks), multiple of DEEMPH_CHUNK
static inline size_t deemph_resample_lds_bytes(int warm, int nt, int L)
{
    const int nst = FR_RANGE + warm + 1;
    return (size_t)(DEEMPH_PAD(nst) + 1 + DEEMPH_PAD(FR_RANGE) + 1 + L * (nt | 1)) * sizeof(float);
}
__global__ void __launch_bounds__(256) k_deemph_resample(const float* __restrict__ x, long long n, double b0, double p1,
                                                         int warm, float* __restrict__ y,
                                                         const FmDeemphState* __restrict__ st_in,
                                                         FmDeemphState* __restrict__ st_out,
                                                         const float* __restrict__ hist, int hist_len, long long i_base,
                                                         long long m0, long long nout, const float* __restrict__ tapT_g,
                                                         int nt, int L, int M, float volume, float* __restrict__ out)
{
    extern __shared__ float dl[];
    constexpr int OWN = FR_RANGE - FR_HALO;
    const int nst = FR_RANGE + warm + 1, ntp = nt | 1;
    float* const yo = dl + DEEMPH_PAD(nst) + 1;             // the workgroup's FR_RANGE outputs, padded
    float* const tapT = yo + DEEMPH_PAD(FR_RANGE) + 1;
    float* const ys = dl;                                   // ... unpadded, once the inputs are consumed
    const long long B0 = (long long)blockIdx.x * OWN;       // first position this workgroup owns
    const long long C0 = B0 - FR_HALO;                      // first position it computes
    const long long g0 = C0 - warm - 1;                     // sample index of staged position 0
    if (g0 >= 0 && g0 + nst <= n) {
        // interior workgroup (uniform): eight unguarded loads in flight per lane before the first LDS store -- a
        // guarded load -> store per round is a chain of nst / 256 = 17 HBM round trips per workgroup
        const float* __restrict__ xg = x + g0;
        for (int j0 = 0; j0 < nst; j0 += 8 * 256) {
            float r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                r[k] = xg[j < nst ? j : nst - 1];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < nst) dl[DEEMPH_PAD(j)] = r[k];
            }
        }
    } else {
        for (int j = threadIdx.x; j < nst; j += 256) {
            const long long g = g0 + j;
            dl[DEEMPH_PAD(j)] = (g >= 0 && g < n) ? x[g] : 0.f;
        }
    }
    {
        const int ntab = L * ntp;
        for (int j0 = 0; j0 < ntab; j0 += 4 * 256) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                r[k] = tapT_g[j < ntab ? j : ntab - 1];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + k * 256 + (int)threadIdx.x;
                if (j < ntab) tapT[j] = r[k];
            }
        }
    }
    __syncthreads();
    const long long a0 = C0 + (long long)threadIdx.x * DEEMPH_CHUNK;
    if (a0 + DEEMPH_CHUNK <= 0) {
        // positions before this call: the resampler's memory (zeros before stream start)
        int o = threadIdx.x * DEEMPH_CHUNK;
        for (int k = 0; k < DEEMPH_CHUNK; ++k, ++o) {
            const long long idx = a0 + k;
            yo[DEEMPH_PAD(o)] = (hist && idx >= -(long long)hist_len) ? hist[hist_len + idx] : 0.f;
        }
    } else if (a0 < n) {
        const long long a1 = (a0 + DEEMPH_CHUNK < n) ? a0 + DEEMPH_CHUNK : n;
        long long i = a0 - warm;
        double yp = 0.0, xp;
        if (i <= 0) {
            i = 0;
            yp = st_in->yp;
            xp = st_in->xp;
        } else {
            const int jp = (int)(i - 1 - g0);
            xp = (double)dl[DEEMPH_PAD(jp)];
        }
        int j = (int)(i - g0);
        // The recurrence's only serial part is acc += p1 * yp (one multiply, one add).  Eight steps at a time: the
        // eight inputs are read from LDS together and b0*x[i] + b0*x[i-1] is formed for all of them before the
        // chain runs -- a step by step loop paid an LDS round trip and five dependent fp64 operations per step
        // (0.041 ms for 1.7 M floats; the same operations in the same order per sample, so the same bits).
        for (; i + 8 <= a0; i += 8, j += 8) {
            double xs8[8], t8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xs8[k] = (double)dl[DEEMPH_PAD(j + k)];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double acc = b0 * xs8[k];
                acc += b0 * (k ? xs8[k - 1] : xp);
                t8[k] = acc;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double acc = t8[k];
                acc += p1 * yp;
                yp = acc;
            }
            xp = xs8[7];
        }
        for (; i < a0; ++i, ++j) {
            const double xi = (double)dl[DEEMPH_PAD(j)];
            double acc = b0 * xi;
            acc += b0 * xp;
            acc += p1 * yp;
            xp = xi;
            yp = acc;
        }
        int o = threadIdx.x * DEEMPH_CHUNK;
        for (; i + 8 <= a1; i += 8, j += 8, o += 8) {
            double xs8[8], t8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) xs8[k] = (double)dl[DEEMPH_PAD(j + k)];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double acc = b0 * xs8[k];
                acc += b0 * (k ? xs8[k - 1] : xp);
                t8[k] = acc;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                double acc = t8[k];
                acc += p1 * yp;
                yp = acc;
                yo[DEEMPH_PAD(o + k)] = (float)acc;
            }
            xp = xs8[7];
        }
        for (; i < a1; ++i, ++j, ++o) {
            const double xi = (double)dl[DEEMPH_PAD(j)];
            double acc = b0 * xi;
            acc += b0 * xp;
            acc += p1 * yp;
            xp = xi;
            yp = acc;
            yo[DEEMPH_PAD(o)] = (float)acc;
        }
        if (a1 == n && a0 >= B0) {   // the owner of the call's last sample (halo threads repeat a neighbour's work)
            st_out->xp = xp;
            st_out->yp = yp;
        }
    }
    __syncthreads();
    // outputs to memory (own part only) and, unpadded, to ys for the resampler
    for (int j = threadIdx.x; j < FR_RANGE; j += 256) {
        const float v = (C0 + j < n) ? yo[DEEMPH_PAD(j)] : 0.f;
        ys[j] = v;
        if (j >= FR_HALO && C0 + j < n) y[C0 + j] = v;
    }
    __syncthreads();
    // resampler outputs m (absolute) whose newest input floor(m M / L) lies in [i_base + B0, i_base + B1)
    const long long B1 = (B0 + OWN < n) ? B0 + OWN : n;
    if (B1 <= B0) return;
    const long long I0 = i_base + B0, I1 = i_base + B1;
    const long long m_lo = (I0 * L + M - 1) / M, m_hi = (I1 * L + M - 1) / M;   // uniform
    long long m = m_lo + threadIdx.x;
    if (m >= m_hi) return;
    const long long num = m * (long long)M;
    long long iq_ = num / L;
    int p = (int)(num - iq_ * L);
    const int di = (256 * M) / L, dp = (256 * M) % L;
    for (; m < m_hi; m += 256) {
        const int base = (int)(iq_ - i_base - C0) - (nt - 1);   // >= FR_HALO - (nt-1) >= 0
        const float* tr = tapT + p * ntp;
        const float* xr = ys + base;
        float a = 0.f;
#pragma unroll 8
        for (int j = 0; j < nt; ++j) a = __builtin_fmaf(tr[j], xr[j], a);   // oldest first
        const long long ml = m - m0;
        if (ml >= 0 && ml < nout) out[ml] = volume * a;
        iq_ += di;
        p += dp;
        if (p >= L) {
            p -= L;
            ++iq_;
        }
    }
}
