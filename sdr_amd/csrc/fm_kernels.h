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
    for (int i = threadIdx.x; i < count; i += T) {
        long long idx = lo + i;
        float s = 0.f;
        if (idx >= 0) {
            if (idx < n_in) s = v[idx];
        } else if (hist && idx >= -(long long)hist_len)
            s = hist[hist_len + idx];
        lds[i] = s;
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
__global__ void __launch_bounds__(256) k_deemph_warm(const float* __restrict__ x, long long n, double b0, double p1,
                                                     int warm, float* __restrict__ y,
                                                     const FmDeemphState* __restrict__ st_in,
                                                     FmDeemphState* __restrict__ st_out)
{
    const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long a0 = c * DEEMPH_CHUNK;
    if (a0 >= n) return;
    const long long a1 = (a0 + DEEMPH_CHUNK < n) ? a0 + DEEMPH_CHUNK : n;
    long long i = a0 - warm;
    double yp = 0.0, xp;
    if (i <= 0) {
        i = 0;
        yp = st_in->yp;
        xp = st_in->xp;
    } else
        xp = (double)x[i - 1];
    for (; i < a1; ++i) {
        double xi = (double)x[i];
        double acc = b0 * xi;
        acc += b0 * xp;
        acc += p1 * yp;
        xp = xi;
        yp = acc;
        if (i >= a0) y[i] = (float)acc;
    }
    if (a1 == n) {
        st_out->xp = xp;
        st_out->yp = yp;
    }
}

// rational_resampler_fff: output m = sum_k h[p + k L] x[i - k], p = (m M) mod L,
// i = (m M) / L (m, i = absolute indices since stream start); then multiply_const(volume).
// This call: outputs m0 .. m0+nout-1 from x = the inputs i_base .. i_base+n_in-1 and the hist_len
// inputs before them (zeros before stream start).
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
