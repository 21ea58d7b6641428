// analog_pwr_squelch_cc between the channel filter and the demodulator
// (TFA/decoder.grc:391-446, TFA/analyzer.grc:544-599) and
// the demodulator + slicer that follows it on the gated (ragged) stream.
//
// p[n] = alpha*|y[n]|^2 + (1-alpha)*p[n-1] in double (single_pole_iir<double,..>),
// item passes iff !(p < thr); gate drops muted items (stream compaction), no gate
// zeroes them.  The recurrence is strictly sequential; here every thread owns a
// contiguous chunk, an affine-map scan hands each chunk its start value and the
// chunk is re-run from it in the sequential op order.  The hand-off differs from
// the sequential value by O(1e-16) relative, which can only matter for an item
// whose p lies within that distance of thr: any item within 1e-9*thr of the
// threshold makes the capture take the exact sequential re-run on thread 0, so
// the pass/mute decisions are always the sequential ones.
#pragma once
#include "sdr_common.h"
#include "frontend_kernels.h"

#define SQ_THREADS 1024

__device__ __forceinline__ float sq_mag2(float2 v) { return v.x * v.x + v.y * v.y; }

__global__ void __launch_bounds__(SQ_THREADS) k_squelch(SquelchParams P)
{
    __shared__ double sA[SQ_THREADS];
    __shared__ double sB[SQ_THREADS];
    __shared__ int sC[SQ_THREADS];
    __shared__ int s_near;
    __shared__ double s_pfinal;
    const int t = threadIdx.x, cap = blockIdx.x;
    const float2* __restrict__ y = P.y + (long long)cap * P.y_stride;
    float2* __restrict__ z = P.z + (long long)cap * P.z_stride;
    SquelchState* st = P.st + cap;
    const double p_in = P.fresh ? 0.0 : st->p;
    const float2 last_in = P.fresh ? make_float2(0.f, 0.f) : st->last;
    const long long n = P.n;
    const double a = P.alpha, b = 1.0 - P.alpha, thr = P.thr;
    if (t == 0) {
        s_near = 0;
        s_pfinal = p_in;
    }
    const long long per = (n + SQ_THREADS - 1) / SQ_THREADS;
    const long long a0 = (long long)t * per < n ? (long long)t * per : n, a1 = (a0 + per < n) ? a0 + per : n;

    // pass 1: chunk as an affine map of its incoming state
    double p = 0.0, pw = 1.0;
    for (long long i = a0; i < a1; ++i) {
        p = a * (double)sq_mag2(y[i]) + b * p;
        pw *= b;
    }
    sA[t] = pw;
    sB[t] = p;
    __syncthreads();
    for (int off = 1; off < SQ_THREADS; off <<= 1) {
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
    const double p0 = (t > 0) ? (sA[t - 1] * p_in + sB[t - 1]) : p_in;

    // pass 2: decisions from the true start value; count what passes
    int keep = 0, near = 0;
    p = p0;
    const double eps = thr * 1e-9;
    for (long long i = a0; i < a1; ++i) {
        p = a * (double)sq_mag2(y[i]) + b * p;
        keep += !(p < thr);
        near |= (__builtin_fabs(p - thr) <= eps);
    }
    if (near) atomicOr(&s_near, 1);
    if (a1 == n && a0 < n) s_pfinal = p;
    sC[t] = P.gate ? keep : (int)(a1 - a0);
    __syncthreads();
    for (int off = 1; off < SQ_THREADS; off <<= 1) {
        int v = sC[t], v2 = (t >= off) ? sC[t - off] : 0;
        __syncthreads();
        sC[t] = v + v2;
        __syncthreads();
    }
    long long w = (t > 0) ? sC[t - 1] : 0;
    const int total = sC[SQ_THREADS - 1];

    if (!s_near) {
        // pass 3: write
        p = p0;
        for (long long i = a0; i < a1; ++i) {
            const float2 v = y[i];
            p = a * (double)sq_mag2(v) + b * p;
            const bool mute = p < thr;
            if (!mute)
                z[w++] = v;
            else if (!P.gate)
                z[w++] = make_float2(0.f, 0.f);
        }
        __syncthreads();
        if (t == 0) {
            z[-1] = last_in;
            st->p = s_pfinal;
            st->last = total > 0 ? z[total - 1] : last_in;
            P.cnt[cap] = total;
        }
    } else if (t == 0) {
        // an item sits on the threshold: the plain sequential statement decides
        p = p_in;
        long long k = 0;
        for (long long i = 0; i < n; ++i) {
            const float2 v = y[i];
            p = a * (double)sq_mag2(v) + b * p;
            const bool mute = p < thr;
            if (!mute)
                z[k++] = v;
            else if (!P.gate)
                z[k++] = make_float2(0.f, 0.f);
        }
        z[-1] = last_in;
        st->p = p;
        st->last = k > 0 ? z[k - 1] : last_in;
        st->near_hits += 1;
        P.cnt[cap] = (int32_t)k;
    }
}

// complex_to_mag_squared / quadrature_demod_cf + multiply_const + threshold_ff +
// binary_slicer + keep_one_in_n on the gated stream: cnt[cap] items per capture,
// z[-1] = the item before this call's first one.  One thread = one item.
__global__ void __launch_bounds__(256) k_demod_slice(DemodParams P)
{
    __shared__ int sh[12];
    const int t = threadIdx.x, tile = blockIdx.x, cap = blockIdx.y;
    const float2* __restrict__ z = P.z + (long long)cap * P.z_stride;
    const long long n = P.cnt[cap];
    const long long o0 = (long long)tile * P.tile_out, o = o0 + t;
    const bool valid = o < n;
    float v = 0.f;
    if (valid) {
        const float2 c = z[o];
        if (P.demod == SDR_DEMOD_MAG2)
            v = __builtin_fmaf(c.x, c.x, c.y * c.y);
        else {
            const float2 pv = z[o - 1];
            const float tr = __builtin_fmaf(c.x, pv.x, c.y * pv.y);
            const float ti = __builtin_fmaf(c.y, pv.x, -(c.x * pv.y));
            v = fast_atan2f_dev(P.atan_tab, ti, tr);
        }
    }
    v = P.gain * v;
    if (valid && P.tap) (P.tap + (long long)cap * P.tap_stride)[o] = v;
    if (!P.do_slice) return;
    int8_t* bits = P.bits + (long long)cap * P.bits_stride;
    slice_tile(valid, v, o, o0, P.lo, P.hi, P.keep_n, P.keep_c0, bits,
               P.tsum + (long long)cap * P.tiles_per_capture + tile, sh);
    // rows of a ragged batch are zero beyond the capture's own length
    if (!valid && o < P.nmax && o >= P.keep_c0 && (o - P.keep_c0) % P.keep_n == 0)
        bits[(o - P.keep_c0) / P.keep_n] = 0;
}
