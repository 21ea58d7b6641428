// Somfy wake-up (preamble) finder block, SURVEY.md 8a row a12
// (somfy/epy_block_1.py:14-67): a 3-state machine over work() chunks of the
// sliced byte stream.  Per chunk it needs the byte sum (which the reference
// accumulates in int8, so it wraps), the first non-zero index, and one of three
// output fills (marker / np.gradient(x, 0.5) cast to bytes / constant).  One
// 1024-thread workgroup per call; the block state lives in HBM.
#pragma once
#include "sdr_common.h"

struct PreambleState {
    int32_t state;
    int32_t preamble;   // int8 arithmetic of the reference, kept widened
    long long position;
    // what this call did, for the Python mirror's print lines
    int32_t events;     // bit0 "position of preamble", bit1 state-1 (len, sum) line, bit2 "length of preamble", bit3 gradient on < 2 items
    int32_t sum8;
    long long preamble_pos;
};

#define PRE_THREADS 1024

__device__ __forceinline__ int wrap8(int v) { return ((v + 128) & 0xff) - 128; }

__global__ void __launch_bounds__(PRE_THREADS) k_preamble(const int8_t* __restrict__ x, long long n,
                                                          int8_t* __restrict__ out, PreambleState* st)
{
    __shared__ int s_sum[PRE_THREADS / 64];
    __shared__ long long s_first[PRE_THREADS / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int sum = 0;
    long long first = n;
    for (long long i = t; i < n; i += PRE_THREADS) {
        const int v = x[i];
        sum += v;
        if (v != 0 && i < first) first = i;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off);
        const long long f2 = __shfl_down(first, off);
        first = f2 < first ? f2 : first;
    }
    if (lane == 0) {
        s_sum[wv] = sum;
        s_first[wv] = first;
    }
    __syncthreads();
    sum = 0;
    first = n;
    for (int w = 0; w < PRE_THREADS / 64; ++w) {
        sum += s_sum[w];
        first = s_first[w] < first ? s_first[w] : first;
    }
    const int s8 = wrap8(sum);
    const PreambleState in = *st;   // every thread reads the state before thread 0 rewrites it
    __syncthreads();
    PreambleState o = in;
    o.position = in.position + n;
    o.events = 0;
    o.sum8 = s8;
    int mode;  // 0 zeros(+marker), 1 ones, 2 gradient
    if (in.state == 0) {
        mode = 0;
        if (s8 > 0) {
            o.preamble = s8;
            o.preamble_pos = o.position - s8;
            o.events |= 1;
            o.state = 1;
        } else
            first = -1;  // no marker
    } else if (in.state == 1) {
        o.events |= 2;
        o.preamble = wrap8(in.preamble + s8);
        if ((long long)s8 < n) {
            o.state = 2;
            o.events |= 4;
            mode = 2;
            if (n < 2) {
                o.events |= 8;  // np.gradient raises: nothing is written
                mode = 3;
            }
        } else
            mode = 1;
    } else
        mode = 1;
    if (t == 0) *st = o;
    if (mode == 0) {
        for (long long i = t; i < n; i += PRE_THREADS) out[i] = (i == first) ? 1 : 0;
    } else if (mode == 1) {
        for (long long i = t; i < n; i += PRE_THREADS) out[i] = 1;
    } else if (mode == 2) {
        for (long long i = t; i < n; i += PRE_THREADS) {
            int g;
            if (i == 0)
                g = 2 * ((int)x[1] - (int)x[0]);
            else if (i == n - 1)
                g = 2 * ((int)x[n - 1] - (int)x[n - 2]);
            else
                g = (int)x[i + 1] - (int)x[i - 1];
            out[i] = (int8_t)g;
        }
    }
}
