// Back-end kernel: one workgroup per capture (or per stream call).
//   phase 0  resolve the hysteresis carry of every front-end tile and patch the
//            bits that were emitted before a tile's first event;
//   phase 1  run-length: positions where the sliced byte stream changes level
//            (per-thread count -> block scan -> compaction);
//   phase 2  the stream decoder's state machine over the run widths and the
//            per-packet decode.
// Integer / float64 work on a few hundred symbols per capture: latency-bound,
// not bandwidth-bound; its traffic (1/(D*keep_n) B per input sample) is noise
// beside the front end's 8 B.
#pragma once
#include <stddef.h>
#include "sdr_common.h"

#define BK_THREADS 256
#define SDR_HD __host__ __device__

SDR_HD inline double run_width(const int32_t* __restrict__ pos, long long k, double rate)
{
    // np.diff(edge_positions)*1e6/self.sample_rate (TS33C/stream_decoder.py:74)
    return (double)(pos[k + 1] - pos[k]) * 1e6 / rate;
}

// ---- numpy helpers restated on the device (float64, IEEE ops only) -----------

SDR_HD inline void insertion_sort(double* a, int n)
{
    for (int i = 1; i < n; ++i) {
        double v = a[i];
        int j = i - 1;
        while (j >= 0 && a[j] > v) {
            a[j + 1] = a[j];
            --j;
        }
        a[j + 1] = v;
    }
}

// np.percentile(x, q*100) method 'linear' on sorted s (numpy/lib/_function_base_impl.py _quantile/_lerp)
SDR_HD inline double pct_sorted(const double* s, int n, double q)
{
    double vi = (double)(n - 1) * q;
    double fl = floor(vi);
    int prev = (int)fl, next = prev + 1;
    double g = vi - fl;
    if (vi >= (double)(n - 1)) prev = next = n - 1;
    if (vi < 0) prev = next = 0;
    if (next > n - 1) next = n - 1;
    double a = s[prev], b = s[next];
    double d = b - a;
    return (g >= 0.5) ? (b - d * (1.0 - g)) : (a + d * g);
}

// edges[edges.size//2] of np.histogram(x, "auto")[1]  (numpy 2.2.6 _hist_bin_auto,
// _get_bin_edges, linspace).  s = sorted copy of x.
SDR_HD inline double hist_auto_mid_edge(const double* s, int n, const double* __restrict__ pow_m13,
                                            const double* __restrict__ log2n)
{
    if (n == 0) return 1.0;  // empty -> one bin on [0, 1]
    double first = s[0], last = s[n - 1];
    double ptp = last - first;
    double iqr = pct_sorted(s, n, 0.75) - pct_sorted(s, n, 0.25);
    int nt = n < SDR_MAX_POW_TABLE ? n : SDR_MAX_POW_TABLE - 1;
    double fd = 2.0 * iqr * pow_m13[nt];
    double stg = ptp / (log2n[nt] + 1.0);
    double width = (fd != 0.0) ? ((stg < fd) ? stg : fd) : stg;
    if (first == last) {
        first = first - 0.5;
        last = last + 0.5;
    }
    long long nb = 1;
    if (width != 0.0) nb = (long long)ceil((last - first) / width);
    if (nb < 1) nb = 1;
    long long idx = (nb + 1) / 2;
    if (idx == nb) return last;
    double delta = last - first;
    double step = delta / (double)nb;
    if (step == 0.0) return ((double)idx / (double)nb) * delta + first;
    return (double)idx * step + first;
}

// np.add.reduce pairwise summation for n < 128 contiguous float64 (numpy/_core/src/umath/loops_utils.h.src)
SDR_HD inline double np_pairwise_sum_small(const double* a, int n)
{
    if (n < 8) {
        double res = 0.0;  // numpy: res = 0.; for i: res += a[i]  (n<8 branch starts from 0.)
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

SDR_HD inline void frame_init(sdr_frame* f, int cap, int dec, long long end_pos)
{
    // every byte of the record, pad2_ and all, then the three fields that are not zero: field-by-field
    // assignment let the compiler merge neighbouring stores across the hole in front of `lim` and fill it with
    // whatever the register held (seen as records that differed in two bytes under a HIP-graph replay)
    __builtin_memset(f, 0, sizeof *f);
    f->capture = cap;
    f->decoder = dec;
    f->end_pos = end_pos;
}

// ---- TS33C packet decode (TS33C/stream_decoder.py:112-202) -------------------
SDR_HD inline void decode_ts33c(const double* cur, int n, double reset_limit, double* scratch,
                                    DecState* st, const BackParams& P, sdr_frame* f)
{
    int np_ = (n + 1) / 2, ng = n / 2;
    // pulses = cur[0::2] -> histogram
    for (int i = 0; i < np_; ++i) scratch[i] = cur[2 * i];
    insertion_sort(scratch, np_);
    double plim = hist_auto_mid_edge(scratch, np_, P.pow_m13, P.log2n);
    // gaps = cur[1::2], packet pause -> 0, histogram over gaps > 0
    int ngf = 0;
    for (int i = 0; i < ng; ++i) {
        double g = cur[2 * i + 1];
        if (g > reset_limit) g = 0.0;
        if (g > 0.0) scratch[ngf++] = g;
    }
    insertion_sort(scratch, ngf);
    double glim = hist_auto_mid_edge(scratch, ngf, P.pow_m13, P.log2n);
    st->pulse_short_limit = plim;
    st->gap_short_limit = glim;
    f->lim[0] = plim;
    f->lim[1] = glim;

    // rawBits -> parity / packet bits; bytes and per-byte parity on the fly
    int nraw = 0, ndata = 0, npar = 0, ones = 0;
    bool par_ok = true;
    unsigned char bytes[16];
    for (int i = 0; i < 16; ++i) bytes[i] = 0;
    for (int k = 0; k < n; ++k) {
        int bit;
        if ((k & 1) == 0) {
            bit = (cur[k] < plim) ? 1 : 0;
        } else {
            double g = cur[k];
            if (g > reset_limit) g = 0.0;
            if (g < glim) continue;  // short gap (255) removed
            bit = 0;
        }
        ++nraw;
        if (nraw % 9 == 0) {
            // parity bit of the byte just completed: True iff popcount(byte) is even
            bool even = (ones & 1) == 0;
            if (even != (bit == 1)) par_ok = false;
            ones = 0;
            ++npar;
        } else {
            int byte = ndata >> 3, j = ndata & 7;
            if (!bit) {  // reverse + invert
                ++ones;
                if (byte < 16) bytes[byte] |= (unsigned char)(1u << j);
            }
            ++ndata;
        }
    }
    f->nbits = nraw;
    if ((ndata % 8) != 0) {
        f->status = SDR_FRAME_BAD_LENGTH;
        return;
    }
    int nbytes = ndata / 8;
    f->nbytes = nbytes < 16 ? nbytes : 16;
    for (int i = 0; i < f->nbytes; ++i) f->bytes[i] = bytes[i];
    // np.array_equal(packetParity, parityBits): same length and all equal
    if (npar != nbytes || !par_ok) {
        f->status = SDR_FRAME_PARITY;
        return;
    }
    if (!(nbytes == 10 && bytes[0] == 0x9f)) {
        f->status = SDR_FRAME_UNKNOWN_HEADER;
        return;
    }
    int channel = (bytes[1] >> 5) & 0x0F;
    if (channel >= 5) channel -= 1;
    int temp = (bytes[5] & 0x0F) * 100 + ((bytes[4] & 0xF0) >> 4) * 10 + (bytes[4] & 0x0F);
    int sign = (((bytes[5] >> 7) & 0x01) * 2) - 1;
    f->v[0] = channel;
    f->v[1] = bytes[1] & 0x0F;
    f->v[2] = ((bytes[5] >> 6) & 0x01) == 1;
    f->v[3] = ((bytes[6] & 0xF0) >> 4) * 10 + (bytes[6] & 0x0F);
    f->temp_tenths = temp * sign;
    f->status = SDR_FRAME_OK;
}

// ---- TFA packet decode (TFA/stream_decoder.py:132-199, analyzer2.grc:891-960) --
SDR_HD inline unsigned char crc8_31(const unsigned char* d, int n)
{
    unsigned int crc = 0;
    for (int i = 0; i < n; ++i) {
        crc ^= d[i];
        for (int b = 0; b < 8; ++b) crc = (crc & 0x80) ? (((crc << 1) ^ 0x31) & 0xFF) : ((crc << 1) & 0xFF);
    }
    return (unsigned char)crc;
}

SDR_HD inline void decode_tfa(const double* cur0, int n0, double* scratch, bool nan_guard, sdr_frame* f)
{
    // self.currentSymbols = self.currentSymbols[6:]
    const double* c = cur0 + (n0 < 6 ? n0 : 6);
    int n = n0 < 6 ? 0 : n0 - 6;
    int ns = n < 150 ? n : 150;
    if (ns & 1) {
        f->status = SDR_FRAME_EXC_RESHAPE;
        return;
    }
    int npair = ns / 2;
    if (npair == 0) {
        f->lim[0] = __builtin_nan("");   // np.mean of an empty sync array
        f->status = nan_guard ? SDR_FRAME_SYMBOL_LENGTH : SDR_FRAME_EXC_NAN;
        return;
    }
    for (int i = 0; i < npair; ++i) scratch[i] = c[2 * i] + c[2 * i + 1];
    double sym = np_pairwise_sum_small(scratch, npair) / (double)npair;
    f->lim[0] = sym;
    double thr = sym * 1.5;
    int ds = -1;
    for (int k = 0; k < n; ++k)
        if (c[k] > thr) {
            ds = k;
            break;
        }
    if (ds < 0) {
        f->status = SDR_FRAME_EXC_INDEX;
        return;
    }
    unsigned char bytes[11];
    for (int i = 0; i < 11; ++i) bytes[i] = 0;
    long long nbits = 0;
    for (int k = ds; k < n - 1; ++k) {
        double w = c[k];
        long long cnt;
        int one = ((k - ds) & 1) == 0;
        if (one) {
            if (w < sym) w = 0.0;
            cnt = (long long)floor(w / sym);
        } else
            cnt = (long long)ceil(w / sym);
        if (one) {
            for (long long i = 0; i < cnt && nbits + i < 88; ++i) {
                long long b = nbits + i;
                bytes[b >> 3] |= (unsigned char)(1u << (b & 7));  // per-byte bit reversal == LSB first
            }
        }
        nbits += cnt;
    }
    f->nbits = (int)(nbits > 0x7fffffff ? 0x7fffffff : nbits);
    if (nbits < 88) {
        f->status = SDR_FRAME_BAD_LENGTH;
        return;
    }
    f->nbytes = 11;
    for (int i = 0; i < 11; ++i) f->bytes[i] = bytes[i];
    if (crc8_31(bytes + 2, 8) != bytes[10]) {
        f->status = SDR_FRAME_CRC;
        return;
    }
    f->v[0] = ((int)bytes[2] << 8) + bytes[3];
    f->v[1] = bytes[6] & 0x7f;
    f->v[2] = ((bytes[7] >> 7) & 0x01) == 0;
    f->temp_tenths = (bytes[4] & 0x0f) * 100 + ((bytes[5] >> 4) & 0x0f) * 10 + (bytes[5] & 0x0f) - 400;
    f->status = SDR_FRAME_OK;
}

#if defined(__HIPCC__)
// Inclusive prefix sum over the 64 lanes in the VALU (DPP row shifts inside the 16-lane rows,
// then the row totals by row_bcast) instead of six ds_bpermute round trips.
__device__ __forceinline__ int wave_incl_scan_dpp(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

#endif

// ---- pigpio-callback rule decoders on run boundaries ---------------------------
struct EdgeCtx {
    const int32_t* pos;
    const long long* ticks;  // optional precomputed ticks (LDS); nullptr -> from pos
    long long R;
    long long nb;
    int first_level;
    long long rate_hz;
    int wave;                // device: all 64 lanes of the wavefront run the rule (EdgeReader may broadcast)
    long long base;          // streaming: decoder-rate samples consumed before this call (ticks are absolute)
    RuleState* rs;           // streaming: the callback's state between calls (nullptr: one call = one capture)
};
SDR_HD inline long long tick_of(const EdgeCtx& e, long long k)
{
    if (e.ticks) return e.ticks[k];
    return ((e.base + (long long)e.pos[k]) * 1000000LL) / e.rate_hz;
}

// What a pigpio callback object keeps between edges (somfy/sniff.py:19-26, berner/sniff.py:17-19,
// mumbi/sniff.py:19-24, kwmobile/decoder.py:87-93, baldr/decoder.py:79-85), as local variables of one
// rule run.  One call = one capture: they start fresh.  Streaming (e.rs): loaded from / stored to the
// handle's RuleState, so the rule sees ONE continuous edge stream however the caller cuts it.
struct RuleVars {
    long long start_tick = 0, prev_tick = 0;
    int state = 0, nbits = 0, level = 0, n_sync = 0, clock = 640;
    double sync_sum = 0.0;
};
SDR_HD inline void rule_sync(const EdgeCtx& e)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.wave) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#endif
}
SDR_HD inline void rule_load(const EdgeCtx& e, RuleVars& v, unsigned char* bits)
{
    if (!e.rs || !e.rs->valid) return;
    v.start_tick = e.rs->start_tick;
    v.prev_tick = e.rs->prev_tick;
    v.state = e.rs->state;
    v.nbits = e.rs->nbits;
    v.level = e.rs->level;
    v.n_sync = e.rs->n_sync;
    v.clock = e.rs->clock;
    v.sync_sum = e.rs->sync_sum;
    if (bits) {
        int i0 = 0, step = 1;
#if defined(__HIP_DEVICE_COMPILE__)
        if (e.wave) {
            i0 = (int)(threadIdx.x & 63);
            step = 64;
        }
#endif
        for (int i = i0; i < 512; i += step) bits[i] = e.rs->bits[i];
        rule_sync(e);
    }
}
SDR_HD inline void rule_store(const EdgeCtx& e, const RuleVars& v, int level, const unsigned char* bits)
{
    if (!e.rs) return;
    rule_sync(e);
    if (bits) {
        int i0 = 0, step = 1;
#if defined(__HIP_DEVICE_COMPILE__)
        if (e.wave) {
            i0 = (int)(threadIdx.x & 63);
            step = 64;
        }
#endif
        const int n = v.nbits < 512 ? v.nbits : 512;
        for (int i = i0; i < n; i += step) e.rs->bits[i] = bits[i];
    }
    e.rs->start_tick = v.start_tick;
    e.rs->prev_tick = v.prev_tick;
    e.rs->state = v.state;
    e.rs->nbits = v.nbits;
    e.rs->level = level;
    e.rs->n_sync = v.n_sync;
    e.rs->clock = v.clock;
    e.rs->sync_sum = v.sync_sum;
    e.rs->base = e.base + e.nb;
    e.rs->valid = 1;
}

// Edge k's tick and position for the rule loops.  On the device, when the whole wavefront
// runs the rule, 64 edges at a time sit one per lane in registers and are handed out with
// v_readlane (uniform index): no LDS round trip inside the sequential loop.
struct EdgeReader {
    const EdgeCtx& e;
    long long c0, treg;
    int32_t preg;
    SDR_HD explicit EdgeReader(const EdgeCtx& e_) : e(e_), c0(-1), treg(0), preg(0) {}
    SDR_HD void get(long long k, long long& tick, long long& at)
    {
#if defined(__HIP_DEVICE_COMPILE__)
        if (e.wave) {
            if (c0 < 0 || k < c0 || k >= c0 + 64) {
                c0 = k & ~63LL;
                const long long idx = c0 + (long long)(threadIdx.x & 63);
                treg = idx < e.R ? tick_of(e, idx) : 0;
                preg = idx <= e.R ? e.pos[idx] : 0;
            }
            const int l = (int)(k - c0);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)treg, l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)treg >> 32), l);
            tick = (long long)(((unsigned long long)hi << 32) | lo);
            at = __builtin_amdgcn_readlane(preg, l);
            return;
        }
#endif
        tick = tick_of(e, k);
        at = e.pos[k];
    }
};

// bits[0..n) (one 0/1 byte each, n <= 64) -> bytes, first bit = MSB of out[0] (np.packbits).  Built in
// registers and stored once: OR-ing bit by bit into the frame record in HBM is a dependent global
// read-modify-write per bit (32 of them were 2/3 of the berner back end).  On the device, when the whole
// wavefront runs the rule, lane i looks at bit i and one ballot gives the packed word.
SDR_HD inline void pack_bits_msb(const EdgeCtx& e, const unsigned char* bits, int n, unsigned char* out)
{
    const int nby = (n + 7) >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.wave) {
        const int lane = (int)(threadIdx.x & 63);
        const unsigned long long m = __ballot(lane < n && bits[lane] != 0);
        for (int j = 0; j < nby; ++j) out[j] = __builtin_bitreverse8((unsigned char)((m >> (8 * j)) & 0xffu));
        return;
    }
#endif
    for (int j = 0; j < nby; ++j) out[j] = 0;
    for (int i = 0; i < n; ++i)
        if (bits[i]) out[i >> 3] |= (unsigned char)(0x80u >> (i & 7));
}

// Differences of ticks are never negative (ticks ascend) and every limit they are compared with is far
// below 2^31: clamped to INT_MAX they compare exactly as the 64-bit values would, and the rule's tests
// become 32-bit scalar compares (gfx950 has no 64-bit scalar ordering compare: each 64-bit in_range was
// a VALU compare feeding a scalar branch).
SDR_HD inline int sat_delta(long long d) { return d > 0x7fffffffLL ? 0x7fffffff : (int)d; }
SDR_HD inline bool in_range(int d, int a, int b) { return d >= a && d < b; }
SDR_HD inline bool in_range(long long d, long long a, long long b) { return d >= a && d < b; }

// The decoder this workgroup runs: P.decoder, or -- sdr_redecode_multi, one launch for several decoders --
// its entry of dec_list (wave-uniform; P itself stays an unmodified kernel argument in scalar registers).
SDR_HD inline int dec_of(const BackParams& P)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (P.ndec) return (int)((P.dec_pack >> (4u * (blockIdx.x / (unsigned)P.vrows))) & 15u);
#endif
    return P.decoder;
}

SDR_HD inline sdr_frame* next_frame(const BackParams& P, DecState* st, int cap, long long end_pos)
{
    if (st->n_frames >= P.max_frames) {
        st->overflow |= 2;
        return nullptr;
    }
    sdr_frame* f = P.frames + (long long)cap * P.max_frames + st->n_frames;
    st->n_frames++;
    frame_init(f, cap, dec_of(P), end_pos);
    return f;
}

// somfy/sniff.py:84-117 on the collected half-symbols b[0..112): Manchester pairs -> 7 bytes ->
// de-obfuscation -> checksum -> fields
SDR_HD inline void somfy_emit(sdr_frame* f, const unsigned char* fr_in, int clock)
{
    unsigned char fr[7];
    for (int i = 0; i < 7; ++i) fr[i] = fr_in[i];
    for (int i = 0; i < 7; ++i) f->raw[i] = fr[i];
    for (int i = 6; i > 0; --i) fr[i] = fr[i] ^ fr[i - 1];
    unsigned int ck = 0;
    for (int i = 0; i < 7; ++i) ck ^= fr[i] ^ (fr[i] >> 4);
    ck &= 0xf;
    f->nbits = 112;
    f->nbytes = 7;
    for (int i = 0; i < 7; ++i) f->bytes[i] = fr[i];
    f->v[0] = (fr[1] >> 4) & 0x0f;
    f->v[1] = ((int)fr[2] << 8) | fr[3];
    f->v[2] = (int)fr[4] | ((int)fr[5] << 8) | ((int)fr[6] << 16);
    f->v[3] = clock;
    f->status = ck == 0 ? SDR_FRAME_OK : SDR_FRAME_CHECKSUM;
}

#if defined(__HIP_DEVICE_COMPILE__)
// somfy rule (somfy/sniff.py:28-117), wavefront form.  The time since the previous edge (every event
// restarts the rule's clock) is classified for 64 edges at a time, one per lane: watchdog (> 20 ms),
// wake-up pulse, hardware-sync length, software-sync pulse.  Outside the data state only those few edges
// are visited (the others cannot change anything there); in the data state the stretch up to the next
// watchdog is handled in one step: every lane knows how many half-symbols its edge appends (two, one or
// none, by the clock measured from the sync), a prefix sum places them in the bit buffer.
__device__ inline void wave_rule_somfy(const EdgeCtx& e, const BackParams& P, DecState* st, int cap, unsigned char* bits)
{
    const int lane = (int)(threadIdx.x & 63);
    const int tol = 100, wd = 20000;
    int state = 0, nbits = 0, clock = 640, n_sync = 0;
    double sync_sum = 0.0;
    auto watchdog = [&](long long at_run) {
        if (state > 0) {
            int n = nbits > 0 ? nbits - 1 : 0;
            const int b0 = nbits > 0 ? 1 : 0;
            if (n < 112 && n + 1 < 512) bits[b0 + n++] = 0;
            if (n == 112) {
                sdr_frame* f = next_frame(P, st, cap, at_run);
                const unsigned long long m = __ballot(lane < 56 && bits[b0 + 2 * lane] == 0);
                if (f) {
                    unsigned char fr[7];
                    for (int j = 0; j < 7; ++j) fr[j] = __builtin_bitreverse8((unsigned char)((m >> (8 * j)) & 0xffu));
                    somfy_emit(f, fr, clock);
                }
            }
            nbits = 0;
            state = 0;
        }
    };
    for (long long j0 = 0; 1 + j0 < e.R; j0 += 64) {
        const long long k = 1 + j0 + lane;
        const bool in = k < e.R;
        const int d = in ? sat_delta(tick_of(e, k) - tick_of(e, k - 1)) : 0;
        const bool falling = in && (int)(k & 1) == e.first_level;        // end of a pulse
        const unsigned long long mIn = __ballot(in);
        unsigned long long mW = __ballot(in && d > wd);
        const unsigned long long mK = __ballot(falling && in_range(d, 10050 - tol, 10050 + tol));
        const unsigned long long mH = __ballot(in && in_range(d, 2500 - 2 * tol, 2500 + 2 * tol));
        const unsigned long long mS = __ballot(falling && in_range(d, 4850 - 2 * tol, 4850 + 2 * tol));
        const int nin = __builtin_popcountll(mIn);
        int cur = 0;
        while (cur < nin) {
            const unsigned long long below = (1ull << cur) - 1ull;
            const unsigned long long wrest = mW & ~below;
            const int w = wrest ? __builtin_ctzll(wrest) : nin;           // next watchdog edge (or end of chunk)
            if (state == 3) {
                // data: edges [cur, w) append 2 / 1 / 0 half-symbols of their level
                const bool mine = in && lane >= cur && lane < w;
                int c = 0;
                if (mine) c = in_range(d, 2 * clock - tol, 2 * clock + tol) ? 2 : (in_range(d, clock - tol, clock + tol) ? 1 : 0);
                const int incl = wave_incl_scan_dpp(c);
                const int at = nbits + incl - c;
                const unsigned char v = falling ? 1 : 0;
                if (c >= 1 && at < 512) bits[at] = v;
                if (c == 2 && at + 1 < 512) bits[at + 1] = v;
                const int total = __builtin_amdgcn_readlane(incl, 63);
                nbits = nbits + total > 512 ? 512 : nbits + total;
                cur = w;
            } else {
                // idle / woken / syncing: only wake-up, sync-length and software-sync edges matter
                const unsigned long long cand = (mK | mH | mS) & ~below & (w >= 64 ? ~0ull : ((1ull << w) - 1ull));
                if (!cand)
                    cur = w;
                else {
                    const int p = __builtin_ctzll(cand);
                    const unsigned long long bit = 1ull << p;
                    const int dp = __builtin_amdgcn_readlane(d, p);
                    if (mH & bit) {                                      // either edge, any state <= 2
                        if (state < 2) {
                            n_sync = 0;
                            sync_sum = 0.0;
                        }
                        sync_sum += (double)dp;
                        ++n_sync;
                        state = 2;
                    } else if ((mK & bit) && state == 0)
                        state = 1;
                    else if ((mS & bit) && state == 2) {
                        clock = (int)((sync_sum / (double)n_sync) / 4.0);
                        nbits = 0;
                        state = 3;
                    }
                    cur = p + 1;
                    continue;
                }
            }
            if (w < nin) {
                watchdog(e.pos[1 + j0 + w]);
                mW &= ~(1ull << w);     // the edge itself is then an ordinary edge of the next stretch
                cur = w;
            }
        }
    }
    const long long end_tick = (e.nb * 1000000LL) / e.rate_hz;
    const long long last_tick = e.R > 1 ? tick_of(e, e.R - 1) : 0;
    if (sat_delta(end_tick - last_tick) > wd) watchdog(e.nb);
}
#endif

// somfy/sniff.py:28-117
SDR_HD inline void rule_somfy(const EdgeCtx& e, const BackParams& P, DecState* st, int cap,
                                  unsigned char* bits /* >= 512 */)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.wave && !e.rs) {
        wave_rule_somfy(e, P, st, cap, bits);
        return;
    }
#endif
    const int tol = 100;
    RuleVars rv;
    rule_load(e, rv, bits);
    long long& start_tick = rv.start_tick;
    int &state = rv.state, &nbits = rv.nbits, &clock = rv.clock, &n_sync = rv.n_sync;
    double& sync_sum = rv.sync_sum;  // np.average of <8 values: sequential sum
    const int wd = 20000;
    long long& prev_tick = rv.prev_tick;
    auto watchdog = [&](long long at_run) {
        if (state > 0) {
            // bits = bits[1:]; pad one zero if short
            int n = nbits > 0 ? nbits - 1 : 0;
            unsigned char* b = bits + (nbits > 0 ? 1 : 0);
            if (n < 112 && n + 1 < 512) b[n++] = 0;
            if (n == 112) {
                sdr_frame* f = next_frame(P, st, cap, at_run);
                if (f) {
                    unsigned char fr[7];
                    for (int i = 0; i < 7; ++i) fr[i] = 0;
                    // np.where(reshape(bits,(-1,2)) == [0,1], 1, 0).ravel()[::2]: first half-symbol == 0
                    for (int i = 0; i < 56; ++i)
                        if (b[2 * i] == 0) fr[i >> 3] |= (unsigned char)(0x80u >> (i & 7));
                    somfy_emit(f, fr, clock);
                }
            }
            nbits = 0;
            state = 0;
        }
    };
    auto push = [&](int v, int cnt) {
        for (int i = 0; i < cnt; ++i)
            if (nbits < 512) bits[nbits++] = (unsigned char)v;
    };
    auto event = [&](int level, long long tick) {
        const int delta = sat_delta(tick - start_tick);
        start_tick = tick;
        if (level == 1) {  // end of gap
            if (state <= 2 && in_range(delta, 2500 - 2 * tol, 2500 + 2 * tol)) {
                if (state < 2) {
                    n_sync = 0;
                    sync_sum = 0.0;
                }
                sync_sum += (double)delta;
                ++n_sync;
                state = 2;
            } else if (state == 3 && in_range(delta, 2 * clock - tol, 2 * clock + tol))
                push(0, 2);
            else if (state == 3 && in_range(delta, clock - tol, clock + tol))
                push(0, 1);
        } else {  // end of pulse
            if (state == 0 && in_range(delta, 10050 - tol, 10050 + tol))
                state = 1;
            else if (state <= 2 && in_range(delta, 2500 - 2 * tol, 2500 + 2 * tol)) {
                if (state < 2) {
                    n_sync = 0;
                    sync_sum = 0.0;
                }
                sync_sum += (double)delta;
                ++n_sync;
                state = 2;
            } else if (state == 2 && in_range(delta, 4850 - 2 * tol, 4850 + 2 * tol)) {
                clock = (int)((sync_sum / (double)n_sync) / 4.0);
                nbits = 0;
                state = 3;
            } else if (state == 3 && in_range(delta, 2 * clock - tol, 2 * clock + tol))
                push(1, 2);
            else if (state == 3 && in_range(delta, clock - tol, clock + tol))
                push(1, 1);
        }
    };
    int level = rv.level;   // level before the row's first byte (0 at stream start)
    EdgeReader rd(e);
    for (long long k = (e.first_level != level) ? 0 : 1; k < e.R; ++k) {
        long long tick, at;
        rd.get(k, tick, at);
        if (sat_delta(tick - prev_tick) > wd) watchdog(at);
        level ^= 1;
        event(level, tick);
        prev_tick = tick;
    }
    long long end_tick = ((e.base + e.nb) * 1000000LL) / e.rate_hz;
    if (sat_delta(end_tick - prev_tick) > wd) watchdog(e.nb);
    rule_store(e, rv, level, bits);
}

#if defined(__HIP_DEVICE_COMPILE__)
// berner rule, wavefront form: the pulse width of 64 falling edges at a time is classified one per
// lane (preamble / "0" / "1" / other), the classes come back as ballot masks and the state machine
// of berner/sniff.py:21-44 walks them with bit scans -- per frame, not per edge: from idle, the next
// preamble; from there the run of "0"/"1" pulses up to the first pulse that is neither (which ends
// the frame and is not looked at again, exactly as the sequential rule leaves it).
__device__ inline void wave_rule_berner(const EdgeCtx& e, const BackParams& P, DecState* st, int cap)
{
    const int lane = (int)(threadIdx.x & 63);
    const long long kf0 = e.first_level ? 1 : 2;   // first falling edge (pos[0] = 0 is the row start)
    int state = 0, nbits = 0;
    unsigned acc = 0;                              // first 32 bits, stream bit i at bit i
    for (long long j0 = 0; kf0 + 2 * j0 < e.R; j0 += 64) {
        const long long k = kf0 + 2 * (j0 + lane);
        const bool in = k < e.R;
        const int d = in ? sat_delta(tick_of(e, k) - tick_of(e, k - 1)) : 0;
        const unsigned long long mIn = __ballot(in);
        const unsigned long long mP = __ballot(in && d > 5300 && d < 5700);
        const unsigned long long mO = __ballot(in && d > 500 && d < 1000);
        const unsigned long long mZO = mO | __ballot(in && d < 500);
        const int nin = __builtin_popcountll(mIn);
        int cur = 0;
        while (cur < nin) {
            if (state == 0) {
                const unsigned long long m = mP >> cur;
                if (!m) break;
                cur += __builtin_ctzll(m) + 1;
                state = 1;
                nbits = 0;
                acc = 0;
            } else {
                const unsigned long long stop = (~mZO & mIn) >> cur;
                const int run = stop ? __builtin_ctzll(stop) : nin - cur;
                if (nbits < 32 && run > 0) {
                    const int take = run < 32 - nbits ? run : 32 - nbits;
                    const unsigned v = (unsigned)((mO >> cur) & ((1ull << take) - 1ull));
                    acc |= v << nbits;
                }
                nbits = nbits + run > 512 ? 512 : nbits + run;
                cur += run;
                if (stop) {
                    if (nbits >= 32) {
                        sdr_frame* f = next_frame(P, st, cap, e.pos[kf0 + 2 * (j0 + cur)]);
                        if (f) {
                            for (int j = 0; j < 4; ++j) f->bytes[j] = __builtin_bitreverse8((unsigned char)((acc >> (8 * j)) & 0xffu));
                            f->nbits = nbits;
                            f->nbytes = 4;
                            f->status = SDR_FRAME_OK;
                        }
                    }
                    state = 0;
                    ++cur;
                }
            }
        }
    }
}
#endif

// berner/sniff.py:21-44
SDR_HD inline void rule_berner(const EdgeCtx& e, const BackParams& P, DecState* st, int cap,
                                   unsigned char* bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.wave && !e.rs) {
        wave_rule_berner(e, P, st, cap);
        return;
    }
#endif
    RuleVars rv;
    rule_load(e, rv, bits);
    long long& start_tick = rv.start_tick;
    int &state = rv.state, &nbits = rv.nbits;
    auto event = [&](int lvl, long long tick, long long at) {
        if (lvl == 1)
            start_tick = tick;
        else {
            const int delta = sat_delta(tick - start_tick);
            if (state == 0 && delta > 5300 && delta < 5700) {
                nbits = 0;
                state = 1;
            } else if (state == 1 && delta < 500) {
                if (nbits < 512) bits[nbits++] = 0;
            } else if (state == 1 && delta > 500 && delta < 1000) {
                if (nbits < 512) bits[nbits++] = 1;
            } else if (state == 1) {
                if (nbits >= 32) {
                    sdr_frame* f = next_frame(P, st, cap, at);
                    if (f) {
                        unsigned char by[4];
                        pack_bits_msb(e, bits, 32, by);
                        for (int j = 0; j < 4; ++j) f->bytes[j] = by[j];
                        f->nbits = nbits;
                        f->nbytes = 4;
                        f->status = SDR_FRAME_OK;
                    }
                }
                state = 0;
            }
        }
    };
    int level = rv.level;   // level before the row's first byte (0 at stream start)
    EdgeReader rd(e);
    for (long long k = (e.first_level != level) ? 0 : 1; k < e.R; ++k) {
        level ^= 1;
        long long tick, at;
        rd.get(k, tick, at);
        event(level, tick, at);
    }
    rule_store(e, rv, level, bits);
}

#if defined(__HIP_DEVICE_COMPILE__)
// mumbi rule (mumbi/sniff.py:26-65), wavefront form: 64 edges at a time, one per lane -- the time since
// the previous edge decides both the 8-ms watchdog (any edge) and, on falling edges, the pulse class
// ("1": 720..920 us, sets the receiving state; "0": 250..350 us).  The masks are walked from watchdog
// to watchdog: the pulses in between append their bits (only the first 34 are ever looked at), the
// watchdog emits when exactly 34 were collected in the receiving state.  As in the sequential rule,
// "0" pulses seen before any "1" stay in the buffer across a watchdog.
__device__ inline void wave_rule_mumbi(const EdgeCtx& e, const BackParams& P, DecState* st, int cap)
{
    const int lane = (int)(threadIdx.x & 63);
    const int wd = 8000;
    int state = 0, nbits = 0;
    unsigned long long acc = 0;   // stream bit i at bit i (i < 34)
    auto watchdog = [&](long long at) {
        if (state > 0) {
            if (nbits == 34) {
                sdr_frame* f = next_frame(P, st, cap, at);
                if (f) {
                    for (int j = 0; j < 5; ++j) f->bytes[j] = __builtin_bitreverse8((unsigned char)((acc >> (8 * j)) & 0xffu));
                    f->nbits = 34;
                    f->nbytes = 5;
                    f->status = SDR_FRAME_OK;
                }
            }
            nbits = 0;
            acc = 0;
            state = 0;
        }
    };
    for (long long j0 = 0; 1 + j0 < e.R; j0 += 64) {
        const long long k = 1 + j0 + lane;
        const bool in = k < e.R;
        const int d = in ? sat_delta(tick_of(e, k) - tick_of(e, k - 1)) : 0;
        const bool falling = in && (int)(k & 1) == e.first_level;
        const unsigned long long mIn = __ballot(in);
        unsigned long long mW = __ballot(in && d > wd);
        const unsigned long long mA = __ballot(falling && in_range(d, 720, 920));
        const unsigned long long mAB = mA | __ballot(falling && in_range(d, 250, 350));
        const int nin = __builtin_popcountll(mIn);
        int cur = 0;
        while (cur < nin) {
            const unsigned long long wrest = mW >> cur;
            const int w = wrest ? cur + __builtin_ctzll(wrest) : nin;
            // pulses of edges [cur, w)
            const unsigned long long range = (w >= 64 ? ~0ull : ((1ull << w) - 1ull)) & ~((1ull << cur) - 1ull);
            unsigned long long ab = mAB & range;
            if (mA & range) state = 1;
            while (ab && nbits < 34) {
                const int p = __builtin_ctzll(ab);
                acc |= ((mA >> p) & 1ull) << nbits;
                ++nbits;
                ab &= ab - 1ull;
            }
            nbits += __builtin_popcountll(ab);
            if (nbits > 512) nbits = 512;
            if (w < nin) {
                watchdog(e.pos[1 + j0 + w]);
                mW &= ~(1ull << w);
            }
            cur = w;
            if (w >= nin) break;
        }
    }
    const long long end_tick = (e.nb * 1000000LL) / e.rate_hz;
    const long long last_tick = e.R > 1 ? tick_of(e, e.R - 1) : 0;
    if (sat_delta(end_tick - last_tick) > wd) watchdog(e.nb);
}
#endif

// mumbi/sniff.py:26-65 (watchdog 8 ms)
SDR_HD inline void rule_mumbi(const EdgeCtx& e, const BackParams& P, DecState* st, int cap,
                                  unsigned char* bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.wave && !e.rs) {
        wave_rule_mumbi(e, P, st, cap);
        return;
    }
#endif
    const int tol = 50;
    const int wd = 8000;
    RuleVars rv;
    rule_load(e, rv, bits);
    long long &start_tick = rv.start_tick, &prev_tick = rv.prev_tick;
    int &state = rv.state, &nbits = rv.nbits;
    auto watchdog = [&](long long at) {
        if (state > 0) {
            if (nbits == 34) {
                sdr_frame* f = next_frame(P, st, cap, at);
                if (f) {
                    unsigned char by[5];
                    pack_bits_msb(e, bits, 34, by);
                    for (int j = 0; j < 5; ++j) f->bytes[j] = by[j];
                    f->nbits = 34;
                    f->nbytes = 5;
                    f->status = SDR_FRAME_OK;
                }
            }
            nbits = 0;
            state = 0;
        }
    };
    auto event = [&](int lvl, long long tick) {
        if (lvl == 1)
            start_tick = tick;
        else {
            const int delta = sat_delta(tick - start_tick);
            if (in_range(delta, 820 - 2 * tol, 820 + 2 * tol)) {
                if (nbits < 512) bits[nbits++] = 1;
                state = 1;
            } else if (in_range(delta, 300 - tol, 300 + tol)) {
                if (nbits < 512) bits[nbits++] = 0;
            }
        }
    };
    int level = rv.level;   // level before the row's first byte (0 at stream start)
    EdgeReader rd(e);
    for (long long k = (e.first_level != level) ? 0 : 1; k < e.R; ++k) {
        long long tick, at;
        rd.get(k, tick, at);
        if (sat_delta(tick - prev_tick) > wd) watchdog(at);
        level ^= 1;
        event(level, tick);
        prev_tick = tick;
    }
    long long end_tick = ((e.base + e.nb) * 1000000LL) / e.rate_hz;
    if (sat_delta(end_tick - prev_tick) > wd) watchdog(e.nb);
    rule_store(e, rv, level, bits);
}

// kwmobile/decoder.py:112-131 fields of the 36-bit frame (5 packed bytes)
SDR_HD inline void kwmobile_fill(sdr_frame* f, const unsigned char* b)
{
    for (int j = 0; j < 5; ++j) f->bytes[j] = b[j];
    f->nbits = 36;
    f->nbytes = 5;
    f->temp_tenths = ((b[1] & 0x0f) << 8) | b[2];
    f->v[0] = b[0];                                   // sensor_id
    f->v[1] = ((b[3] & 0x0f) << 4) + (b[4] >> 4);     // humidity
    f->v[2] = (b[1] & 0x80) == 0x80;                  // battery_ok
    f->v[3] = (b[1] & 0x30) >> 4;                     // channel
    f->status = SDR_FRAME_OK;
}

// baldr/decoder.py:118-139 fields of the 37-bit frame
SDR_HD inline void baldr_fill(sdr_frame* f, const unsigned char* b)
{
    for (int j = 0; j < 5; ++j) f->bytes[j] = b[j];
    f->nbits = 37;
    f->nbytes = 5;
    int t = ((int)b[2] << 4) | (b[3] >> 4);
    if (t > 0x7ff) t -= 4096;
    f->temp_tenths = t;
    f->v[0] = ((b[0] & 0x0f) << 4) | (b[1] >> 4);     // sensor_id
    f->v[1] = ((b[3] & 0x07) << 4) | (b[4] >> 4);     // humidity
    f->v[2] = b[1] & 0x04;                            // txMode
    f->v[3] = b[1] & 0x03;                            // channel
    f->status = SDR_FRAME_OK;
}

#if defined(__HIP_DEVICE_COMPILE__)
// PPM rules (kwmobile / baldr), wavefront form.  Only rising edges carry information: the time since
// the previous rising edge is a preamble, a "1" or a "0"; the time since the previous edge of either
// kind drives the watchdog.  64 edges per step, one per lane; the masks are walked from watchdog to
// watchdog.  BALDR: a second preamble-length gap ends the frame (kept for the watchdog if exactly
// 37 bits were collected, dropped otherwise), as baldr/decoder.py:104-110.
template <bool BALDR>
__device__ inline void wave_rule_ppm(const EdgeCtx& e, const BackParams& P, DecState* st, int cap)
{
    const int lane = (int)(threadIdx.x & 63);
    const int wd = BALDR ? 18000 : 3000, want = BALDR ? 37 : 36;
    const int pre_lo = BALDR ? 9600 - 200 : 4500 - 200, pre_hi = BALDR ? 9600 + 200 : 4500 + 200;
    const int one_lo = BALDR ? 4700 - 200 : 2500 - 100, one_hi = BALDR ? 4700 + 200 : 2500 + 100;
    const int zer_lo = BALDR ? 2650 - 200 : 1500 - 100, zer_hi = BALDR ? 2650 + 200 : 1500 + 100;
    int state = 0, nbits = 0;
    unsigned long long acc = 0;   // stream bit i at bit i (first 64)
    auto watchdog = [&](long long at) {
        if (state > 0) {
            if (nbits == want) {
                sdr_frame* f = next_frame(P, st, cap, at);
                if (f) {
                    unsigned char b[5];
                    for (int j = 0; j < 5; ++j) b[j] = __builtin_bitreverse8((unsigned char)((acc >> (8 * j)) & 0xffu));
                    if (BALDR)
                        baldr_fill(f, b);
                    else
                        kwmobile_fill(f, b);
                }
            }
            nbits = 0;
            acc = 0;
            state = 0;
        }
    };
    auto append = [&](unsigned long long ones, unsigned long long any) {   // pulses of one stretch, in edge order
        while (any && nbits < 64) {
            const int p = __builtin_ctzll(any);
            acc |= ((ones >> p) & 1ull) << nbits;
            ++nbits;
            any &= any - 1ull;
        }
        nbits += __builtin_popcountll(any);
        if (nbits > 512) nbits = 512;
    };
    for (long long j0 = 0; 1 + j0 < e.R; j0 += 64) {
        const long long k = 1 + j0 + lane;
        const bool in = k < e.R;
        const long long tk = in ? tick_of(e, k) : 0;
        const int d1 = in ? sat_delta(tk - tick_of(e, k - 1)) : 0;
        const bool rising = in && (int)(k & 1) != e.first_level;
        const int dr = rising ? sat_delta(tk - (k >= 2 ? tick_of(e, k - 2) : 0)) : 0;
        const unsigned long long mIn = __ballot(in);
        unsigned long long mW = __ballot(in && d1 > wd);
        const unsigned long long mP = __ballot(rising && in_range(dr, pre_lo, pre_hi));
        const unsigned long long m1 = __ballot(rising && in_range(dr, one_lo, one_hi));
        const unsigned long long m01 = m1 | __ballot(rising && in_range(dr, zer_lo, zer_hi));
        const int nin = __builtin_popcountll(mIn);
        int cur = 0;
        while (cur < nin) {
            const unsigned long long below = (1ull << cur) - 1ull;
            const unsigned long long wrest = mW & ~below;
            const int w = wrest ? __builtin_ctzll(wrest) : nin;
            const unsigned long long upto = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
            const unsigned long long pre = mP & ~below & upto;
            if (state == 0) {
                if (pre) {
                    state = 1;
                    cur = __builtin_ctzll(pre) + 1;
                    continue;
                }
            } else if (state == 1) {
                if (BALDR && pre) {
                    const int p = __builtin_ctzll(pre);
                    append(m1, m01 & ~below & ((1ull << p) - 1ull));
                    if (nbits == want)
                        state = 2;
                    else {
                        nbits = 0;
                        acc = 0;
                        state = 0;
                    }
                    cur = p + 1;
                    continue;
                }
                append(m1, m01 & ~below & upto);
            }
            if (w < nin) {
                watchdog(e.pos[1 + j0 + w]);
                mW &= ~(1ull << w);
            }
            cur = w;
        }
    }
    const long long end_tick = (e.nb * 1000000LL) / e.rate_hz;
    const long long last_tick = e.R > 1 ? tick_of(e, e.R - 1) : 0;
    if (sat_delta(end_tick - last_tick) > wd) watchdog(e.nb);
}
#endif

// kwmobile/decoder.py:95-135 (Nexus-type PPM sensor; watchdog 3 ms, :145).  Only the
// rising edges matter: delta = time between consecutive ends of gap.
SDR_HD inline void rule_kwmobile(const EdgeCtx& e, const BackParams& P, DecState* st, int cap,
                                 unsigned char* bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.wave && !e.rs) {
        wave_rule_ppm<false>(e, P, st, cap);
        return;
    }
#endif
    const int tol = 50;
    const int wd = 3000;
    RuleVars rv;
    rule_load(e, rv, bits);
    long long &start_tick = rv.start_tick, &prev_tick = rv.prev_tick;
    int &state = rv.state, &nbits = rv.nbits;
    auto watchdog = [&](long long at) {
        if (state > 0) {
            if (nbits == 36) {
                sdr_frame* f = next_frame(P, st, cap, at);
                if (f) {
                    unsigned char b[5];
                    pack_bits_msb(e, bits, 36, b);
                    kwmobile_fill(f, b);
                }
            }
            nbits = 0;
            state = 0;
        }
    };
    auto event = [&](int lvl, long long tick) {
        if (lvl != 1) return;
        const int delta = sat_delta(tick - start_tick);
        start_tick = tick;
        if (state == 0 && in_range(delta, 4500 - 4 * tol, 4500 + 4 * tol))
            state = 1;
        else if (state == 1 && in_range(delta, 2500 - 2 * tol, 2500 + 2 * tol)) {
            if (nbits < 512) bits[nbits++] = 1;
        } else if (state == 1 && in_range(delta, 1500 - 2 * tol, 1500 + 2 * tol)) {
            if (nbits < 512) bits[nbits++] = 0;
        }
    };
    int level = rv.level;   // level before the row's first byte (0 at stream start)
    EdgeReader rd(e);
    for (long long k = (e.first_level != level) ? 0 : 1; k < e.R; ++k) {
        long long tick, at;
        rd.get(k, tick, at);
        if (sat_delta(tick - prev_tick) > wd) watchdog(at);
        level ^= 1;
        event(level, tick);
        prev_tick = tick;
    }
    long long end_tick = ((e.base + e.nb) * 1000000LL) / e.rate_hz;
    if (sat_delta(end_tick - prev_tick) > wd) watchdog(e.nb);
    rule_store(e, rv, level, bits);
}

// baldr/decoder.py:87-142 (watchdog 18 ms, :152): one decode per transmission, of the
// first complete 37-bit frame that follows an inter-frame gap.
SDR_HD inline void rule_baldr(const EdgeCtx& e, const BackParams& P, DecState* st, int cap,
                              unsigned char* bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (e.wave && !e.rs) {
        wave_rule_ppm<true>(e, P, st, cap);
        return;
    }
#endif
    const int tol = 200;
    const int wd = 18000;
    const int medium = 2650, lng = 4700, verylong = 9600;
    RuleVars rv;
    rule_load(e, rv, bits);
    long long &start_tick = rv.start_tick, &prev_tick = rv.prev_tick;
    int &state = rv.state, &nbits = rv.nbits;
    auto watchdog = [&](long long at) {
        if (state > 0) {
            if (nbits == 37) {
                sdr_frame* f = next_frame(P, st, cap, at);
                if (f) {
                    unsigned char b[5];
                    pack_bits_msb(e, bits, 37, b);
                    baldr_fill(f, b);
                }
            }
            nbits = 0;
            state = 0;
        }
    };
    auto event = [&](int lvl, long long tick) {
        if (lvl != 1) return;
        const int delta = sat_delta(tick - start_tick);
        start_tick = tick;
        if (state == 0 && in_range(delta, verylong - tol, verylong + tol))
            state = 1;
        else if (state == 1 && in_range(delta, lng - tol, lng + tol)) {
            if (nbits < 512) bits[nbits++] = 1;
        } else if (state == 1 && in_range(delta, medium - tol, medium + tol)) {
            if (nbits < 512) bits[nbits++] = 0;
        } else if (state == 1 && in_range(delta, verylong - tol, verylong + tol)) {
            if (nbits == 37)
                state = 2;
            else {
                nbits = 0;
                state = 0;
            }
        }
    };
    int level = rv.level;   // level before the row's first byte (0 at stream start)
    EdgeReader rd(e);
    for (long long k = (e.first_level != level) ? 0 : 1; k < e.R; ++k) {
        long long tick, at;
        rd.get(k, tick, at);
        if (sat_delta(tick - prev_tick) > wd) watchdog(at);
        level ^= 1;
        event(level, tick);
        prev_tick = tick;
    }
    long long end_tick = ((e.base + e.nb) * 1000000LL) / e.rate_hz;
    if (sat_delta(end_tick - prev_tick) > wd) watchdog(e.nb);
    rule_store(e, rv, level, bits);
}

// velux/decoder.py:30-101 (io-homecontrol): the sliced stream is one byte per symbol.
// Find the start of frame "0 11111111 1 0 11001100 1", drop the UART start/stop bits of
// every 10-bit group, bytes LSB first (the script reverses every byte except the last
// group it holds), message length = control1 & 0x1f, CRC-16/KERMIT over bytes
// [2, 3+len) against the little-endian word behind them.  One frame per call (the
// script decodes the first start of frame only).  The packet (<= 36 bytes) is stored
// from sdr_frame.bytes on through .raw and .v (40 contiguous bytes).
SDR_HD inline void rule_velux(const EdgeCtx& e, const BackParams& P, DecState* st, int cap)
{
    static_assert(offsetof(sdr_frame, raw) == offsetof(sdr_frame, bytes) + 16 &&
                      offsetof(sdr_frame, v) == offsetof(sdr_frame, bytes) + 24,
                  "velux packet spans bytes/raw/v");
    const unsigned int SOF = 0x7FD99u;   // 0 11111111 1 0 11001100 1
    long long k = 0;                     // run index of bit i
    int level = e.first_level;
    unsigned int win = 0;
    long long start = -1;
    long long i = 0;
    for (; i < e.nb; ++i) {
        while (k + 1 < e.R && i >= e.pos[k + 1]) {
            ++k;
            level ^= 1;
        }
        win = ((win << 1) | (unsigned)level) & 0xFFFFFu;
        if (i >= 19 && win == SOF) {
            start = i - 19;
            break;
        }
    }
    if (start < 0) return;
    sdr_frame* f = next_frame(P, st, cap, e.nb);
    if (!f) return;
    // rewind the run pointer to `start`
    k = 0;
    level = e.first_level;
    const long long nrem = e.nb - start;
    const long long q = nrem / 10, r = nrem % 10;
    const long long L = 8 * q + (r > 0 ? r - 1 : 0);   // data bits the script keeps
    unsigned char* pk = f->bytes;                        // 40 contiguous bytes
    for (int b = 0; b < 40; ++b) pk[b] = 0;
    for (long long j = 0; j < nrem && j < 400; ++j) {
        const long long ii = start + j;
        while (k + 1 < e.R && ii >= e.pos[k + 1]) {
            ++k;
            level ^= 1;
        }
        const int t = (int)(j % 10) - 1;
        if (t < 0 || t > 7) continue;
        const long long by = j / 10;
        const bool reversed = 8 * by < L - 8;
        if (level) pk[by] |= (unsigned char)(reversed ? (1u << t) : (0x80u >> t));
    }
    f->nbits = (int)(L > 0x7fffffff ? 0x7fffffff : L);
    f->lim[0] = (double)start;
    if (L < 24) {
        f->nbytes = (int)((L + 7) / 8);
        f->status = SDR_FRAME_EXC_SHORT;
        return;
    }
    const int control1 = pk[2], ml = control1 & 0x1f, plen = 3 + ml + 2;
    f->pad_ = control1;
    f->lim[1] = (double)ml;
    if (L < 8LL * plen) {
        const long long have = (L + 7) / 8;
        f->nbytes = (int)(have < 40 ? have : 40);
        f->status = SDR_FRAME_EXC_SHORT;
        return;
    }
    for (int b = plen; b < 40; ++b) pk[b] = 0;   // strip trailing noise
    // a partial trailing bit group never reaches into the packet here (L >= 8*plen), but the
    // un-reversed last group can: the script's quirk is kept by `reversed` above
    unsigned int crc = 0;
    for (int b = 2; b < 3 + ml; ++b) {
        crc ^= pk[b];
        for (int s2 = 0; s2 < 8; ++s2) crc = (crc & 1u) ? ((crc >> 1) ^ 0x8408u) : (crc >> 1);
    }
    const unsigned int got = (unsigned)pk[3 + ml] | ((unsigned)pk[3 + ml + 1] << 8);
    f->nbytes = plen;
    f->temp_tenths = ml >= 16 ? (((int)pk[3 + ml - 8] << 8) | pk[3 + ml - 7]) : 0;   // counter
    f->status = (crc == got) ? (ml >= 16 ? SDR_FRAME_OK : SDR_FRAME_EXC_SHORT) : SDR_FRAME_CRC;
}

// ---- the stream decoder front end (TS33C/stream_decoder.py:39-110,204-217) ----
// W(k) = k-th run width in us; `cur` = currentSymbols working array (LDS on the
// device), `st` = a private copy of the persistent state.
struct WidthsArray {
    const double* w;
    SDR_HD double operator()(long long k) const { return w[k]; }
};
struct WidthsFromPos {
    const int32_t* pos;
    double rate;
    SDR_HD double operator()(long long k) const { return run_width(pos, k, rate); }
};

template <class WF>
SDR_HD inline void stream_decode(const BackParams& P, int cap, WF W, const int32_t* pos, long long R,
                                 int first_level, int last_level, bool any_one, double* cur,
                                 DecState* st, double* scratch)
{
    const bool ts33c = (dec_of(P) == SDR_DEC_TS33C);
    const double reset_limit = ts33c ? 1550.0 : 500.0;
    // skip empty or zero-only chunks (:50)
    if (!(P.nb > 0 && (any_one || st->n_cur > 0))) {
        st->state = 0;
        return;
    }
    const int cur_type = first_level;
    st->chunk_id += 1;
    // pieces end after every width > reset_limit (:77)
    long long n_over = 0;
    for (long long k = 0; k < R; ++k)
        if (W(k) > reset_limit) ++n_over;
    const long long npieces = n_over + 1;
    const bool size_is_1 = (npieces == 1 && R == 1);
    if (npieces > 1) st->state = 1;
    long long a = 0;
    for (long long pc = 0; pc < npieces; ++pc) {
        long long b = a;
        if (pc == npieces - 1)
            b = R;
        else {
            while (W(b) <= reset_limit) ++b;
            ++b;  // the over-limit width closes the piece
        }
        long long s = a;
        if (size_is_1 || st->state == 0) ++s;  // skip leading inter-frame gap (:90-92)
        if (s < b) {
            st->state = 1;
            if (st->n_cur > 0 && st->prev_type == cur_type) {  // (:98-100)
                cur[st->n_cur - 1] += W(s);
                ++s;
            }
            for (long long k = s; k < b; ++k) {
                if (st->n_cur < SDR_CUR_CAP)
                    cur[st->n_cur++] = W(k);
                else
                    st->overflow |= 4;
            }
            if (cur[0] > reset_limit) {  // (:104-105)
                for (int i = 1; i < st->n_cur; ++i) cur[i - 1] = cur[i];
                st->n_cur--;
            }
            if (st->n_cur > 1 && cur[st->n_cur - 1] > reset_limit) {  // (:110)
                sdr_frame* f = next_frame(P, st, cap, pos[b]);
                if (f) {
                    if (ts33c)
                        decode_ts33c(cur, st->n_cur, reset_limit, scratch, st, P, f);
                    else
                        decode_tfa(cur, st->n_cur, scratch, dec_of(P) == SDR_DEC_TFA_GRC, f);
                }
                st->n_cur = 0;  // done with this packet (:204)
            }
        } else {
            st->state = 0;  // "skipping empty chunk" (:207-209)
        }
        a = b;
    }
    st->prev_type = last_level;  // rawBuffer[-2] (:212)
}

