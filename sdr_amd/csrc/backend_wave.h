// Wave-parallel form of the stream decoder for the device: the same steps as
// the sequential stream_decode / decode_ts33c / decode_tfa in backend_kernels.h
// (which remain the host-testable statement of the algorithm), executed by the
// 64 lanes of one wavefront on LDS-resident symbols:
//   * piece boundaries and packet-complete tests by ballot,
//   * np.histogram "auto" order statistics by an O(n) per-lane rank sort,
//   * symbol -> bit classification, short-gap removal and TFA run expansion by
//     ballot / shuffle prefix sums, bytes assembled with LDS atomic OR.
// Every function is called by all 64 lanes with wave-uniform arguments.
#pragma once
#include "backend_kernels.h"

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    return (1ull << (threadIdx.x & 63)) - 1ull;
}

struct WaveLds {
    double* cur;            // currentSymbols [SDR_CUR_CAP]
    double* scratch;        // [SDR_CUR_CAP]
    unsigned int* words;    // [64]  packed data bits
    unsigned char* par;     // [256] parity bits
};

__device__ __forceinline__ double readlane_f64(double x, int l)
{
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}

// v[lane ^ J] through the vector ALU instead of the LDS crossbar (ds_bpermute: ~120 cycles per exchange for a lone
// wavefront): DPP quad_perm / row_half_mirror / row_mirror for J = 1, 2, 4, 8, gfx950's v_permlane16_swap /
// v_permlane32_swap for J = 16, 32 (tools/micro/xorlane.hip checks all six against __shfl_xor on the GPU).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int J>
__device__ __forceinline__ int xor_lane_i32(int v)
{
    if (J == 1) return dpp_i32<0xB1>(v);                      // quad_perm [1,0,3,2]
    if (J == 2) return dpp_i32<0x4E>(v);                      // quad_perm [2,3,0,1]
    if (J == 4) return dpp_i32<0x1B>(dpp_i32<0x141>(v));      // row_half_mirror (^7), quad_perm [3,2,1,0] (^3)
    if (J == 8) return dpp_i32<0x141>(dpp_i32<0x140>(v));     // row_mirror (^15), row_half_mirror (^7)
    const int lane = threadIdx.x & 63;
    if (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (lane & 32) ? r[0] : r[1];
}
template <int J>
__device__ __forceinline__ double xor_lane_f64(double x)
{
    const long long b = __builtin_bit_cast(long long, x);
    const unsigned lo = (unsigned)xor_lane_i32<J>((int)(unsigned)b);
    const unsigned hi = (unsigned)xor_lane_i32<J>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
// one compare-exchange round of the 128-slot bitonic network at distance J < 64 inside the merge of size K
template <int K, int J>
__device__ __forceinline__ void bitonic_round(double& a0, double& a1, int lane)
{
    const double b0 = xor_lane_f64<J>(a0), b1 = xor_lane_f64<J>(a1);
    const bool lower = (lane & J) == 0;
    // slot index e = lane (h = 0) or 64 + lane (h = 1): ascending iff (e & K) == 0
    const bool asc0 = (lane & K) == 0, asc1 = ((64 + lane) & K) == 0;
    a0 = (lower == asc0) ? fmin(a0, b0) : fmax(a0, b0);
    a1 = (lower == asc1) ? fmin(a1, b1) : fmax(a1, b1);
}
template <int K, int J>
__device__ __forceinline__ void bitonic_merge(double& a0, double& a1, int lane)
{
    if constexpr (J == 64) {   // partner = the lane's other slot; K == 128: ascending
        const double lo = fmin(a0, a1), hi = fmax(a0, a1);
        a0 = lo;
        a1 = hi;
    } else {
        bitonic_round<K, J>(a0, a1, lane);
    }
    if constexpr (J > 1) bitonic_merge<K, J / 2>(a0, a1, lane);
}

// dst[rank] = value for every valid element; returns the number of valid ones.
// Candidates are held one per lane and broadcast with v_readlane (uniform lane index):
// no LDS round trip inside the O(n^2/64) loop -- the first version read every
// candidate from LDS and made the sort ~14 us of single-wave latency per call.
template <class V, class OK>
__device__ inline int wave_rank_sort(V val, OK ok, int n, double* dst)
{
    const int lane = threadIdx.x & 63;
    if (n <= 128) {
        // Bitonic network over 128 slots, two per lane (element e = 64 h + lane), invalid slots
        // hold +inf and sort to the end: 28 compare-exchange rounds of v_min/v_max_f64 instead
        // of an O(n^2/64) rank count (measured: 12 us -> ~1 us of single-wave latency for the 73
        // pulses of a TS33C packet).  The sorted array is the same whatever the order of ties.
        const double inf = __builtin_inf();
        const bool m0 = lane < n && ok(lane), m1 = lane + 64 < n && ok(lane + 64);
        double a0 = m0 ? val(lane) : inf, a1 = m1 ? val(lane + 64) : inf;
        const int total = __builtin_popcountll(__ballot(m0)) + __builtin_popcountll(__ballot(m1));
        // (round 3: the 27 exchanges between lanes by DPP / permlane instead of ds_bpermute, the rounds unrolled)
        bitonic_merge<2, 1>(a0, a1, lane);
        bitonic_merge<4, 2>(a0, a1, lane);
        bitonic_merge<8, 4>(a0, a1, lane);
        bitonic_merge<16, 8>(a0, a1, lane);
        bitonic_merge<32, 16>(a0, a1, lane);
        bitonic_merge<64, 32>(a0, a1, lane);
        bitonic_merge<128, 64>(a0, a1, lane);
        if (lane < total) dst[lane] = a0;
        if (lane + 64 < total) dst[lane + 64] = a1;
        wave_sync();
        return total;
    }
    int total = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool mine = i < n && ok(i);
        const double v = mine ? val(i) : 0.0;
        int rank = 0;
        for (int bj = 0; bj < n; bj += 64) {
            const int jj = bj + lane;
            const bool okj = jj < n && ok(jj);
            const double ul = okj ? val(jj) : 0.0;
            const unsigned long long okmask = __ballot(okj);
            const int nt = okmask ? 64 - __builtin_clzll(okmask) : 0;   // uniform: nothing valid beyond
            for (int t = 0; t < nt; ++t) {
                const double u = readlane_f64(ul, t);
                const int jx = bj + t;
                rank += (((okmask >> t) & 1ull) && (u < v || (u == v && jx < i))) ? 1 : 0;
            }
        }
        if (mine) dst[rank] = v;
        total += __builtin_popcountll(__ballot(mine));
    }
    wave_sync();
    return total;
}

// TS33C/stream_decoder.py:112-202
__device__ inline void wave_decode_ts33c(const WaveLds& L, int n, double reset_limit, DecState* st,
                                         const BackParams& P, sdr_frame* f /* lane 0 writes */)
{
    const int lane = threadIdx.x & 63;
    const double* cur = L.cur;
    const int np_ = (n + 1) / 2, ng = n / 2;
    auto pulse = [&](int j) { return cur[2 * j]; };
    auto all_ok = [&](int) { return true; };
#ifdef SDR_BK_TIMING
    long long q0 = __builtin_readcyclecounter();
#endif
    wave_rank_sort(pulse, all_ok, np_, L.scratch);
#ifdef SDR_BK_TIMING
    long long q1 = __builtin_readcyclecounter();
#endif
    const double plim = hist_auto_mid_edge(L.scratch, np_, P.pow_m13, P.log2n);
    wave_sync();
#ifdef SDR_BK_TIMING
    long long q2 = __builtin_readcyclecounter();
#endif
    auto gap = [&](int j) { return cur[2 * j + 1]; };
    auto gap_ok = [&](int j) {
        double g = cur[2 * j + 1];
        return g > 0.0 && !(g > reset_limit);
    };
    const int ngf = wave_rank_sort(gap, gap_ok, ng, L.scratch);
    const double glim = hist_auto_mid_edge(L.scratch, ngf, P.pow_m13, P.log2n);
    wave_sync();
    st->pulse_short_limit = plim;
    st->gap_short_limit = glim;

    L.words[lane] = 0;
    wave_sync();
    int base_r = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        bool valid = false;
        int bit = 0;
        if (k < n) {
            if ((k & 1) == 0) {
                bit = (cur[k] < plim) ? 1 : 0;
                valid = true;
            } else {
                double g = cur[k];
                if (g > reset_limit) g = 0.0;
                valid = !(g < glim);  // short gaps (255) are removed
            }
        }
        const unsigned long long m = __ballot(valid);
        const int r = base_r + __builtin_popcountll(m & lanemask_lt());
        if (valid) {
            if ((r + 1) % 9 == 0) {
                const int pi = (r + 1) / 9 - 1;
                if (pi < 256) L.par[pi] = (unsigned char)bit;
            } else {
                const int d = r - r / 9;
                if (!bit && d < 2048) atomicOr(&L.words[d >> 5], 1u << (d & 31));  // reverse + invert
            }
        }
        base_r += __builtin_popcountll(m);
    }
    wave_sync();
    const int nraw = base_r, npar = nraw / 9, ndata = nraw - npar;
    int status;
    int nbytes = 0;
    if ((ndata % 8) != 0)
        status = SDR_FRAME_BAD_LENGTH;
    else {
        nbytes = ndata / 8;
        bool par_ok = true;
        if (npar == nbytes) {
            for (int i0 = 0; i0 < nbytes; i0 += 64) {
                const int i = i0 + lane;
                bool bad = false;
                if (i < nbytes) {
                    unsigned int b = (L.words[(i >> 2) & 63] >> (8 * (i & 3))) & 0xffu;
                    bool even = (__builtin_popcount(b) & 1) == 0;
                    bad = even != (L.par[i & 255] == 1);
                }
                if (__ballot(bad)) par_ok = false;
            }
        }
        const unsigned int b0 = L.words[0] & 0xffu;
        if (npar != nbytes || !par_ok)
            status = SDR_FRAME_PARITY;
        else if (!(nbytes == 10 && b0 == 0x9f))
            status = SDR_FRAME_UNKNOWN_HEADER;
        else
            status = SDR_FRAME_OK;
    }
#ifdef SDR_BK_TIMING
    long long q3 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) printf("  ts33c n=%d: sort %lld hist %lld rest(sort2+hist2+bits) %lld\n", n, q1 - q0, q2 - q1, q3 - q2);
#endif
    if (lane == 0 && f) {
        f->lim[0] = plim;
        f->lim[1] = glim;
        f->nbits = nraw;
        f->status = status;
        if (status != SDR_FRAME_BAD_LENGTH) {
            f->nbytes = nbytes < 16 ? nbytes : 16;
            unsigned char bytes[16];
            for (int i = 0; i < 16; ++i) bytes[i] = (unsigned char)((L.words[i >> 2] >> (8 * (i & 3))) & 0xffu);
            for (int i = 0; i < f->nbytes; ++i) f->bytes[i] = bytes[i];
            if (status == SDR_FRAME_OK) {
                int channel = (bytes[1] >> 5) & 0x0F;
                if (channel >= 5) channel -= 1;
                int temp = (bytes[5] & 0x0F) * 100 + ((bytes[4] & 0xF0) >> 4) * 10 + (bytes[4] & 0x0F);
                int sign = (((bytes[5] >> 7) & 0x01) * 2) - 1;
                f->v[0] = channel;
                f->v[1] = bytes[1] & 0x0F;
                f->v[2] = ((bytes[5] >> 6) & 0x01) == 1;
                f->v[3] = ((bytes[6] & 0xF0) >> 4) * 10 + (bytes[6] & 0x0F);
                f->temp_tenths = temp * sign;
            }
        }
    }
    wave_sync();
}

// TFA/stream_decoder.py:132-199 (+ NaN guard of analyzer2.grc:897,959-960)
__device__ inline void wave_decode_tfa(const WaveLds& L, int n0, bool nan_guard, sdr_frame* f)
{
    const int lane = threadIdx.x & 63;
    const double* c = L.cur + (n0 < 6 ? n0 : 6);
    const int n = n0 < 6 ? 0 : n0 - 6;
    const int ns = n < 150 ? n : 150;
    int status = -1;
    double sym = 0.0;
    long long nbits = 0;
    if (ns & 1)
        status = SDR_FRAME_EXC_RESHAPE;
    else if (ns == 0) {
        sym = __builtin_nan("");   // np.mean of an empty sync array
        status = nan_guard ? SDR_FRAME_SYMBOL_LENGTH : SDR_FRAME_EXC_NAN;
    }
    int ds = -1;
    if (status < 0) {
        const int npair = ns / 2;
        for (int i = lane; i < npair; i += 64) L.scratch[i] = c[2 * i] + c[2 * i + 1];
        wave_sync();
        sym = np_pairwise_sum_small(L.scratch, npair) / (double)npair;
        const double thr = sym * 1.5;
        for (int k0 = 0; k0 < n && ds < 0; k0 += 64) {
            const int k = k0 + lane;
            const unsigned long long m = __ballot(k < n && c[k] > thr);
            if (m) ds = k0 + __builtin_ctzll(m);
        }
        if (ds < 0) status = SDR_FRAME_EXC_INDEX;
    }
    if (status < 0) {
        if (lane < 3) L.words[lane] = 0;
        wave_sync();
        for (int k0 = ds; k0 < n - 1; k0 += 64) {
            const int k = k0 + lane;
            long long cnt = 0;
            bool one = false;
            if (k < n - 1) {
                double w = c[k];
                one = ((k - ds) & 1) == 0;
                if (one) {
                    if (w < sym) w = 0.0;
                    cnt = (long long)floor(w / sym);
                } else
                    cnt = (long long)ceil(w / sym);
            }
            long long incl = cnt;  // inclusive prefix sum over the wave
            for (int off = 1; off < 64; off <<= 1) {
                long long t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            const long long start = nbits + incl - cnt;
            if (one)
                for (long long b = start; b < start + cnt && b < 88; ++b)
                    atomicOr(&L.words[b >> 5], 1u << (b & 31));  // per-byte LSB first == bit b of the stream
            nbits += __shfl(incl, 63);
        }
        wave_sync();
        if (nbits < 88)
            status = SDR_FRAME_BAD_LENGTH;
        else {
            unsigned char bytes[11];
            for (int i = 0; i < 11; ++i) bytes[i] = (unsigned char)((L.words[i >> 2] >> (8 * (i & 3))) & 0xffu);
            status = (crc8_31(bytes + 2, 8) == bytes[10]) ? SDR_FRAME_OK : SDR_FRAME_CRC;
            if (lane == 0 && f) {
                f->nbytes = 11;
                for (int i = 0; i < 11; ++i) f->bytes[i] = bytes[i];
                if (status == SDR_FRAME_OK) {
                    f->v[0] = ((int)bytes[2] << 8) + bytes[3];
                    f->v[1] = bytes[6] & 0x7f;
                    f->v[2] = ((bytes[7] >> 7) & 0x01) == 0;
                    f->temp_tenths = (bytes[4] & 0x0f) * 100 + ((bytes[5] >> 4) & 0x0f) * 10 + (bytes[5] & 0x0f) - 400;
                }
            }
        }
    }
    if (lane == 0 && f) {
        f->lim[0] = sym;
        f->nbits = (int)(nbits > 0x7fffffff ? 0x7fffffff : nbits);
        f->status = status;
    }
    wave_sync();
}

// TS33C/stream_decoder.py:39-110,204-217 on one wavefront
template <class WF>
__device__ inline void wave_stream_decode(const BackParams& P, int cap, WF W, const int32_t* pos, long long R,
                                          int first_level, int last_level, bool any_one, const WaveLds& L,
                                          DecState* st)
{
    const int lane = threadIdx.x & 63;
    const bool ts33c = (dec_of(P) == SDR_DEC_TS33C);
    const double reset_limit = ts33c ? 1550.0 : 500.0;
    if (!(R > 0 && (any_one || st->n_cur > 0))) {
        st->state = 0;
        return;
    }
    const int cur_type = first_level;
    st->chunk_id += 1;
    // One pass over the run widths (four blocks of 64 per step, so that their loads overlap): the number
    // of over-limit widths and -- long rows (P.pieces_hdr) -- the run index behind each of them, left in
    // pieces_b.  Every piece that both starts behind and ends with an over-limit width is a packet of its
    // own (the piece before it has left no pending symbols), so on long rows only the first and the last
    // piece are walked here and the pieces in between go to k_decode_pieces, one wavefront each; a middle
    // piece longer than the pending-symbol table keeps the row on the sequential walk (its overflow
    // semantics).
    long long n_over = 0, last_b = 0, nf_mid = 0;
    bool too_long = false;
    int32_t* pb = P.pieces_hdr ? P.pieces_b + (long long)cap * P.bits_stride : nullptr;
    for (long long k00 = 0; k00 < R; k00 += 256) {
        bool ov[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long k = k00 + 64 * u + lane;
            ov[u] = k < R && W(k) > reset_limit;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long k0 = k00 + 64 * u, k = k0 + lane;
            const bool over = ov[u];
            const unsigned long long m = __ballot(over);
            if (!m) continue;
            if (pb) {
                const unsigned long long below = m & lanemask_lt();
                const long long prev = below ? k0 + (63 - __builtin_clzll(below)) + 1 : last_b;
                const long long idx = n_over + __builtin_popcountll(below);   // piece number this width closes
                const long long len = k + 1 - prev;
                if (over) pb[idx] = (int32_t)(k + 1);
                too_long = too_long || __ballot(over && idx >= 1 && len > SDR_CUR_CAP) != 0;
                nf_mid += __builtin_popcountll(__ballot(over && idx >= 1 && len >= 2));
                last_b = k0 + (63 - __builtin_clzll(m)) + 1;
            }
            n_over += __builtin_popcountll(m);
        }
    }
    const long long npieces = n_over + 1;
    const bool size_is_1 = (npieces == 1 && R == 1);
    if (npieces > 1) st->state = 1;
    double* cur = L.cur;
    bool delegate = pb && npieces >= 4 && !too_long;
    long long a = 0;
    for (long long pc = 0; pc < npieces; ++pc) {
        if (delegate && pc == 1) {
            if (st->n_cur != 0 || (st->overflow & 4))
                delegate = false;            // the first piece did not end clean: walk the row as before
            else {
                int32_t* hdr = P.pieces_hdr + (long long)cap * 8;
                const long long f1 = st->n_frames;
                if (lane == 0) {
                    hdr[1] = (int32_t)n_over;
                    hdr[2] = (int32_t)f1;
                    hdr[3] = (int32_t)nf_mid;
                    hdr[0] = 1;
                }
                if (f1 + nf_mid > P.max_frames) {
                    st->overflow |= 2;
                    st->n_frames = P.max_frames;
                } else
                    st->n_frames = (int)(f1 + nf_mid);
                a = last_b;
                pc = npieces - 1;
            }
        }
        long long b;
        if (pc == npieces - 1)
            b = R;
        else {
            b = -1;
            for (long long k0 = a; b < 0; k0 += 64) {
                const long long k = k0 + lane;
                const unsigned long long m = __ballot(k < R && W(k) > reset_limit);
                if (m) b = k0 + __builtin_ctzll(m) + 1;  // the over-limit width closes the piece
            }
        }
        long long s = a;
        if (size_is_1 || st->state == 0) ++s;
        if (s < b) {
            st->state = 1;
            if (st->n_cur > 0 && st->prev_type == cur_type) {
                if (lane == 0) cur[st->n_cur - 1] += W(s);
                ++s;
            }
            const long long room = SDR_CUR_CAP - st->n_cur;
            long long cnt = b - s;
            if (cnt > room) {
                cnt = room;
                st->overflow |= 4;
            }
            for (long long i = lane; i < cnt; i += 64) cur[st->n_cur + i] = W(s + i);
            st->n_cur += (int)cnt;
            wave_sync();
            if (cur[0] > reset_limit) {
                for (int i0 = 0; i0 < st->n_cur - 1; i0 += 64) {
                    const int i = i0 + lane;
                    double v = (i < st->n_cur - 1) ? cur[i + 1] : 0.0;
                    wave_sync();
                    if (i < st->n_cur - 1) cur[i] = v;
                    wave_sync();
                }
                st->n_cur--;
            }
            if (st->n_cur > 1 && cur[st->n_cur - 1] > reset_limit) {
                sdr_frame* f = nullptr;
                if (st->n_frames >= P.max_frames)
                    st->overflow |= 2;
                else {
                    f = P.frames + (long long)cap * P.max_frames + st->n_frames;
                    st->n_frames++;
                    if (lane == 0) frame_init(f, cap, dec_of(P), pos[b]);
                }
                wave_sync();
                if (f) {
                    if (ts33c)
                        wave_decode_ts33c(L, st->n_cur, reset_limit, st, P, f);
                    else
                        wave_decode_tfa(L, st->n_cur, dec_of(P) == SDR_DEC_TFA_GRC, f);
                }
                st->n_cur = 0;
            }
        } else {
            st->state = 0;
        }
        a = b;
    }
    st->prev_type = last_level;
}

// The closed middle pieces of a long row (see wave_stream_decode): piece i = runs [b[i-1], b[i]),
// one wavefront per piece, frame slot = frames of the first piece + number of earlier middle pieces
// that hold a packet (two symbols or more).  TS33C: the decoder keeps the limits of the LAST decoded
// packet as attributes; the piece that owns the last frame slot writes them.
__global__ void __launch_bounds__(64) k_decode_pieces(BackParams P)
{
    __shared__ __align__(16) double s_cur[SDR_CUR_CAP];
    __shared__ __align__(16) double s_scr[SDR_CUR_CAP];
    __shared__ unsigned int s_words[64];
    __shared__ unsigned char s_par[256];
    const int cap = blockIdx.y, lane = threadIdx.x & 63;
    const int32_t* hdr = P.pieces_hdr + (long long)cap * 8;
    if (hdr[0] != 1) return;
    const long long n_over = hdr[1], f1 = hdr[2], nf_mid = hdr[3];
    const int32_t* pb = P.pieces_b + (long long)cap * P.bits_stride;
    const int32_t* pos = P.runs + (long long)cap * (P.runs_cap + 2);
    const WidthsFromPos W{pos, P.dec_rate};
    const bool ts33c = dec_of(P) == SDR_DEC_TS33C;
    const long long total = f1 + nf_mid > P.max_frames ? P.max_frames : f1 + nf_mid;
    DecState* st = P.st + cap;
    const WaveLds L{s_cur, s_scr, s_words, s_par};
    for (long long i = 1 + blockIdx.x; i <= n_over - 1; i += gridDim.x) {
        const long long a = pb[i - 1], b = pb[i];
        const long long len = b - a;
        if (len < 2) continue;
        long long cnt = 0;
        for (long long j0 = 1; j0 < i; j0 += 64) {
            const long long j = j0 + lane;
            cnt += __builtin_popcountll(__ballot(j < i && pb[j] - pb[j - 1] >= 2));
        }
        const long long slot = f1 + cnt;
        if (slot >= P.max_frames) continue;
        for (long long t = lane; t < len; t += 64) s_cur[t] = W(a + t);
        sdr_frame* f = P.frames + (long long)cap * P.max_frames + slot;
        if (lane == 0) frame_init(f, cap, dec_of(P), pos[b]);
        wave_sync();
        if (ts33c) {
            DecState tmp;
            tmp.pulse_short_limit = 0.0;
            tmp.gap_short_limit = 0.0;
            wave_decode_ts33c(L, (int)len, 1550.0, &tmp, P, f);
            if (slot == total - 1 && lane == 0) {
                st->pulse_short_limit = tmp.pulse_short_limit;
                st->gap_short_limit = tmp.gap_short_limit;
            }
        } else
            wave_decode_tfa(L, (int)len, dec_of(P) == SDR_DEC_TFA_GRC, f);
        wave_sync();
    }
}
