// k_backend: one 256-thread workgroup per capture (or per streaming call).
// LDS plan (26 KiB -> 6 workgroups per CU, so a 1250-capture batch is one
// round on 256 CUs):
//   A  8 KiB  currentSymbols (pending + this call's widths of the open packet)
//   B  8 KiB  phase 1: the sliced bytes | phase 2: sort scratch
//   C  8 KiB  phase 0: per-thread carry table | phase 1+: run widths (or ticks)
#pragma once
#include "backend_kernels.h"
#include "backend_wave.h"

#define BK_BYTES_LDS 8192

// Run starts of bits[b0, b1) (b0 a multiple of the 16-KiB round, b1 <= nb): entry number w (counted from 1
// inside this range) goes to pos[w] (and to s_pos[w] while w < 2048, if s_pos is given).  Every thread
// owns 64 bytes per round, held in registers as 16 dwords straight from global memory; edges come from
// word arithmetic (d = w ^ (w << 8 | previous byte); a byte of d is non-zero where the level changes).
// Returns the number of edges; called by all BK_THREADS threads.
__device__ __forceinline__ long long rl_blocks(const int8_t* __restrict__ bits, const long long nb, const long long b0,
                                               const long long b1, int32_t* pos, const long long cap_w, int32_t* s_pos,
                                               int* s_wtot)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
        long long rbase = 0;   // edges found in the blocks before this one
        // one block = RL_W words (64 bytes) per thread + the byte in front of them; rows are padded
        // to 32 bytes, so every 16-byte piece that starts inside the row can be read whole
        constexpr int RL_W = 16, RL_BYTES = 4 * RL_W, RL_BLK = BK_THREADS * RL_BYTES;
        auto fetch_blk = [&](long long blk, unsigned (&o)[RL_W], unsigned& pbyte) {
            const long long j0 = blk + tid * RL_BYTES;
#pragma unroll
            for (int q = 0; q < RL_W / 4; ++q) {
                // unconditional (pieces past the row's end re-read its first piece; their words are masked by
                // `valid` below): a load under `if` is an exec-masked block of its own that the compiler puts a
                // wait for the previous load in front of -- four round trips per block instead of one
                const long long off = (j0 + 16 * q < nb) ? j0 + 16 * q : 0;
                const uint4 a = *reinterpret_cast<const uint4*>(bits + off);
                o[4 * q] = a.x; o[4 * q + 1] = a.y; o[4 * q + 2] = a.z; o[4 * q + 3] = a.w;
            }
            const unsigned pb = (unsigned)(unsigned char)bits[(j0 > 0 && j0 < nb) ? j0 - 1 : 0];
            pbyte = (j0 < nb) ? (j0 > 0 ? pb : (o[0] & 0xffu)) : 0u;
        };
        unsigned wd[RL_W], prevb;
        int par = 0;
        fetch_blk(b0, wd, prevb);
        for (long long blk = b0; blk < b1; blk += RL_BLK) {   // long rows (channelizer, D=1 chains): 16 KiB at a time
            const long long j0 = blk + tid * RL_BYTES;
            long long vl = nb - j0;
            const int valid = vl < 0 ? 0 : (vl > RL_BYTES ? RL_BYTES : (int)vl);
            unsigned nz[RL_W];
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < RL_W; ++k) {
                const unsigned d = wd[k] ^ ((wd[k] << 8) | prevb);
                unsigned m = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;   // bit 7 of every non-zero byte
                const int left = valid - 4 * k;                                      // bytes of this word inside the row
                if (left <= 0)
                    m = 0;
                else if (left < 4)
                    m &= (1u << (8 * left)) - 1u;
                nz[k] = m;
                cnt += __builtin_popcount(m);
                prevb = wd[k] >> 24;
            }
            // the masks are all this block still needs: the next block's loads go out now and are
            // in flight during the scan and the edge stores (issued any earlier, the wait for this
            // block's words would also wait for them)
            if (blk + RL_BLK < b1) fetch_blk(blk + RL_BLK, wd, prevb);
            const int incl = wave_incl_scan_dpp(cnt);
            int* const wt = s_wtot + (par << 2);
            par ^= 1;
            if (lane == 63) wt[wv] = incl;
            lds_only_barrier();   // the only barrier of a round (the next round writes the other set of totals); it must
                                  // not wait for the next block's loads, which are in flight on purpose
            int wbase = 0, total = 0;
            for (int w = 0; w < 4; ++w) {
                if (w < wv) wbase += wt[w];
                total += wt[w];
            }
            long long w = 1 + rbase + wbase + incl - cnt;
            if (cnt) {
#pragma unroll
                for (int k = 0; k < RL_W; ++k) {
                    unsigned m = nz[k];
                    while (m) {
                        const int by = __builtin_ctz(m) >> 3;
                        const int32_t at = (int32_t)(j0 + 4 * k + by);
                        if (w < cap_w) {
                            pos[w] = at;
                            if (s_pos && w < 2048) s_pos[w] = at;
                        }
                        ++w;
                        m &= m - 1;
                    }
                }
            }
            rbase += total;
        }
        return rbase;
}

// Long rows: one workgroup per seg_bytes of a row finds that segment's run starts (absolute byte
// positions) and leaves them, compacted, in the row's runs_seg slice from index seg*seg_bytes on
// (a segment cannot hold more edges than bytes) with the count in seg_cnt; k_backend (phase_sel 2)
// scans the counts and gathers.  Runs after the carries are patched (k_backend phase_sel 1).
__global__ void __launch_bounds__(BK_THREADS) k_runs_seg(BackParams P)
{
    __shared__ int s_wtot[8];
    const int seg = blockIdx.x, cap = blockIdx.y;
    const int8_t* bits = P.bits + (long long)cap * P.bits_stride;
    const long long b0 = (long long)seg * P.seg_bytes;
    const long long b1 = b0 + P.seg_bytes < P.nb ? b0 + P.seg_bytes : P.nb;
    long long n = 0;
    if (b0 < b1) n = rl_blocks(bits, P.nb, b0, b1, P.runs_seg + (long long)cap * P.bits_stride + b0 - 1, P.seg_bytes + 1, nullptr, s_wtot);
    if (threadIdx.x == 0) P.seg_cnt[(long long)cap * P.nseg + seg] = (int32_t)n;
}

__global__ void __launch_bounds__(BK_THREADS, 6) k_backend(BackParams P)
{
    __shared__ __align__(16) unsigned char s_A[SDR_CUR_CAP * 8];
    __shared__ __align__(16) unsigned char s_B[SDR_CUR_CAP * 8];
    __shared__ __align__(16) unsigned char s_C[SDR_CUR_CAP * 8];
    __shared__ int s_cnt[BK_THREADS];
    __shared__ int s_wtot[8];   // two sets of per-wave totals, alternating by round
    __shared__ int s_fe[4];
    __shared__ unsigned int s_words[64];
    __shared__ unsigned char s_par[256];
    __shared__ unsigned char s_bits[512];

    double* s_cur = reinterpret_cast<double*>(s_A);
    double* s_scratch = reinterpret_cast<double*>(s_B);
    double* s_w = reinterpret_cast<double*>(s_C);
    long long* s_tick = reinterpret_cast<long long*>(s_C);
    int* s_has = reinterpret_cast<int*>(s_C);
    int* s_last = s_has + BK_THREADS;

    const int vcap = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int cap = vcap;                  // row of the sliced bytes / run table
    if (P.ndec) cap = vcap % P.vrows;   // several decoders over the same rows in one launch: dec_of(P) picks this workgroup's
#ifdef SDR_BK_TIMING
    long long tk0 = __builtin_readcyclecounter(), tk1 = 0, tk2 = 0, tk3 = 0;
#endif
    int8_t* bits = P.bits + (long long)cap * P.bits_stride;
    DecState* st = P.st + vcap;
    // behind a gating squelch every capture of a batch has its own length
    long long nout = P.nout, nb = P.nb;
    int run_decoder = P.run_decoder;
    if (P.nout_var) {
        nout = P.nout_var[cap];
        nb = P.patch_off + (nout > P.keep_c0 ? (nout - P.keep_c0 + P.keep_n - 1) / P.keep_n : 0);
        if (dec_of(P) >= SDR_DEC_TS33C && dec_of(P) <= SDR_DEC_TFA_GRC)
            run_decoder = ((double)nb * 1e6 / P.dec_rate > (dec_of(P) == SDR_DEC_TS33C ? 1550.0 : 500.0)) ? 1 : 0;
    }

    if (tid == 0) {
        if (P.fresh) {
            st->state = 0;
            st->n_cur = 0;
            st->prev_type = 0;
            st->chunk_id = 0;
            st->thr_state = P.thr_init;
            st->overflow = 0;
            st->pulse_short_limit = 775.0;
            st->gap_short_limit = 775.0;
        }
        st->n_frames = 0;
        if (P.phase_sel != 2) st->first_evt = -1;   // phase_sel 2 follows a phase-0 launch that has set it
    }
    // the carry launch of a long row clears the row's piece header for the launches behind it
    if (P.phase_sel == 1 && P.pieces_hdr && tid < 8) P.pieces_hdr[(long long)cap * 8 + tid] = 0;
    __syncthreads();

    // ---- phase 0: hysteresis carries --------------------------------------
    if (P.tsum) {
        const TileSum* ts = P.tsum + (long long)cap * P.tiles_per_capture;
        const int nt = P.tiles_per_capture;
        const int per = (nt + BK_THREADS - 1) / BK_THREADS;
        const int t0 = tid * per, t1 = (t0 + per < nt) ? t0 + per : nt;
        int has = 0, last = 0, myfe = -1;
        const long long* ts64 = reinterpret_cast<const long long*>(ts);   // (first_evt, last_state) as one 8-B load
#pragma unroll 8
        for (int i = t0; i < t1; ++i) {
            const long long v = ts64[i];
            const int fe = (int)v;
            if (fe >= 0) {
                if (!has) myfe = fe;
                has = 1;
                last = (int)(v >> 32);
            }
        }
        // nearest earlier thread that saw an event: ballot inside the wave, a
        // 4-entry table across waves
        const unsigned long long mh = __ballot(has), ml = __ballot(last);
        if (lane == 0) {
            s_has[wv] = mh != 0;
            s_last[wv] = mh ? (int)((ml >> (63 - __builtin_clzll(mh))) & 1) : 0;
            if (!mh) s_fe[wv] = -1;
        }
        // tiles are handed out in ascending order: the call's first event is the
        // first event of the first thread that saw one
        if (mh && lane == __builtin_ctzll(mh)) s_fe[wv] = myfe;
        __syncthreads();
        int carry = P.use_state_thr ? st->thr_state : P.thr_init;
        const unsigned long long below = mh & ((1ull << lane) - 1ull);
        if (below)
            carry = (int)((ml >> (63 - __builtin_clzll(below))) & 1);
        else
            for (int w = wv - 1; w >= 0; --w)
                if (s_has[w]) {
                    carry = s_last[w];
                    break;
                }
        int8_t* wbits = bits + P.patch_off;
#pragma unroll 4
        for (int i = t0; i < t1; ++i) {
            const long long v = ts64[i];
            long long o_beg = (long long)i * P.tile_out;
            long long o_end = o_beg + P.tile_out < nout ? o_beg + P.tile_out : nout;
            long long fe = (int)v;
            if (carry == 1) {
                long long e = fe >= 0 ? fe : o_end;
                long long o = o_beg < P.keep_c0 ? P.keep_c0 : o_beg;
                if (P.keep_n > 1) {
                    long long r = (o - P.keep_c0) % P.keep_n;
                    if (r) o += P.keep_n - r;
                }
                for (; o < e; o += P.keep_n) wbits[(o - P.keep_c0) / P.keep_n] = 1;
            }
            if (fe >= 0) carry = (int)(v >> 32);
        }
        __syncthreads();
        // the owner of the last tile hands the carry-out to thread 0 through LDS;
        // thread 0 is the only writer of *st
        if ((t0 < nt && t1 == nt) || (nt == 0 && tid == 0)) s_cnt[0] = carry;
        __syncthreads();
        if (tid == 0) {
            st->thr_state = s_cnt[0];
            int fe = -1;
            for (int w = 3; w >= 0; --w)
                if (s_fe[w] >= 0) fe = s_fe[w];
            st->first_evt = fe;
        }
    }
    if (!run_decoder || dec_of(P) == SDR_DEC_NONE || P.phase_sel == 1) return;
    __syncthreads();
#ifdef SDR_BK_TIMING
    tk1 = __builtin_readcyclecounter();
#endif

    // ---- phase 1: run-length ------------------------------------------------
    int32_t* pos = P.runs + (long long)cap * (P.runs_cap + 2);
    // Aligned rows, 8 KiB per round: every thread owns 32 bytes, held in
    // registers as 8 dwords straight from global memory; edges come from word arithmetic
    // (d = w ^ (w << 8 | previous byte); a byte of d is non-zero where the level changes)
    // instead of a byte-at-a-time walk through LDS, and the run starts are kept in LDS too
    // so that the widths do not wait on a global write/read round trip.
    const bool fast = ((reinterpret_cast<uintptr_t>(bits) & 15) == 0) && (P.bits_stride & 31) == 0;
    int32_t* s_pos = reinterpret_cast<int32_t*>(s_B);   // [2048], free until phase 2 needs its sort scratch
    long long R = 0;
    bool runs_overflow = false;
    bool w_in_lds = false, pos_in_lds = false;
    int first_level = 0, last_level = 0;
    if (P.reuse_runs) {
        // a further decoder over rows whose run table the previous launch has left in place
        // (sdr_redecode): only the LDS mirror of the run starts is refilled
        const unsigned rn = (unsigned)P.runs_n[cap];
        R = nb > 0 ? (long long)(rn & 0x7fffffffu) : 0;
        runs_overflow = (rn >> 31) != 0;
        w_in_lds = R <= SDR_CUR_CAP;
        pos_in_lds = R + 1 <= 2048;
        if (pos_in_lds)
            for (long long k = tid; k <= R; k += BK_THREADS) s_pos[k] = pos[k];
        if (P.lvl_override & 1) {   // the table was handed in without the bytes it describes
            first_level = (P.lvl_override >> 1) & 1;
            last_level = (P.lvl_override >> 2) & 1;
        } else {
            first_level = nb > 0 ? (bits[0] != 0) : 0;
            last_level = nb > 0 ? (bits[nb - 1] != 0) : 0;
        }
        __syncthreads();
    } else if (fast) {
        if (tid == 0) {
            pos[0] = 0;
            s_pos[0] = 0;
        }
        long long rbase;
        if (P.phase_sel == 2) {
            // the segments' run starts are already in runs_seg (k_runs_seg): scan the counts, one per
            // thread, and move every segment's list to its place in the row's run table
            const int32_t* sc = P.seg_cnt + (long long)cap * P.nseg;
            const int c = tid < P.nseg ? sc[tid] : 0;
            const int incl = wave_incl_scan_dpp(c);
            if (lane == 63) s_wtot[wv] = incl;
            __syncthreads();
            int wbase = 0, total = 0;
            for (int w = 0; w < 4; ++w) {
                if (w < wv) wbase += s_wtot[w];
                total += s_wtot[w];
            }
            s_cnt[tid] = wbase + incl - c;   // exclusive offset of segment tid
            __syncthreads();
            const int32_t* src = P.runs_seg + (long long)cap * P.bits_stride;
            for (int sg = wv; sg < P.nseg; sg += BK_THREADS / 64) {
                const int n_s = sc[sg];
                const long long o_s = 1 + s_cnt[sg];
                const int32_t* from = src + (long long)sg * P.seg_bytes;
                for (int i0 = 0; i0 < n_s; i0 += 256) {   // four loads in flight per lane
                    int32_t at[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + 64 * u + lane;
                        at[u] = from[i < n_s ? i : 0];   // unconditional: the four loads stay back to back
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + 64 * u + lane;
                        const long long w = o_s + i;
                        if (i < n_s && w < P.runs_cap) {
                            pos[w] = at[u];
                            if (w < 2048) s_pos[w] = at[u];
                        }
                    }
                }
            }
            rbase = total;
        } else
            rbase = rl_blocks(bits, nb, 0, nb, pos, P.runs_cap, s_pos, s_wtot);
        R = nb > 0 ? rbase + 1 : 0;
        if (R > P.runs_cap) {
            runs_overflow = true;
            R = P.runs_cap;
        }
        w_in_lds = R <= SDR_CUR_CAP;
        pos_in_lds = R + 1 <= 2048;
        first_level = nb > 0 ? (bits[0] != 0) : 0;
        last_level = nb > 0 ? (bits[nb - 1] != 0) : 0;
        if (tid == 0) {
            pos[R] = (int32_t)nb;
            if (R < 2048) s_pos[R] = (int32_t)nb;
        }
        __syncthreads();
    } else {
    const bool b_in_lds = false;
    const int8_t* bb = bits;
    const long long per = (nb + BK_THREADS - 1) / BK_THREADS;
    const long long j0 = (long long)tid * per, j1 = (j0 + per < nb) ? j0 + per : nb;
    int cnt = 0;
    {
        int prev = (j0 > 0 && j0 < nb) ? bb[j0 - 1] : (nb > 0 ? bb[0] : 0);
        for (long long j = j0; j < j1; ++j) {
            int b = bb[j];
            cnt += (b != prev);
            prev = b;
        }
    }
    // block exclusive scan of the per-thread edge counts
    const int incl = wave_incl_scan_dpp(cnt);
    if (lane == 63) s_wtot[wv] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wv) wbase += s_wtot[w];
        total += s_wtot[w];
    }
    R = (long long)total + 1;
    if (nb == 0) R = 0;
    if (R > P.runs_cap) {
        runs_overflow = true;
        R = P.runs_cap;
    }
    w_in_lds = R <= SDR_CUR_CAP;
    first_level = nb > 0 ? (bb[0] != 0) : 0;
    last_level = nb > 0 ? (bb[nb - 1] != 0) : 0;
    if (tid == 0) pos[0] = 0;
    {
        long long w = 1 + wbase + incl - cnt;
        int prev = (j0 > 0 && j0 < nb) ? bb[j0 - 1] : (nb > 0 ? bb[0] : 0);
        for (long long j = j0; j < j1; ++j) {
            int b = bb[j];
            if (b != prev) {
                if (w < R) pos[w] = (int32_t)j;
                ++w;
            }
            prev = b;
        }
    }
    if (tid == 0) pos[R] = (int32_t)nb;
    __syncthreads();  // pos[] (global) written by this block is read back below
    (void)b_in_lds;
    }
    if (tid == 0 && P.runs_n && !P.reuse_runs) P.runs_n[cap] = (int32_t)((unsigned)R | (runs_overflow ? 0x80000000u : 0u));
    if (nb == 0) return;

    // run widths (stream decoders) or edge ticks (callback rules) into LDS
    const bool stream_dec = dec_of(P) <= SDR_DEC_TFA_GRC;
    const int32_t* rpos = pos_in_lds ? s_pos : pos;
    // streaming rule decoders: ticks are absolute, the callback's state comes from / goes to P.rule
    RuleState* const rs = (!stream_dec && dec_of(P) != SDR_DEC_VELUX && P.rule) ? P.rule + vcap : nullptr;
    const long long rule_base = (rs && rs->valid) ? rs->base : 0;
    if (w_in_lds) {
        if (stream_dec)
            for (long long k = tid; k < R; k += BK_THREADS) s_w[k] = run_width(rpos, k, P.dec_rate);
        else
            for (long long k = tid; k < R; k += BK_THREADS)
                s_tick[k] = ((rule_base + (long long)rpos[k]) * 1000000LL) / (long long)P.dec_rate;
    }
    double* gcur = P.cur + (long long)vcap * SDR_CUR_CAP;
    if (stream_dec) {
        const int npend = st->n_cur;
        for (int k = tid; k < npend && k < SDR_CUR_CAP; k += BK_THREADS) s_cur[k] = gcur[k];
    }
    __syncthreads();
#ifdef SDR_BK_TIMING
    tk2 = __builtin_readcyclecounter();
#endif
    if (tid >= 64) return;

    // ---- phase 2: decode on wavefront 0 ----------------------------------------
    DecState ls = *st;
    if (runs_overflow) ls.overflow |= 1;
    const bool any_one = first_level || R > 1;
    if (stream_dec) {
        WaveLds L{s_cur, s_scratch, s_words, s_par};
        if (w_in_lds)
            wave_stream_decode(P, vcap, WidthsArray{s_w}, pos, R, first_level, last_level, any_one, L, &ls);
        else
            wave_stream_decode(P, vcap, WidthsFromPos{pos, P.dec_rate}, pos, R, first_level, last_level, any_one, L,
                               &ls);
        for (int k = tid; k < ls.n_cur; k += 64) gcur[k] = s_cur[k];
    } else {
        // all 64 lanes of wavefront 0 run the rule on identical (wave-uniform) values: control flow is
        // scalar branches instead of exec masking, every LDS read is a broadcast; the redundant stores
        // hit the same addresses with the same data
        EdgeCtx e{pos_in_lds ? s_pos : pos, w_in_lds ? s_tick : nullptr, R, nb, first_level, (long long)P.dec_rate, 1, rule_base, rs};
        if (dec_of(P) == SDR_DEC_SOMFY)
            rule_somfy(e, P, &ls, vcap, s_bits);
        else if (dec_of(P) == SDR_DEC_BERNER)
            rule_berner(e, P, &ls, vcap, s_bits);
        else if (dec_of(P) == SDR_DEC_MUMBI)
            rule_mumbi(e, P, &ls, vcap, s_bits);
        else if (dec_of(P) == SDR_DEC_KWMOBILE)
            rule_kwmobile(e, P, &ls, vcap, s_bits);
        else if (dec_of(P) == SDR_DEC_BALDR)
            rule_baldr(e, P, &ls, vcap, s_bits);
        else if (dec_of(P) == SDR_DEC_VELUX)
            rule_velux(e, P, &ls, vcap);
    }
    if (tid == 0) *st = ls;
#ifdef SDR_BK_TIMING
    tk3 = __builtin_readcyclecounter();
    if (tid == 0 && (cap == 0 || cap == 700)) printf("cap %d: phase0 %lld phase1 %lld phase2 %lld cycles (R=%lld)\n", cap, tk1 - tk0, tk2 - tk1, tk3 - tk2, R);
#endif
}
