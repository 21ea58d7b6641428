// Host-compiled test hooks for the __host__ __device__ decode helpers in
// backend_kernels.h.  They exist so that the CPU-only test suite can check the
// numpy restatements (histogram "auto" edge, pairwise mean, the stream-decoder
// state machine) against numpy itself without a GPU.  NOT a CPU fallback: no
// product entry point (sdr_process_* / sdr_batch_process) ever reaches them, and
// the Python mirror never calls them.
#pragma once
#include <algorithm>

struct SelftestDec {
    int decoder;
    double rate;
    DecState st;
    RuleState rs;
    std::vector<double> cur, cur_l, scratch, pw, lg;
    std::vector<int8_t> raw;
    std::vector<sdr_frame> frames;
};

extern "C" SDR_API int sdr_selftest_hist_mid_edge(const double* x, int n, double* out)
{
    if (!out || n < 0 || n >= SDR_MAX_POW_TABLE) return SDR_E_ARG;
    std::vector<double> s(x, x + n), pw, lg;
    host_pow_tables(pw, lg);
    insertion_sort(s.data(), n);
    *out = hist_auto_mid_edge(s.data(), n, pw.data(), lg.data());
    return SDR_OK;
}

extern "C" SDR_API int sdr_selftest_pairwise_mean(const double* x, int n, double* out)
{
    if (!out || n < 1 || n >= 128) return SDR_E_ARG;
    *out = np_pairwise_sum_small(x, n) / (double)n;
    return SDR_OK;
}

extern "C" SDR_API void* sdr_selftest_dec_create(int decoder, double rate)
{
    SelftestDec* d = new SelftestDec();
    d->decoder = decoder;
    d->rate = rate;
    memset(&d->st, 0, sizeof d->st);
    memset(&d->rs, 0, sizeof d->rs);
    d->st.pulse_short_limit = d->st.gap_short_limit = 775.0;
    d->cur.assign(SDR_CUR_CAP, 0.0);
    d->cur_l.assign(SDR_CUR_CAP, 0.0);
    d->scratch.assign(SDR_CUR_CAP, 0.0);
    host_pow_tables(d->pw, d->lg);
    return d;
}

extern "C" SDR_API void sdr_selftest_dec_destroy(void* p) { delete (SelftestDec*)p; }

// one work([chunk]) call of the decoder on host memory; frames of this call -> out
extern "C" SDR_API int sdr_selftest_dec_work(void* p, const int8_t* bits, size_t n, sdr_frame* out, size_t cap,
                                             size_t* nframes, int32_t state[4])
{
    SelftestDec* d = (SelftestDec*)p;
    if (!d || !nframes) return SDR_E_ARG;
    *nframes = 0;
    const bool sd = is_stream_decoder(d->decoder);
    if (!sd) d->raw.clear();
    d->raw.insert(d->raw.end(), bits, bits + n);
    const long long nb = (long long)d->raw.size();
    bool run = true;
    if (sd) run = ((double)nb * 1e6 / d->rate > decoder_reset_limit(d->decoder));
    if (run && nb > 0) {
        std::vector<int32_t> pos;
        pos.push_back(0);
        for (long long j = 1; j < nb; ++j)
            if (d->raw[j] != d->raw[j - 1]) pos.push_back((int32_t)j);
        const long long R = (long long)pos.size();
        pos.push_back((int32_t)nb);
        d->frames.assign(64, sdr_frame());
        BackParams P;
        memset(&P, 0, sizeof P);
        P.nb = nb;
        P.decoder = d->decoder;
        P.dec_rate = d->rate;
        P.frames = d->frames.data();
        P.max_frames = 64;
        P.pow_m13 = d->pw.data();
        P.log2n = d->lg.data();
        d->st.n_frames = 0;
        const int first_level = d->raw[0] != 0, last_level = d->raw[nb - 1] != 0;
        const bool any_one = first_level || R > 1;
        if (sd)
            stream_decode(P, 0, WidthsFromPos{pos.data(), d->rate}, pos.data(), R, first_level, last_level, any_one,
                          d->cur.data(), &d->st, d->scratch.data());
        else {
            std::vector<unsigned char> sb(512);
            // consecutive calls are one continuous edge stream for the callback rules (velux: one call = one capture)
            RuleState* rs = d->decoder == SDR_DEC_VELUX ? nullptr : &d->rs;
            EdgeCtx e{pos.data(), nullptr, R, nb, first_level, (long long)d->rate, 0, (rs && rs->valid) ? rs->base : 0, rs};
            if (d->decoder == SDR_DEC_SOMFY) rule_somfy(e, P, &d->st, 0, sb.data());
            if (d->decoder == SDR_DEC_BERNER) rule_berner(e, P, &d->st, 0, sb.data());
            if (d->decoder == SDR_DEC_MUMBI) rule_mumbi(e, P, &d->st, 0, sb.data());
            if (d->decoder == SDR_DEC_KWMOBILE) rule_kwmobile(e, P, &d->st, 0, sb.data());
            if (d->decoder == SDR_DEC_BALDR) rule_baldr(e, P, &d->st, 0, sb.data());
            if (d->decoder == SDR_DEC_VELUX) rule_velux(e, P, &d->st, 0);
        }
        *nframes = (size_t)d->st.n_frames;
        for (size_t i = 0; i < *nframes && i < cap && out; ++i) out[i] = d->frames[i];
    }
    if (sd && run) d->raw.clear();
    if (state) {
        state[0] = d->st.state;
        state[1] = d->st.n_cur;
        state[2] = d->st.prev_type;
        state[3] = d->st.chunk_id;
    }
    return d->st.overflow ? SDR_E_OVERFLOW : SDR_OK;
}

// SDR_POISON runs: guard bands found overwritten when their buffers were freed / bands checked so far
extern "C" SDR_API int sdr_selftest_guard_hits(int* checked)
{
    std::lock_guard<std::mutex> lk(g_guard_mu);
    if (checked) *checked = g_guard_checked;
    return g_guard_hits;
}
