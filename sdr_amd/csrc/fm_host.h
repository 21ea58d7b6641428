// Host side of the FM-Radio chain (sdr_fm_* in include/sdrhip.h).  The two
// complex stages reuse k_frontend through two internal chain handles:
//   A: input pre-rotation (xlating taps=[1]) + rational_resampler_ccc 1/D  -> complex
//   B: low_pass_filter (33 taps) + quadrature_demod                         -> float
// then k_fir_fff_decim, k_deemph, k_resample_fff.
#pragma once

struct sdr_fm {
    sdr_fm_cfg cfg;
    sdr_handle* A = nullptr;
    sdr_handle* B = nullptr;
    hipStream_t stream = nullptr;
    std::string err;
    float *d_ataps = nullptr, *d_rtaps = nullptr;  // audio FIR (reversed), resampler taps
    float* d_rtapsT = nullptr;                // resampler taps as the polyphase table k_deemph_resample reads (row = phase, oldest step first)
    bool fused_tail = true;                   // SDR_FM_TAIL=split: de-emphasis and resampler as two launches (tuning / test override)
    int n_ataps = 0, n_rtaps = 0, L = 1, M = 1;
    int audio_nw = 4;                         // waves per audio-FIR tile (SDR_FM_AUDIO_NW: tuning)
    bool audio_valu = false;                  // SDR_FM_AUDIO=valu: tuning/test override, read once at create
    double b0 = 0, p1 = 0;
    float *d_a2 = nullptr, *d_a3 = nullptr, *d_audio = nullptr;
    long long n1 = 0, n2 = 0, n4 = 0, cap2 = 0, cap4 = 0;
    // streaming state of the back half (the chain handles A and B carry their own)
    float* d_chist[2] = {nullptr, nullptr};   // audio FIR: the last n_ataps-1+2D discriminator floats
    float* d_ehist[2] = {nullptr, nullptr};   // audio resampler: the last taps-per-phase de-emphasised floats
    int chist_len = 0, ehist_len = 0, csel = 0, esel = 0, dsel = 0;
    unsigned long long c_in = 0, c_out = 0;   // discriminator floats consumed / audio-FIR outputs emitted
    unsigned long long e_in = 0, e_out = 0;   // de-emphasised floats consumed / audio samples emitted
    FmDeemphState* d_deemph = nullptr;        // two slots, read dsel / write dsel^1
    TailBatch tails{};                        // the call's history updates of the two channel-filter stages, flushed in one launch
    TailBatch btails{};                       // ... and of the back half (audio FIR, audio resampler)
    // sdr_fm_set_overlap: the back half (audio FIR, de-emphasis, audio resampler: ~2 M floats, latency-bound launches
    // of a few hundred workgroups) of chunk k runs on a second stream beside the channel filter of chunk k+1
    bool overlap = false;
    hipStream_t bstream = nullptr;
    hipEvent_t ev_front = nullptr, ev_back[2] = {nullptr, nullptr};
    bool back_pending = false;                // bstream holds work the main stream has not been ordered behind
    // the discriminator stage alternates between two output buffers, so that stage B of call k+1 does not wait for
    // the audio half of call k (still reading its floats) but only for that of call k-1
    float* d_tap_alt = nullptr;
    int bsel = 0;                             // buffer / event the current call uses
    bool back_rec[2] = {false, false};
};

// the main stream behind everything the back-half stream holds (fetches, reset, destroy, stream changes)
static int fm_join(sdr_fm* h)
{
    if (h && h->back_pending) {
        if (hipStreamWaitEvent(h->stream, h->ev_back[h->bsel], 0) != hipSuccess) return SDR_E_HIP;
        h->back_pending = false;
    }
    return SDR_OK;
}

// last H floats of (hist ++ x[0..n)) -> the other history buffer
static int fm_hist_update(sdr_fm* h, float** bufs, int& sel, int H, const float* x, long long n);
static int chain_restart(sdr_handle* c)
{
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 2; ++i) HIPCHK(c, memset_done(c->d_hist[i], 0, (size_t)c->hist_cap * sizeof(float2)));
    reset_stream_counters(c);
    return reset_dec_state(c);
}

static int fm_fail(sdr_fm* h, int code, const std::string& msg)
{
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

#define FMCHK(h, call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fm_fail(h, SDR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));    \
    } while (0)

extern "C" const char* sdr_fm_last_error(sdr_fm* h) { return h ? h->err.c_str() : g_err.c_str(); }

extern "C" int sdr_fm_destroy(sdr_fm* h)
{
    if (!h) return SDR_OK;
    if (h->bstream) {
        (void)hipStreamSynchronize(h->bstream);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->bstream);
        if (h->ev_front) (void)hipEventDestroy(h->ev_front);
        for (hipEvent_t e : h->ev_back)
            if (e) (void)hipEventDestroy(e);
    }
    if (h->A) sdr_destroy(h->A);
    if (h->B) sdr_destroy(h->B);
    void* ptrs[] = {h->d_rtapsT, h->d_ataps, h->d_rtaps, h->d_a2, h->d_a3, h->d_audio, h->d_chist[0], h->d_chist[1], h->d_ehist[0],
                    h->d_ehist[1], h->d_deemph, h->d_tap_alt};
    for (void* p : ptrs)
        if (p) (void)dev_free(p);
    delete h;
    return SDR_OK;
}

extern "C" int sdr_fm_create(const sdr_fm_cfg* cfg, sdr_fm** out)
{
    if (!cfg || !out) return fm_fail(nullptr, SDR_E_ARG, "sdr_fm_create: null argument");
    *out = nullptr;
    if (cfg->resamp_decim < 1 || cfg->audio_decim < 1 || cfg->audio_interp < 1 || cfg->audio_decimation < 1 ||
        cfg->max_samples < 1 || !(cfg->sample_rate > 0) || !(cfg->quad_rate > 0) || !(cfg->tau > 0))
        return fm_fail(nullptr, SDR_E_ARG, "sdr_fm_create: bad configuration");
    sdr_fm* h = new sdr_fm();
    h->cfg = *cfg;
    auto body = [&]() -> int {
        std::vector<float> buf(65536);
        // A: xlating(taps=[1]) + resampler 1/D (design_filter, Kaiser beta 7)
        sdr_chain_cfg a;
        memset(&a, 0, sizeof a);
        a.sample_rate = cfg->sample_rate;
        a.center_freq = cfg->center_freq;
        a.decim = cfg->resamp_decim;
        a.demod = SDR_DEMOD_NONE;
        a.keep_n = 1;
        a.tap_demod = 1;
        a.max_batch = 1;
        a.max_frames = 1;
        a.max_samples = cfg->max_samples;
        int rc = sdr_create(&a, &h->A);
        if (rc) return fm_fail(h, rc, std::string("stage A: ") + g_err);
        h->A->prerotate = true;
        int nt = sdr_resampler_taps(1, cfg->resamp_decim, 0.0, buf.data(), (int)buf.size());
        if (cfg->resamp_decim == 1) {
            buf[0] = 1.0f;
            nt = 1;
        }
        if (nt < 1) return fm_fail(h, SDR_E_ARG, "resampler tap design failed");
        rc = sdr_set_taps(h->A, buf.data(), nt, 0);
        if (rc) return fm_fail(h, rc, std::string("stage A taps: ") + h->A->err);
        {
            int T;
            size_t lds;
            bool ul;
            pick_tile(h->A, T, lds, ul);
            if (!ul) return fm_fail(h, SDR_E_ARG, "stage A tile does not fit LDS");
        }
        // B: low_pass_filter (Blackman) + quadrature_demod
        const long long n1max = (long long)(cfg->max_samples / cfg->resamp_decim) + 2;
        sdr_chain_cfg b;
        memset(&b, 0, sizeof b);
        b.sample_rate = cfg->sample_rate / cfg->resamp_decim;
        b.decim = 1;
        b.demod = SDR_DEMOD_QUAD;
        b.demod_gain = (float)(cfg->quad_rate / (2 * M_PI * 75e3));
        b.keep_n = 1;
        b.tap_demod = 1;
        b.max_batch = 1;
        b.max_frames = 1;
        b.max_samples = (uint64_t)n1max;
        rc = sdr_create(&b, &h->B);
        if (rc) return fm_fail(h, rc, std::string("stage B: ") + g_err);
        h->B->skip_slice = true;
        nt = sdr_firdes_low_pass(1.0, cfg->lpf_fs, cfg->lpf_fc, cfg->lpf_tw, SDR_WIN_BLACKMAN, 6.76, buf.data(),
                                 (int)buf.size());
        if (nt < 1) return fm_fail(h, SDR_E_ARG, "low_pass_filter design failed");
        rc = sdr_set_taps(h->B, buf.data(), nt, 0);
        if (rc) return fm_fail(h, rc, std::string("stage B taps: ") + h->B->err);
        // wfm_rcv audio filter: low_pass(1, quad_rate, audio_rate/2 - width, width, HAMMING)
        const double audio_rate = cfg->quad_rate / cfg->audio_decim;
        const double width = audio_rate / 32;
        nt = sdr_firdes_low_pass(1.0, cfg->quad_rate, audio_rate / 2 - width, width, SDR_WIN_HAMMING, 6.76,
                                 buf.data(), (int)buf.size());
        if (nt < 1) return fm_fail(h, SDR_E_ARG, "audio filter design failed");
        h->n_ataps = nt;
        if (const char* e = getenv("SDR_FM_AUDIO")) h->audio_valu = !strcmp(e, "valu");
        if (const char* e = getenv("SDR_FM_AUDIO_NW")) h->audio_nw = (atoi(e) == 1 || atoi(e) == 2) ? atoi(e) : 4;
        std::vector<float> rt(nt);
        for (int k = 0; k < nt; ++k) rt[nt - 1 - k] = buf[k];
        FMCHK(h, dev_alloc((void**)&h->d_ataps, nt * sizeof(float)));
        FMCHK(h, hipMemcpy(h->d_ataps, rt.data(), nt * sizeof(float), hipMemcpyHostToDevice));
        // fm_deemph(audio_rate, tau)
        {
            const double fs = audio_rate, w_c = 1.0 / cfg->tau;
            const double w_ca = 2.0 * fs * tan(w_c / (2.0 * fs));
            const double k = -w_ca / (2.0 * fs);
            h->p1 = (1.0 + k) / (1.0 - k);
            h->b0 = -k / (1.0 - k);
        }
        // rational_resampler_fff
        int g1 = cfg->audio_interp, g2 = cfg->audio_decimation;
        while (g2) {
            int t = g1 % g2;
            g1 = g2;
            g2 = t;
        }
        h->L = cfg->audio_interp / g1;
        h->M = cfg->audio_decimation / g1;
        nt = sdr_resampler_taps(h->L, h->M, 0.0, buf.data(), (int)buf.size());
        if (nt < 1) return fm_fail(h, SDR_E_ARG, "audio resampler design failed");
        h->n_rtaps = nt;
        FMCHK(h, dev_alloc((void**)&h->d_rtaps, nt * sizeof(float)));
        FMCHK(h, hipMemcpy(h->d_rtaps, buf.data(), nt * sizeof(float), hipMemcpyHostToDevice));
        {
            const int ntq = (nt + h->L - 1) / h->L, ntp = ntq | 1;
            std::vector<float> tt((size_t)h->L * ntp, 0.0f);
            for (int p = 0; p < h->L; ++p)
                for (int j = 0; j < ntq; ++j) {
                    const int ti = p + (ntq - 1 - j) * h->L;
                    tt[(size_t)p * ntp + j] = ti < nt ? buf[ti] : 0.0f;
                }
            FMCHK(h, dev_alloc((void**)&h->d_rtapsT, tt.size() * sizeof(float)));
            FMCHK(h, hipMemcpy(h->d_rtapsT, tt.data(), tt.size() * sizeof(float), hipMemcpyHostToDevice));
            if (const char* e = getenv("SDR_FM_TAIL")) h->fused_tail = strcmp(e, "split") != 0;
        }
        h->cap2 = n1max / cfg->audio_decim + 2;
        h->cap4 = (h->cap2 * h->L + h->M - 1) / h->M + 2;
        FMCHK(h, dev_alloc((void**)&h->d_a2, (size_t)h->cap2 * sizeof(float)));
        FMCHK(h, dev_alloc((void**)&h->d_a3, (size_t)h->cap2 * sizeof(float)));
        FMCHK(h, dev_alloc((void**)&h->d_audio, (size_t)h->cap4 * sizeof(float)));
        h->chist_len = h->n_ataps - 1 + 2 * cfg->audio_decim;
        h->ehist_len = (h->n_rtaps + h->L - 1) / h->L;
        for (int i = 0; i < 2; ++i) {
            FMCHK(h, dev_alloc((void**)&h->d_chist[i], (size_t)h->chist_len * sizeof(float)));
            FMCHK(h, dev_alloc((void**)&h->d_ehist[i], (size_t)h->ehist_len * sizeof(float)));
            FMCHK(h, memset_done(h->d_chist[i], 0, (size_t)h->chist_len * sizeof(float)));
            FMCHK(h, memset_done(h->d_ehist[i], 0, (size_t)h->ehist_len * sizeof(float)));
        }
        FMCHK(h, dev_alloc((void**)&h->d_deemph, 2 * sizeof(FmDeemphState)));
        FMCHK(h, memset_done(h->d_deemph, 0, 2 * sizeof(FmDeemphState)));
        h->stream = h->A->own_stream;
        sdr_set_stream(h->B, (void*)h->stream);
        return SDR_OK;
    };
    int rc = body();
    if (rc != SDR_OK) {
        std::string keep = h->err;
        sdr_fm_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return SDR_OK;
}

extern "C" int sdr_fm_set_overlap(sdr_fm* h, int on)
{
    if (!h) return SDR_E_ARG;
    (void)hipSetDevice(h->A->device);
    if (fm_join(h)) return fm_fail(h, SDR_E_HIP, "sdr_fm_set_overlap: hipStreamWaitEvent failed");
    FMCHK(h, hipStreamSynchronize(h->stream));
    if (on && !h->bstream) {
        int r = make_side_stream(h->A, h->stream, &h->bstream);
        if (r) return fm_fail(h, r, h->A->err);
        FMCHK(h, hipEventCreateWithFlags(&h->ev_front, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) FMCHK(h, hipEventCreateWithFlags(&h->ev_back[i], hipEventDisableTiming));
        FMCHK(h, dev_alloc((void**)&h->d_tap_alt, (size_t)h->B->cfg.max_batch * (size_t)h->B->tap_stride * sizeof(float)));
    }
    h->overlap = on != 0;
    return SDR_OK;
}

extern "C" int sdr_fm_set_stream(sdr_fm* h, void* s)
{
    if (!h) return SDR_E_ARG;
    if (fm_join(h)) return fm_fail(h, SDR_E_HIP, "sdr_fm_set_stream: hipStreamWaitEvent failed");
    FMCHK(h, hipStreamSynchronize(h->stream));
    h->stream = s ? (hipStream_t)s : h->A->own_stream;
    sdr_set_stream(h->A, (void*)h->stream);
    sdr_set_stream(h->B, (void*)h->stream);
    return SDR_OK;
}

static int fm_hist_update(sdr_fm* h, float** bufs, int& sel, int H, const float* x, long long n)
{
    if (n <= 0 || H <= 0) return SDR_OK;
    float* cur = bufs[sel];
    float* nxt = bufs[sel ^ 1];
    FMCHK(h, tail_update(h->stream, cur, x, (size_t)H * sizeof(float), (size_t)n * sizeof(float), nxt, &h->btails));
    sel ^= 1;
    return SDR_OK;
}

// back to stream start (as after sdr_fm_create)
extern "C" int sdr_fm_reset(sdr_fm* h)
{
    if (!h) return SDR_E_ARG;
    if (fm_join(h)) return fm_fail(h, SDR_E_HIP, "sdr_fm_reset: hipStreamWaitEvent failed");
    FMCHK(h, hipStreamSynchronize(h->stream));
    int rc = chain_restart(h->A);
    if (rc) return fm_fail(h, rc, h->A->err);
    rc = chain_restart(h->B);
    if (rc) return fm_fail(h, rc, h->B->err);
    for (int i = 0; i < 2; ++i) {
        FMCHK(h, memset_done(h->d_chist[i], 0, (size_t)h->chist_len * sizeof(float)));
        FMCHK(h, memset_done(h->d_ehist[i], 0, (size_t)h->ehist_len * sizeof(float)));
    }
    FMCHK(h, memset_done(h->d_deemph, 0, 2 * sizeof(FmDeemphState)));
    h->c_in = h->c_out = h->e_in = h->e_out = 0;
    h->csel = h->esel = h->dsel = 0;
    h->n1 = h->n2 = h->n4 = 0;
    return SDR_OK;
}

// One scheduler pass of the flowgraph over n more input samples.  Every block's state persists: the two
// channel-filter stages in their chain handles (FIR history, input-rotator phase, discriminator's previous
// sample), the audio FIR's history and decimation phase, the de-emphasis delay line (exact, double), the
// audio resampler's history and output phase.  The first call after create / sdr_fm_reset is the capture from
// stream start; any chunking of a stream yields the same audio.
static int fm_process_body(sdr_fm* h, const void* iq, size_t n, int is_dev, size_t* n_audio, hipStream_t* bsp);
extern "C" int sdr_fm_process(sdr_fm* h, const void* iq, size_t n, int is_dev, size_t* n_audio)
{
    if (!h || (!iq && n)) return fm_fail(h, SDR_E_ARG, "sdr_fm_process: bad argument");
    if (n > h->cfg.max_samples) return fm_fail(h, SDR_E_CAPACITY, "sdr_fm_process: chunk larger than max_samples");
    // the blocks' memories (two channel-filter histories, audio FIR, audio resampler) are updated by ONE launch
    // behind the call's last kernel; whatever path the body leaves by, what it collected is flushed
    h->tails.count = 0;
    h->btails.count = 0;
    h->A->tail_defer = h->B->tail_defer = &h->tails;
    (void)hipSetDevice(h->A->device);
    hipStream_t bs = h->stream;       // the stream the back half of THIS call runs on (fm_process_body switches it)
    int rc = fm_process_body(h, iq, n, is_dev, n_audio, &bs);
    h->A->tail_defer = h->B->tail_defer = nullptr;
    if (bs == h->stream) {   // one stream: the four history updates of the call in ONE launch
        for (int i = 0; i < h->btails.count && h->tails.count < 6; ++i) h->tails.job[h->tails.count++] = h->btails.job[i];
        h->btails.count = 0;
    }
    hipError_t e = tail_flush(h->stream, &h->tails);
    hipError_t e2 = tail_flush(bs, &h->btails);
    if (e == hipSuccess) e = e2;
    if (bs != h->stream) {            // whatever the body queued there is what the next call's stage B (and every reader) waits for
        hipError_t e3 = hipEventRecord(h->ev_back[h->bsel], bs);
        h->back_rec[h->bsel] = true;
        h->back_pending = true;
        if (e == hipSuccess) e = e3;
    }
    if (rc == SDR_OK && e != hipSuccess) return fm_fail(h, SDR_E_HIP, std::string("history update: ") + hipGetErrorString(e));
    return rc;
}

static int fm_process_body(sdr_fm* h, const void* iq, size_t n, int is_dev, size_t* n_audio, hipStream_t* bsp)
{
    h->n1 = h->n2 = h->n4 = 0;
    if (n_audio) *n_audio = 0;
    int rc = sdr_process_iq(h->A, iq, n, is_dev, nullptr);
    if (rc) return fm_fail(h, rc, std::string("stage A: ") + h->A->err);
    h->n1 = h->A->last_nout;
    if (h->n1 == 0) return SDR_OK;
    if (h->overlap && h->bstream) {
        // stage B writes the buffer the audio half of the call BEFORE LAST read: wait for that one only
        std::swap(h->B->d_tap, h->d_tap_alt);
        h->bsel ^= 1;
        if (h->back_rec[h->bsel]) FMCHK(h, hipStreamWaitEvent(h->stream, h->ev_back[h->bsel], 0));
    } else if (fm_join(h))
        return fm_fail(h, SDR_E_HIP, "hipStreamWaitEvent failed");
    rc = sdr_process_iq(h->B, h->A->d_tap, (size_t)h->n1, 1, nullptr);
    if (rc) return fm_fail(h, rc, std::string("stage B: ") + h->B->err);
    if (h->overlap && h->bstream) {   // everything below runs behind stage B on the second stream
        FMCHK(h, hipEventRecord(h->ev_front, h->stream));
        FMCHK(h, hipStreamWaitEvent(h->bstream, h->ev_front, 0));
        *bsp = h->bstream;
    }
    const hipStream_t bs = *bsp;
    // audio FIR: outputs are emitted once audio_decim discriminator floats are consumed
    const int D = h->cfg.audio_decim;
    const unsigned long long c_end = (h->c_in + (unsigned long long)h->n1) / (unsigned long long)D;
    h->n2 = (long long)(c_end - h->c_out);
    if (h->n2 > h->cap2) return fm_fail(h, SDR_E_CAPACITY, "sdr_fm_process: audio buffer too small");
    if (h->n2 > 0) {
        const long long in0 = (long long)(h->c_out * (unsigned long long)D) - (long long)h->c_in;
        const int hv = h->c_in < (unsigned long long)h->chist_len ? (int)h->c_in : h->chist_len;
        // matrix-core form when its tile fits LDS and the band is mostly taps (ntaps >= 4*D: at most ~4x padding)
        int nw = 0;
        if (!h->audio_valu && h->n_ataps >= 4 * D)
            for (nw = h->audio_nw; nw >= 1; nw >>= 1)
                if (fir_mfma_lds_bytes(D, h->n_ataps, nw) <= 60 * 1024) break;
        if (nw >= 1) {
            const int TB = 256 * nw;
            hipLaunchKernelGGL(k_fir_fff_mfma, dim3((unsigned)((h->n2 + TB - 1) / TB)), dim3(64 * nw),
                               fir_mfma_lds_bytes(D, h->n_ataps, nw), bs, h->B->d_tap,
                               h->d_chist[h->csel] + (h->chist_len - hv), hv, h->n1, in0, h->n2, h->d_ataps,
                               h->n_ataps, D, h->d_a2);
        } else {
            const int T = 256;
            size_t lds = ((size_t)(T - 1) * D + h->n_ataps) * sizeof(float);
            int blocks = (int)((h->n2 + T - 1) / T);
            hipLaunchKernelGGL(k_fir_fff_decim, dim3(blocks), dim3(T), lds, bs, h->B->d_tap,
                               h->d_chist[h->csel] + (h->chist_len - hv), hv, h->n1, in0, h->n2, h->d_ataps,
                               h->n_ataps, D, h->d_a2);
        }
        FMCHK(h, hipGetLastError());
    }
    rc = fm_hist_update(h, h->d_chist, h->csel, h->chist_len, h->B->d_tap, h->n1);
    if (rc) return rc;
    h->c_in += (unsigned long long)h->n1;
    h->c_out = c_end;
    if (h->n2 == 0) return SDR_OK;
    // audio resampler: output m exists once input floor(m*M/L) does
    const unsigned long long e_in_end = h->e_in + (unsigned long long)h->n2;
    const unsigned long long e_end = (e_in_end * (unsigned long long)h->L + (unsigned long long)h->M - 1ull) / (unsigned long long)h->M;
    h->n4 = (long long)(e_end - h->e_out);
    if (h->n4 > h->cap4) return fm_fail(h, SDR_E_CAPACITY, "sdr_fm_process: audio buffer too small");
    // warm-up length after which p1^k < 2^-60; fall back to the scan kernel for poles near 1
    const int warm = (h->p1 > 0.0 && h->p1 < 1.0) ? (int)ceil(-60.0 * log(2.0) / log(h->p1)) : 1 << 30;
    const FmDeemphState* st_in = h->d_deemph + h->dsel;
    FmDeemphState* st_out = h->d_deemph + (h->dsel ^ 1);
    const int ehv = h->e_in < (unsigned long long)h->ehist_len ? (int)h->e_in : h->ehist_len;
    const int ntq = (h->n_rtaps + h->L - 1) / h->L;
    if (h->fused_tail && warm <= 2048 && ntq - 1 <= FR_HALO && deemph_resample_lds_bytes(warm, ntq, h->L) <= 64 * 1024) {
        // both blocks in one launch: the de-emphasised floats reach the resampler through LDS
        const long long own = FR_RANGE - FR_HALO;
        hipLaunchKernelGGL(k_deemph_resample, dim3((unsigned)((h->n2 + own - 1) / own)), dim3(256),
                           deemph_resample_lds_bytes(warm, ntq, h->L), bs, h->d_a2, h->n2, h->b0, h->p1, warm, h->d_a3, st_in,
                           st_out, h->d_ehist[h->esel] + (h->ehist_len - ehv), ehv, (long long)h->e_in, (long long)h->e_out,
                           h->n4, h->d_rtapsT, ntq, h->L, h->M, h->cfg.volume, h->d_audio);
        FMCHK(h, hipGetLastError());
        h->dsel ^= 1;
    } else {
        if (warm <= 2048) {
            long long chunks = (h->n2 + DEEMPH_CHUNK - 1) / DEEMPH_CHUNK;
            hipLaunchKernelGGL(k_deemph_warm, dim3((unsigned)((chunks + 255) / 256)), dim3(256), deemph_lds_bytes(warm), bs, h->d_a2,
                               h->n2, h->b0, h->p1, warm, h->d_a3, st_in, st_out);
        } else
            hipLaunchKernelGGL(k_deemph, dim3(1), dim3(DEEMPH_THREADS), 0, bs, h->d_a2, h->n2, h->b0, h->p1,
                               h->d_a3, st_in, st_out);
        FMCHK(h, hipGetLastError());
        h->dsel ^= 1;
        if (h->n4 > 0) {
            int blocks = (int)((h->n4 + 255) / 256);
            const size_t rlds = resample_lds_bytes(h->n_rtaps, h->L, h->M);
            if (rlds <= 48 * 1024)
                hipLaunchKernelGGL(k_resample_fff_lds, dim3(blocks), dim3(256), rlds, bs, h->d_a3,
                                   h->d_ehist[h->esel] + (h->ehist_len - ehv), ehv, h->n2, (long long)h->e_in, (long long)h->e_out,
                                   h->n4, h->d_rtaps, h->n_rtaps, h->L, h->M, h->cfg.volume, h->d_audio);
            else   // a polyphase table that does not fit LDS: every step from global memory
                hipLaunchKernelGGL(k_resample_fff, dim3(blocks), dim3(256), 0, bs, h->d_a3,
                                   h->d_ehist[h->esel] + (h->ehist_len - ehv), ehv, h->n2, (long long)h->e_in, (long long)h->e_out,
                                   h->n4, h->d_rtaps, h->n_rtaps, h->L, h->M, h->cfg.volume, h->d_audio);
            FMCHK(h, hipGetLastError());
        }
    }
    rc = fm_hist_update(h, h->d_ehist, h->esel, h->ehist_len, h->d_a3, h->n2);
    if (rc) return rc;
    h->e_in = e_in_end;
    h->e_out = e_end;
    if (n_audio) *n_audio = (size_t)h->n4;
    return SDR_OK;
}

extern "C" int sdr_fm_fetch(sdr_fm* h, int stage, float* out, size_t cap, size_t* n)
{
    if (!h || !n) return fm_fail(h, SDR_E_ARG, "sdr_fm_fetch: bad argument");
    const float* src = nullptr;
    size_t cnt = 0;
    switch (stage) {
    case 0: src = h->A->d_tap; cnt = (size_t)h->n1 * 2; break;
    case 1: src = h->B->d_tap; cnt = (size_t)h->n1; break;
    case 2: src = h->d_a2; cnt = (size_t)h->n2; break;
    case 3: src = h->d_a3; cnt = (size_t)h->n2; break;
    case 4: src = h->d_audio; cnt = (size_t)h->n4; break;
    default: return fm_fail(h, SDR_E_ARG, "sdr_fm_fetch: stage 0..4");
    }
    if (fm_join(h)) return fm_fail(h, SDR_E_HIP, "sdr_fm_fetch: hipStreamWaitEvent failed");
    FMCHK(h, hipStreamSynchronize(h->stream));
    *n = cnt;
    if (!out || cap < cnt || cnt == 0) return SDR_OK;
    FMCHK(h, hipMemcpy(out, src, cnt * sizeof(float), hipMemcpyDeviceToHost));
    return SDR_OK;
}
