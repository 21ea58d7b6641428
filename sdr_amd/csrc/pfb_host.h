// Host side of the 64-channel polyphase channelizer (sdr_pfb_create): a normal
// chain handle whose batch rows are (capture, channel) pairs and whose front end
// is k_pfb.  Everything behind the sliced bytes (k_backend, fetches, frame export)
// is shared with the plain chain.
#pragma once

static int pfb_upload_taps(sdr_handle* h)
{
    if (h->proto_complex) return fail(h, SDR_E_ARG, "channelizer: the prototype filter must have real taps");
    const int M = h->pfb_M, n = h->ntaps;
    int P = 0;
    for (int cand : {4, 8, 16})
        if (n <= cand * M) {
            P = cand;
            break;
        }
    if (!P) return fail(h, SDR_E_ARG, "channelizer: at most 16 taps per branch (1024 taps)");
    std::vector<float> t((size_t)P * M, 0.0f);
    for (int k = 0; k < n; ++k) t[k] = h->proto_taps[k];
    if (h->d_pfb_taps) (void)dev_free(h->d_pfb_taps);
    h->d_pfb_taps = nullptr;
    HIPCHK(h, dev_alloc((void**)&h->d_pfb_taps, t.size() * sizeof(float)));
    HIPCHK(h, hipMemcpy(h->d_pfb_taps, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    h->pfb_P = P;
    // streaming history as for a decimating FIR of P*64 taps: ntaps-1 + 2*D samples
    h->hist_len = P * M - 1 + 2 * M;
    if (h->hist_len > h->hist_cap) {
        for (int i = 0; i < 2; ++i) {
            if (h->d_hist[i]) (void)dev_free(h->d_hist[i]);
            h->d_hist[i] = nullptr;
            HIPCHK(h, dev_alloc((void**)&h->d_hist[i], (size_t)h->hist_len * sizeof(float2)));
        }
        h->hist_cap = h->hist_len;
    }
    for (int i = 0; i < 2; ++i) HIPCHK(h, memset_done(h->d_hist[i], 0, (size_t)h->hist_cap * sizeof(float2)));
    reset_stream_counters(h);
    return reset_dec_state(h);
}

extern "C" int sdr_pfb_create(const sdr_chain_cfg* cfg, int channels, sdr_handle** out)
{
    if (!cfg || !out) return fail(nullptr, SDR_E_ARG, "sdr_pfb_create: null argument");
    *out = nullptr;
    if (channels != PFB_M) return fail(nullptr, SDR_E_ARG, "sdr_pfb_create: 64 channels (one per wavefront lane) is the supported size");
    if (cfg->demod == SDR_DEMOD_QUAD || cfg->post_ntaps > 0 || cfg->center_freq != 0.0)
        return fail(nullptr, SDR_E_ARG, "sdr_pfb_create: demod NONE or MAG2, no post filter, no centre frequency");
    if (cfg->max_batch < 1 || cfg->max_batch > (1 << 20) / PFB_M)
        return fail(nullptr, SDR_E_ARG, "sdr_pfb_create: bad max_batch");
    sdr_chain_cfg c2 = *cfg;
    c2.decim = PFB_M;
    c2.max_batch = cfg->max_batch * PFB_M;
    sdr_handle* h = nullptr;
    int r = sdr_create(&c2, &h);
    if (r) return r;
    h->pfb_M = PFB_M;
    auto body = [&]() -> int {
        std::vector<float2> tw(PFB_M);
        for (int k = 0; k < PFB_M; ++k) {
            tw[k].x = (float)cos(2.0 * M_PI * k / PFB_M);
            tw[k].y = (float)sin(2.0 * M_PI * k / PFB_M);
        }
        HIPCHK(h, dev_alloc((void**)&h->d_pfb_tw, tw.size() * sizeof(float2)));
        HIPCHK(h, hipMemcpy(h->d_pfb_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
        return pfb_upload_taps(h);
    };
    r = body();
    if (r) {
        g_err = h->err;
        sdr_destroy(h);
        return r;
    }
    *out = h;
    return SDR_OK;
}

// k_pfb over `ncap` captures (batch) or one streaming call: nout outputs per channel row, bits written
// from patch_off on in every row
static int pfb_launch(sdr_handle* h, const float2* x, long long stride, long long n, int ncap, const CallGeom& g,
                      const float2* hist, int hist_len, long long patch_off)
{
    const sdr_chain_cfg& c = h->cfg;
    if (g.tiles > h->tsum_stride) return fail(h, SDR_E_CAPACITY, "tile table too small");
    if (g.nout >= 0x7fffffffLL) return fail(h, SDR_E_CAPACITY, "capture too long for the channelizer");
    if (c.demod == SDR_DEMOD_NONE && !h->d_tap)
        return fail(h, SDR_E_ARG, "demod NONE needs cfg.tap_demod (complex output buffer)");
    if (g.nout <= 0) return SDR_OK;
    PfbParams Q;
    memset(&Q, 0, sizeof Q);
    Q.x = x;
    Q.hist = hist;
    Q.hist_len = hist_len;
    Q.in0 = g.in0;
    Q.stride = stride;
    Q.n = n;
    Q.nout = g.nout;
    Q.vec_ok = ((((uintptr_t)x) & 15) == 0 && (stride % 2) == 0) ? 1 : 0;
    Q.taps = h->d_pfb_taps;
    Q.tw = h->d_pfb_tw;
    Q.demod = c.demod;
    Q.gain = c.demod_gain;
    Q.lo = c.thr_lo;
    Q.hi = c.thr_hi;
    Q.keep_n = c.keep_n;
    Q.keep_c0 = g.keep_c0;
    Q.bits = h->d_bits + patch_off;
    Q.bits_stride = h->bits_stride;
    Q.tap = (c.tap_demod || c.demod == SDR_DEMOD_NONE) ? h->d_tap : nullptr;
    Q.tap_stride = h->tap_stride;
    Q.tsum = h->d_tsum;
    Q.tiles_per_row = g.tiles;
    size_t lds = ((size_t)(h->pfb_P + PFB_TILE - 1) * PFB_M + 2) * sizeof(float2);
    if (lds < (size_t)PFB_NW * PFB_SLAB * sizeof(float2)) lds = (size_t)PFB_NW * PFB_SLAB * sizeof(float2);
    dim3 grid(g.tiles, ncap), block(PFB_THREADS);
    switch (h->pfb_P) {
    case 4: hipLaunchKernelGGL(k_pfb<4>, grid, block, lds, h->stream, Q); break;
    case 8: hipLaunchKernelGGL(k_pfb<8>, grid, block, lds, h->stream, Q); break;
    default: hipLaunchKernelGGL(k_pfb<16>, grid, block, lds, h->stream, Q); break;
    }
    HIPCHK(h, hipGetLastError());
    return SDR_OK;
}

static void pfb_geom(const sdr_handle* h, long long nout, long long in0, long long keep_c0, CallGeom& g)
{
    memset(&g, 0, sizeof g);
    const long long kn = h->cfg.keep_n;
    g.nout = nout;
    g.in0 = in0;
    g.keep_c0 = keep_c0;
    g.ne = nout > keep_c0 ? (nout - keep_c0 + kn - 1) / kn : 0;
    g.T = 256;
    g.tile_out = PFB_TILE;
    g.tiles = (int)((nout + PFB_TILE - 1) / PFB_TILE);
    g.use_lds = true;
}

static int pfb_batch(sdr_handle* h, const void* iq, size_t n, size_t stride, int ncap, int is_dev, sdr_out* out)
{
    const sdr_chain_cfg& c = h->cfg;
    const int M = h->pfb_M;
    if (ncap * M > c.max_batch || n > c.max_samples)
        return fail(h, SDR_E_CAPACITY, "sdr_batch_process: batch larger than the channelizer was created for");
    if (stride < n) return fail(h, SDR_E_ARG, "sdr_batch_process: stride < n_per_capture");
    if (h->in_fmt != SDR_IN_FC32) return fail(h, SDR_E_ARG, "channelizer: complex64 input only");
    (void)hipSetDevice(h->device);
    if (h->overlap) {
        int r = ov_next_buffers(h);
        if (r) return r;
    }
    OvScope ovs(h, h->overlap);
    const float2* x = (const float2*)iq;
    if (!is_dev) {
        int r = ensure_iq_staging(h, (size_t)ncap * stride);
        if (r) return r;
        HIPCHK(h, hipMemcpyAsync(h->d_iq, iq, (size_t)((ncap - 1) * stride + n) * sizeof(float2), hipMemcpyHostToDevice,
                                 h->stream));
        x = h->d_iq;
    }
    CallGeom g;
    pfb_geom(h, (long long)(n / M), 0, c.keep_n - 1, g);
    if (prof_begin(h)) HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING)], h->stream));
    int r = pfb_launch(h, x, (long long)stride, (long long)n, ncap, g, nullptr, 0, 0);
    if (r) return r;
    int run = 1;
    if (is_stream_decoder(c.decoder))
        run = ((double)g.ne * 1e6 / h->dec_rate > decoder_reset_limit(c.decoder)) ? 1 : 0;
    r = launch_back(h, ncap * M, g, 0, g.ne, 1, 0, run);
    if (r) return r;
    h->last_ncap = ncap * M;
    h->last_nb = g.ne;
    h->last_nout = g.nout;
    h->last_bits_off = 0;
    h->last_keep_c0 = g.keep_c0;
    h->last_out0 = 0;
    return finish_out(h, out, (unsigned long long)n * ncap, g.ne);
}

// One streaming call on a channelizer handle (max_batch == 64 rows: one wideband stream): the 64 channel
// rows advance together -- same FIR history ((P-1)*64+63 input samples and the 64-sample output phase),
// same keep_one_in_n counter and rawBuffer fill; threshold bit, decoder and rule state are per row.
static int pfb_stream(sdr_handle* h, const void* iq, size_t n, int is_dev, sdr_out* out)
{
    const sdr_chain_cfg& c = h->cfg;
    const int M = h->pfb_M;
    if (c.max_batch != M) return fail(h, SDR_E_ARG, "sdr_process_iq: a channelizer streams ONE wideband capture (create it with max_batch 1)");
    if (n > c.max_samples) return fail(h, SDR_E_CAPACITY, "sdr_process_iq: chunk larger than max_samples");
    if (h->in_fmt != SDR_IN_FC32) return fail(h, SDR_E_ARG, "channelizer: complex64 input only");
    (void)hipSetDevice(h->device);
    const float2* x = (const float2*)iq;
    if (!is_dev && n) {
        int r = ensure_iq_staging(h, n);
        if (r) return r;
        HIPCHK(h, hipMemcpyAsync(h->d_iq, iq, n * sizeof(float2), hipMemcpyHostToDevice, h->stream));
        x = h->d_iq;
    }
    const unsigned long long A = h->in_count;
    const unsigned long long m_end = (A + n) / (unsigned long long)M;
    const long long nout = (long long)(m_end - h->out_count);
    CallGeom g;
    pfb_geom(h, nout, (long long)(h->out_count * M) - (long long)A, h->keep_ctr - 1, g);
    const long long total = h->raw_n + g.ne;
    if (total > h->bits_stride) return fail(h, SDR_E_CAPACITY, "rawBuffer capacity exceeded");
    int run = 0;
    if (is_stream_decoder(c.decoder))
        run = ((double)total * 1e6 / h->dec_rate > decoder_reset_limit(c.decoder)) ? 1 : 0;
    else if (c.decoder != SDR_DEC_NONE)
        run = 1;
    const long long off = is_stream_decoder(c.decoder) ? h->raw_n : 0;
    const long long nb_total = off == 0 ? g.ne : total;
    if (prof_begin(h)) HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING)], h->stream));
    int r = pfb_launch(h, x, 0, (long long)n, 1, g, hist_ptr(h), hist_valid(h), off);
    if (r) return r;
    r = launch_back(h, M, g, off, nb_total, 0, 1, run);
    if (r) return r;
    h->last_ncap = M;
    h->last_nb = g.ne;
    h->last_nout = g.nout;
    h->last_bits_off = off;
    h->last_keep_c0 = g.keep_c0;
    h->last_out0 = h->out_count;
    h->carry_last = h->carry_next;
    h->carry_next = -1;
    if (is_stream_decoder(c.decoder)) h->raw_n = run ? 0 : total;
    if (n) {
        const size_t H = (size_t)h->hist_len, ss = sizeof(float2);
        const char* xb = reinterpret_cast<const char*>(x);
        char* cur = reinterpret_cast<char*>(h->d_hist[h->hist_sel]);
        char* nxt = reinterpret_cast<char*>(h->d_hist[h->hist_sel ^ 1]);
        HIPCHK(h, tail_update(h->stream, cur, xb, H * ss, n * ss, nxt));
        h->hist_sel ^= 1;
    }
    h->in_count = A + n;
    h->out_count = m_end;
    if (g.ne == 0)
        h->keep_ctr -= nout;
    else {
        const long long last = g.keep_c0 + (g.ne - 1) * c.keep_n;
        h->keep_ctr = c.keep_n - (nout - 1 - last);
    }
    return finish_out(h, out, n, g.ne);
}

// Run another decoder over the sliced bytes of the last batch (every row decoded
// from a fresh decoder object, as in sdr_batch_process); the frame table is replaced.
extern "C" int sdr_redecode(sdr_handle* h, int decoder, double decoder_rate)
{
    if (!h || decoder <= SDR_DEC_NONE || decoder > SDR_DEC_VELUX) return fail(h, SDR_E_ARG, "sdr_redecode: bad decoder");
    if (h->cfg.demod == SDR_DEMOD_NONE || h->skip_slice) return fail(h, SDR_E_ARG, "sdr_redecode: the chain has no sliced stream");
    if (h->last_ncap < 1) return fail(h, SDR_E_ARG, "sdr_redecode: no batch processed yet");
    if (h->last_bits_off != 0 || h->raw_n != 0) return fail(h, SDR_E_ARG, "sdr_redecode: only after sdr_batch_process");
    (void)hipSetDevice(h->device);
    OvScope ovs(h, h->overlap);   // back-end work only: on the result stream when batches overlap
    const int keep_dec = h->cfg.decoder;
    const double keep_rate = h->dec_rate;
    h->cfg.decoder = decoder;
    if (decoder_rate > 0) h->dec_rate = decoder_rate;
    CallGeom g;
    memset(&g, 0, sizeof g);
    g.ne = h->last_nb;
    g.T = 256;
    g.tile_out = 256;
    int run = 1;
    if (is_stream_decoder(decoder))
        run = ((double)g.ne * 1e6 / h->dec_rate > decoder_reset_limit(decoder)) ? 1 : 0;
    if (h->profile) HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING)], h->stream));
    // bits are final (tsum == nullptr): phase 0 is skipped, thr_state untouched
    h->reuse_next = true;   // the run table of the batch's own pass (or of an earlier redecode) is still valid
    int r = launch_back(h, h->last_ncap, g, 0, g.ne, 1, 0, run);
    h->reuse_next = false;
    h->cfg.decoder = keep_dec;
    h->dec_rate = keep_rate;
    return r;
}

// Several decoders over the sliced rows of the last batch in ONE launch: workgroup v decodes row v % rows
// with decoders[v / rows], each from a fresh decoder object; the run table of the batch's own pass is reused.
// Four k_backend launches over the 64 channel rows cost 0.15 ms (latency-bound, 64 workgroups each); one
// launch of 64 x n workgroups costs what one of them did.
extern "C" int sdr_redecode_multi(sdr_handle* h, const int32_t* decoders, int n, double decoder_rate)
{
    if (!h || !decoders || n < 1 || n > 8) return fail(h, SDR_E_ARG, "sdr_redecode_multi: 1..8 decoders");
    for (int i = 0; i < n; ++i)
        if (decoders[i] <= SDR_DEC_NONE || decoders[i] > SDR_DEC_VELUX) return fail(h, SDR_E_ARG, "sdr_redecode_multi: bad decoder");
    if (h->cfg.demod == SDR_DEMOD_NONE || h->skip_slice) return fail(h, SDR_E_ARG, "sdr_redecode_multi: the chain has no sliced stream");
    if (h->last_ncap < 1) return fail(h, SDR_E_ARG, "sdr_redecode_multi: no batch processed yet");
    if (h->last_bits_off != 0 || h->raw_n != 0) return fail(h, SDR_E_ARG, "sdr_redecode_multi: only after sdr_batch_process");
    (void)hipSetDevice(h->device);
    OvScope ovs(h, h->overlap);   // as sdr_redecode
    const int rows = h->last_ncap;
    const long long need = (long long)rows * n;
    if (need > h->m_cap) {
        OV_JOIN(h);
        HIPCHK(h, hipStreamSynchronize(h->stream));
        for (void* p : {(void*)h->d_m_cur, (void*)h->d_m_st, (void*)h->d_m_frames})
            if (p) (void)dev_free(p);
        h->d_m_cur = nullptr;
        h->d_m_st = nullptr;
        h->d_m_frames = nullptr;
        h->m_cap = 0;
        HIPCHK(h, dev_alloc((void**)&h->d_m_cur, (size_t)(need * SDR_CUR_CAP) * sizeof(double)));
        HIPCHK(h, dev_alloc((void**)&h->d_m_st, (size_t)need * sizeof(DecState)));
        HIPCHK(h, dev_alloc((void**)&h->d_m_frames, (size_t)(need * h->cfg.max_frames) * sizeof(sdr_frame)));
        HIPCHK(h, memset_done(h->d_m_frames, 0, (size_t)(need * h->cfg.max_frames) * sizeof(sdr_frame)));
        h->m_cap = need;
    }
    const double rate = decoder_rate > 0 ? decoder_rate : h->dec_rate;
    if (!h->runs_valid) {
        // no run table from the batch's own pass (it ran without a decoder): one ordinary pass builds it
        int r = sdr_redecode(h, decoders[0], decoder_rate);
        if (r) return r;
        if (!h->runs_valid) return fail(h, SDR_E_ARG, "sdr_redecode_multi: rows too short for the first decoder's pass");
    }
    BackParams Bp;
    memset(&Bp, 0, sizeof Bp);
    Bp.bits = h->d_bits;
    Bp.bits_stride = h->bits_stride;
    Bp.nb = h->last_nb;
    Bp.keep_n = h->cfg.keep_n;
    Bp.thr_init = h->cfg.thr_init;
    Bp.run_decoder = 1;
    Bp.decoder = decoders[0];
    Bp.fresh = 1;
    Bp.dec_rate = rate;
    Bp.runs = h->d_runs;
    Bp.runs_cap = h->runs_cap;
    Bp.cur = h->d_m_cur;
    Bp.st = h->d_m_st;
    Bp.frames = h->d_m_frames;
    Bp.max_frames = h->cfg.max_frames;
    Bp.pow_m13 = h->d_pow;
    Bp.log2n = h->d_log2;
    Bp.runs_n = h->d_runs_n;
    Bp.reuse_runs = 1;
    Bp.ndec = n;
    Bp.vrows = rows;
    for (int i = 0; i < n; ++i) Bp.dec_pack |= (unsigned)decoders[i] << (4 * i);
    // a stream decoder on rows shorter than its reset_limit would only fill its rawBuffer: nothing to report
    for (int i = 0; i < n; ++i)
        if (is_stream_decoder(decoders[i]) && !((double)Bp.nb * 1e6 / rate > decoder_reset_limit(decoders[i])))
            return fail(h, SDR_E_ARG, "sdr_redecode_multi: rows shorter than a stream decoder's reset limit");
    hipStream_t bs;
    {
        int rb = result_stream_begin(h, &bs);
        if (rb) return rb;
    }
    if (h->profile) {
        HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING)], bs));
        HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING) + 1], bs));
    }
    hipLaunchKernelGGL(k_backend, dim3((unsigned)need), dim3(BK_THREADS), 0, bs, Bp);
    HIPCHK(h, hipGetLastError());
    if (h->profile) {
        HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING) + 2], bs));
        h->pcount++;
    }
    h->m_ndec = n;
    h->m_rows = rows;
    return result_stream_end(h);
}

// frames of decoder number k (0-based position in the list) of the last sdr_redecode_multi
extern "C" int sdr_fetch_frames_multi(sdr_handle* h, int k, sdr_frame* outp, size_t cap, size_t* n)
{
    if (!h || !n || k < 0 || k >= h->m_ndec) return fail(h, SDR_E_ARG, "sdr_fetch_frames_multi: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int B = h->m_rows;
    *n = 0;
    std::vector<DecState> st(B);
    HIPCHK(h, hipMemcpy(st.data(), h->d_m_st + (size_t)k * B, (size_t)B * sizeof(DecState), hipMemcpyDeviceToHost));
    size_t total = 0;
    int overflow = 0;
    for (int i = 0; i < B; ++i) {
        total += st[i].n_frames;
        overflow |= st[i].overflow;
    }
    *n = total;
    if (outp && cap) {
        std::vector<sdr_frame> all((size_t)B * h->cfg.max_frames);
        HIPCHK(h, hipMemcpy(all.data(), h->d_m_frames + (size_t)k * B * h->cfg.max_frames, all.size() * sizeof(sdr_frame),
                            hipMemcpyDeviceToHost));
        size_t j2 = 0;
        for (int i = 0; i < B && j2 < cap; ++i)
            for (int j = 0; j < st[i].n_frames && j2 < cap; ++j) {
                outp[j2] = all[(size_t)i * h->cfg.max_frames + j];
                outp[j2].capture = i;   // the kernel numbered the virtual capture (row + k*rows)
                ++j2;
            }
    }
    if (overflow) return fail(h, SDR_E_OVERFLOW, "device table overflow (runs/frames/symbols), code " + std::to_string(overflow));
    return SDR_OK;
}

