/*
 * oracle/dsp_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the GNU Radio 3.7.11 blocks the reference flowgraphs
 * instantiate on the receive-chain hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (sdr_amd/, libsdrhip.so) never does.
 *
 * PARITY STATUS: "parity unpinned" for every function in this file.  GNU Radio
 * and VOLK are third-party dependencies of the reference (GRC 3.7.11 header,
 * /root/reference/TS33C/decoder.grc:2) that are neither vendored nor installed;
 * the reference holds no golden vectors for them.  Each function restates the
 * published upstream algorithm (file named in its comment) and is anchored on
 * the reference call site cited beside it.
 *
 * NUMERIC CONTRACT (what the HIP kernels reproduce bit for bit):
 *   - float32 data, one accumulator per real/imag lane, fused multiply-add
 *     (fmaf) per tap, taps applied oldest-sample-first (ascending input index);
 *   - no contraction other than the explicit fmaf calls (-ffp-contract=off);
 *   - the frequency-translation phase is an exact 64-bit integer accumulator
 *     ("nco64"), evaluated through a 1024-entry table + fixed polynomial; the
 *     GNU Radio float recurrence ("gr rotator") is kept beside it for
 *     comparison only (it is sequential and drifts).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define ORC_API __attribute__((visibility("default")))

/* gr::filter::firdes::win_type (gr-filter/include/gnuradio/filter/firdes.h) */
enum { WIN_HAMMING = 0, WIN_HANN = 1, WIN_BLACKMAN = 2, WIN_RECTANGULAR = 3,
       WIN_KAISER = 4, WIN_BLACKMAN_HARRIS = 5 };

/* ---------------------------------------------------------------- windows */
/* gr-fft/lib/window.cc: windows are returned as std::vector<float>. */
static double izero(double x)
{
    double sum = 1, u = 1, halfx = x / 2.0, temp;
    int n = 1;
    do {
        temp = halfx / (double)n;
        n += 1;
        temp *= temp;
        u *= temp;
        sum += u;
    } while (u >= 1e-21 * sum);
    return sum;
}

static double max_attenuation(int win, double beta)
{
    switch (win) {
    case WIN_HAMMING: return 53;
    case WIN_HANN: return 44;
    case WIN_BLACKMAN: return 74;
    case WIN_RECTANGULAR: return 21;
    case WIN_KAISER: return beta / 0.1102 + 8.7;
    case WIN_BLACKMAN_HARRIS: return 92;
    default: return -1;
    }
}

ORC_API int orc_window(int win, int ntaps, double beta, float* w)
{
    double M = (double)(ntaps - 1);
    int n;
    switch (win) {
    case WIN_HAMMING:
        for (n = 0; n < ntaps; n++) w[n] = 0.54 - 0.46 * cos((2 * M_PI * n) / M);
        break;
    case WIN_HANN:
        for (n = 0; n < ntaps; n++) w[n] = 0.5 - 0.5 * cos((2 * M_PI * n) / M);
        break;
    case WIN_BLACKMAN:
        for (n = 0; n < ntaps; n++)
            w[n] = 0.42 - 0.5 * cos((2 * M_PI * n) / M) + 0.08 * cos((4 * M_PI * n) / M);
        break;
    case WIN_RECTANGULAR:
        for (n = 0; n < ntaps; n++) w[n] = 1;
        break;
    case WIN_KAISER: {
        double ibeta = 1.0 / izero(beta), inm1 = 1.0 / M;
        for (n = 0; n < ntaps; n++) {
            double t = 2 * n * inm1 - 1;
            w[n] = izero(beta * sqrt(1.0 - t * t)) * ibeta;
        }
        break;
    }
    case WIN_BLACKMAN_HARRIS:
        for (n = 0; n < ntaps; n++)
            w[n] = 0.35875 - 0.48829 * cos((2 * M_PI * n) / M) +
                   0.14128 * cos((4 * M_PI * n) / M) - 0.01168 * cos((6 * M_PI * n) / M);
        break;
    default:
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------- firdes.low_pass
 * gr-filter/lib/firdes.cc low_pass()+compute_ntaps().  Call sites:
 * TS33C/live_decoder.grc:917-988 (121 taps), mumbi/FS300.grc:731-802 (97),
 * FM-Radio/FM-Radio.grc:674-746 (33), TFA/decoder.grc:1163-1234 (27),
 * TFA/analyzer2.grc:1075-1147 (7).  Returns ntaps (or -needed if cap small). */
ORC_API int orc_firdes_low_pass(double gain, double fs, double fc, double tw, int win,
                                double beta, float* taps, int cap)
{
    double a = max_attenuation(win, beta);
    int ntaps = (int)(a * fs / (22.0 * tw));
    if ((ntaps & 1) == 0) ntaps++;
    if (ntaps > cap) return -ntaps;
    float* w = (float*)malloc(sizeof(float) * ntaps);
    orc_window(win, ntaps, beta, w);
    int M = (ntaps - 1) / 2;
    double fwT0 = 2 * M_PI * fc / fs;
    for (int n = -M; n <= M; n++) {
        if (n == 0)
            taps[n + M] = fwT0 / M_PI * w[n + M];
        else
            taps[n + M] = sin(n * fwT0) / (n * M_PI) * w[n + M];
    }
    double fmax = taps[0 + M];
    for (int n = 1; n <= M; n++) fmax += 2 * taps[n + M];
    gain /= fmax;
    for (int i = 0; i < ntaps; i++) taps[i] *= gain;
    free(w);
    return ntaps;
}

/* gr-filter/python/filter/rational_resampler.py design_filter().  Call sites:
 * FM-Radio/FM-Radio.grc:1378-1493 (1/4 -> 131 taps, 24/25 -> 821),
 * mumbi/FS300.grc:1691-1746 (1/10 -> 329).  interp/decim already gcd-reduced. */
ORC_API int orc_resampler_design(int interp, int decim, double fbw, float* taps, int cap)
{
    double beta = 7.0, halfband = 0.5, rate = (double)interp / (double)decim, tw, mid;
    if (rate >= 1.0) {
        tw = halfband - fbw;
        mid = halfband - tw / 2.0;
    } else {
        tw = rate * (halfband - fbw);
        mid = rate * halfband - tw / 2.0;
    }
    return orc_firdes_low_pass(interp, interp, mid, tw, WIN_KAISER, beta, taps, cap);
}

/* -------------------------------------------------------------- FIR blocks
 * gr-filter/lib/fir_filter.cc: y[m] = sum_k taps[k] x[m*D-k]; the filter keeps
 * the taps reversed and walks the input in ascending order, i.e. oldest
 * sample first -- that is the summation order fixed here.  `x` points at the
 * first NEW sample; x[-1] .. x[-(ntaps-1)] must be readable (history; zeros
 * at stream start, gr::block::set_history).  nout = number of outputs.
 * Input sample index of output m is m*D + phase0 (phase0 = decimation phase
 * carried across calls, 0 at stream start). */
ORC_API void orc_fir_ccf(const float* x, int64_t nout, const float* taps, int ntaps, int D,
                         int64_t phase0, float* y)
{
    for (int64_t m = 0; m < nout; m++) {
        const float* p = x + 2 * (m * D + phase0 - (ntaps - 1));
        float ar = 0.0f, ai = 0.0f;
        for (int j = 0; j < ntaps; j++) { /* j-th oldest sample, tap ntaps-1-j */
            float t = taps[ntaps - 1 - j];
            ar = fmaf(t, p[2 * j], ar);
            ai = fmaf(t, p[2 * j + 1], ai);
        }
        y[2 * m] = ar;
        y[2 * m + 1] = ai;
    }
}

ORC_API void orc_fir_fff(const float* x, int64_t nout, const float* taps, int ntaps, int D,
                         int64_t phase0, float* y)
{
    for (int64_t m = 0; m < nout; m++) {
        const float* p = x + (m * D + phase0 - (ntaps - 1));
        float a = 0.0f;
        for (int j = 0; j < ntaps; j++) a = fmaf(taps[ntaps - 1 - j], p[j], a);
        y[m] = a;
    }
}

/* complex taps (gr_complex * gr_complex accumulate): 4 fmaf per tap. */
ORC_API void orc_fir_ccc(const float* x, int64_t nout, const float* ctaps, int ntaps, int D,
                         int64_t phase0, float* y)
{
    for (int64_t m = 0; m < nout; m++) {
        const float* p = x + 2 * (m * D + phase0 - (ntaps - 1));
        float ar = 0.0f, ai = 0.0f;
        for (int j = 0; j < ntaps; j++) {
            float tr = ctaps[2 * (ntaps - 1 - j)], ti = ctaps[2 * (ntaps - 1 - j) + 1];
            float xr = p[2 * j], xi = p[2 * j + 1];
            ar = fmaf(tr, xr, ar);
            ar = fmaf(-ti, xi, ar);
            ai = fmaf(tr, xi, ai);
            ai = fmaf(ti, xr, ai);
        }
        y[2 * m] = ar;
        y[2 * m + 1] = ai;
    }
}

/* ------------------------------------------------- frequency translation
 * gr-filter/lib/freq_xlating_fir_filter_XXX_impl.cc build_composite_fir():
 * ctaps[k] = taps[k] * exp(j k 2 pi fc/fs); rotator phase increment
 * exp(-j 2 pi fc/fs * D).  Call sites: FM-Radio/FM-Radio.grc:615-671,
 * TFA/analyzer2.grc:1016-1072 (both taps=[1], D=1). */
ORC_API void orc_xlating_taps(const float* taps, int ntaps, double fc, double fs, float* ctaps)
{
    double th = 2 * M_PI * fc / fs;
    for (int k = 0; k < ntaps; k++) {
        ctaps[2 * k] = (float)((double)taps[k] * cos(k * th));
        ctaps[2 * k + 1] = (float)((double)taps[k] * sin(k * th));
    }
}

/* GNU Radio rotator (gr-blocks/include/gnuradio/blocks/rotator.h): float
 * complex recurrence, renormalised every 512 calls.  Sequential; kept for the
 * "how far is nco64 from GR" comparison only.  state = {re, im, counter}. */
ORC_API void orc_rotator_gr(float* y, int64_t n, double fc, double fs, int D, float* state,
                            int64_t* counter)
{
    float ir = (float)cos(-2 * M_PI * fc / fs * D), ii = (float)sin(-2 * M_PI * fc / fs * D);
    float pr = state[0], pi = state[1];
    int64_t c = *counter;
    for (int64_t m = 0; m < n; m++) {
        c++;
        float xr = y[2 * m], xi = y[2 * m + 1];
        y[2 * m] = xr * pr - xi * pi;
        y[2 * m + 1] = xr * pi + xi * pr;
        float nr = pr * ir - pi * ii, ni = pr * ii + pi * ir;
        pr = nr;
        pi = ni;
        if ((c % 512) == 0) {
            float mag = sqrtf(pr * pr + pi * pi);
            pr /= mag;
            pi /= mag;
        }
    }
    state[0] = pr;
    state[1] = pi;
    *counter = c;
}

/* nco64: exact integer phase.  inc = round(frac(-fc/fs*D) * 2^64); phase of
 * output m = (m * inc) mod 2^64; top 32 bits u: table index u>>22 (1024
 * entries of cos/sin(2 pi i/1024) rounded to float), residual angle
 * r = (u & 0x3fffff) * (2 pi / 2^32) as float, cos r / sin r by the fixed
 * polynomials below, then one complex product.  Every step is a single
 * IEEE-754 float op or fmaf, so any conforming target gives identical bits. */
ORC_API uint64_t orc_nco_inc(double fc, double fs, int D)
{
    double f = -fc / fs * (double)D;
    f -= floor(f); /* [0,1) */
    if (f >= 1.0) f = 0.0;
    return (uint64_t)ldexp(f, 64); /* exact: f has <=53 significant bits */
}

ORC_API void orc_nco_table(float* tab /* 2*1024 */)
{
    for (int i = 0; i < 1024; i++) {
        tab[2 * i] = (float)cos(2 * M_PI * i / 1024.0);
        tab[2 * i + 1] = (float)sin(2 * M_PI * i / 1024.0);
    }
}

static inline void nco_eval(const float* tab, uint64_t ph, float* c, float* s)
{
    uint32_t u = (uint32_t)(ph >> 32);
    uint32_t i = u >> 22;
    float r = (float)(u & 0x3fffffu) * 1.46291807926715968105e-9f; /* 2 pi / 2^32 */
    float r2 = r * r;
    float cr = fmaf(r2, fmaf(r2, 4.16666666666666666667e-2f, -0.5f), 1.0f);
    float sr = r * fmaf(r2, fmaf(r2, 8.33333333333333333333e-3f, -1.66666666666666666667e-1f), 1.0f);
    float tc = tab[2 * i], ts = tab[2 * i + 1];
    *c = fmaf(tc, cr, -(ts * sr));
    *s = fmaf(ts, cr, tc * sr);
}

/* y[m] *= exp(j*phase(m0+m)) in place. */
ORC_API void orc_rotate_nco(float* y, int64_t n, uint64_t inc, uint64_t m0, const float* tab)
{
    for (int64_t m = 0; m < n; m++) {
        float c, s;
        nco_eval(tab, (uint64_t)(m0 + (uint64_t)m) * inc, &c, &s);
        float xr = y[2 * m], xi = y[2 * m + 1];
        y[2 * m] = fmaf(xr, c, -(xi * s));
        y[2 * m + 1] = fmaf(xr, s, xi * c);
    }
}

/* ------------------------------------------------------------------ demod */
/* gr-blocks/lib/complex_to_mag_squared_impl.cc (TS33C/live_decoder.grc:307-349) */
ORC_API void orc_mag2(const float* x, int64_t n, float* y)
{
    for (int64_t i = 0; i < n; i++) y[i] = fmaf(x[2 * i], x[2 * i], x[2 * i + 1] * x[2 * i + 1]);
}

/* gnuradio-runtime/lib/math/fast_atan2f.cc: 255-step table + linear interp.
 * Upstream ships the table as 7-significant-digit literals; restated here as
 * the float nearest to "%.6e" of atan(i/255) (unpinned). */
static float g_atan_tab[258];
static int g_atan_init = 0;
ORC_API void orc_atan_table(float* out /* 257 */)
{
    if (!g_atan_init) {
        char buf[64];
        for (int i = 0; i <= 255; i++) {
            snprintf(buf, sizeof buf, "%.6e", atan((double)i / 255.0));
            g_atan_tab[i] = strtof(buf, NULL);
        }
        g_atan_tab[256] = g_atan_tab[255];
        g_atan_tab[257] = g_atan_tab[255];
        g_atan_init = 1;
    }
    if (out) memcpy(out, g_atan_tab, sizeof(float) * 257);
}

ORC_API float orc_fast_atan2f(float y, float x)
{
    orc_atan_table(NULL);
    float y_abs = fabsf(y), x_abs = fabsf(x), z, base;
    if (!((y_abs > 0.0f) || (x_abs > 0.0f))) return 0.0f;
    if (y_abs < x_abs)
        z = y_abs / x_abs;
    else
        z = x_abs / y_abs;
    if (z < 0.003921569f)
        base = z;
    else {
        float alpha = z * 255.0f;
        int index = ((int)alpha) & 0xff;
        alpha -= (float)index;
        base = g_atan_tab[index];
        base += (g_atan_tab[index + 1] - g_atan_tab[index]) * alpha;
    }
    float angle;
    if (x_abs > y_abs) {
        if (x >= 0.0f)
            angle = (y >= 0.0f) ? base : -base;
        else {
            angle = 3.14159265358979323846f;
            if (y >= 0.0f)
                angle -= base;
            else
                angle = base - angle;
        }
    } else {
        if (y >= 0.0f) {
            angle = 1.57079632679489661923f;
            if (x >= 0.0f)
                angle -= base;
            else
                angle += base;
        } else {
            angle = -1.57079632679489661923f;
            if (x >= 0.0f)
                angle += base;
            else
                angle -= base;
        }
    }
    return angle;
}

/* gr-analog/lib/quadrature_demod_cf_impl.cc: gain*fast_atan2f(imag, real) of
 * x[n]*conj(x[n-1]); x[-1] readable (history 2).  Call sites:
 * TFA/decoder.grc:450-481, inside wfm_rcv FM-Radio/FM-Radio.grc:427-471. */
ORC_API void orc_quad_demod(const float* x, int64_t n, float gain, float* y)
{
    for (int64_t i = 0; i < n; i++) {
        float xr = x[2 * i], xi = x[2 * i + 1], pr = x[2 * i - 2], pi = x[2 * i - 1];
        float tr = fmaf(xr, pr, xi * pi);
        float ti = fmaf(xi, pr, -(xr * pi));
        y[i] = gain * orc_fast_atan2f(ti, tr);
    }
}

/* gr-analog/python/analog/fm_emph.py fm_deemph + gr-filter iir_filter_ffd:
 * y[n] = b0 x[n] + b0 x[n-1] + p1 y[n-1], double state, float I/O.
 * state = {x[n-1] (as double), y[n-1]}.  FM-Radio/FM-Radio.grc:427-471. */
ORC_API void orc_fm_deemph_coeffs(double fs, double tau, double* b0, double* p1)
{
    double w_c = 1.0 / tau;
    double w_ca = 2.0 * fs * tan(w_c / (2.0 * fs));
    double k = -w_ca / (2.0 * fs);
    *p1 = (1.0 + k) / (1.0 - k);
    *b0 = -k / (1.0 - k);
}

ORC_API void orc_iir_deemph(const float* x, int64_t n, double b0, double p1, double* state,
                            float* y)
{
    double xp = state[0], yp = state[1];
    for (int64_t i = 0; i < n; i++) {
        double acc = b0 * (double)x[i];
        acc += b0 * xp;
        acc += p1 * yp;
        xp = (double)x[i];
        yp = acc;
        y[i] = (float)acc;
    }
    state[0] = xp;
    state[1] = yp;
}

/* ------------------------------------------------- pwr_squelch_cc (row a7)
 * gr-analog/lib/pwr_squelch_cc_impl.{h,cc} + squelch_base_cc_impl.cc (3.7.11),
 * ramp = 0: d_pwr = d_iir.filter(re*re + im*im) with
 * filter::single_pole_iir<double,double,double> (y = alpha*x + (1-alpha)*y_prev
 * in double, y_prev starts at 0), the magnitude itself in float; mute() is
 * d_pwr < d_threshold with d_threshold = pow(10, db/10) (double).  gate: muted
 * items are dropped, otherwise zeroed.  TFA/decoder.grc:391-446 (alpha 1e-4,
 * -40 dB, gate True).  state[0] = d_pwr carried between calls.
 * Returns #items written. */
ORC_API int64_t orc_pwr_squelch(const float* x, int64_t n, double db, double alpha, int gate,
                                double* state /* {power} */, float* y)
{
    const double thr = pow(10.0, db / 10.0);
    const double a = alpha, oma = 1.0 - alpha;
    double p = state[0];
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        float m2 = x[2 * i] * x[2 * i] + x[2 * i + 1] * x[2 * i + 1];
        p = a * (double)m2 + oma * p;
        int mute = p < thr;
        if (!mute) {
            y[2 * k] = x[2 * i];
            y[2 * k + 1] = x[2 * i + 1];
            k++;
        } else if (!gate) {
            y[2 * k] = 0;
            y[2 * k + 1] = 0;
            k++;
        }
    }
    state[0] = p;
    return k;
}

/* ------------------------------------------------- slicer chain (row a8)
 * multiply_const -> threshold_ff -> add_const(-0.5) -> binary_slicer_fb ->
 * keep_one_in_n, TS33C/live_decoder.grc:452-553,256-306,878-916,401-451.
 * state[0] = hysteresis bit, counter = keep_one_in_n countdown (starts at n).
 * Returns #bytes written. */
ORC_API int64_t orc_slice(const float* v, int64_t n, float gain, float lo, float hi, int keep_n,
                          int* state, int* counter, int8_t* out)
{
    int s = *state, c = *counter;
    int64_t k = 0;
    for (int64_t i = 0; i < n; i++) {
        float x = gain * v[i];
        if (x > hi)
            s = 1;
        else if (x < lo)
            s = 0;
        float t = (s ? 1.0f : 0.0f) + (-0.5f);
        int8_t b = (t >= 0.0f) ? 1 : 0;
        if (--c <= 0) {
            out[k++] = b;
            c = keep_n;
        }
    }
    *state = s;
    *counter = c;
    return k;
}

/* ------------------------------------------------------ rational resampler
 * gr-filter/lib/rational_resampler_base_XXX_impl.cc: polyphase, output m uses
 * phase (m*M) mod L and newest input floor(m*M/L); sub-filter p = taps[p::L]
 * zero-padded to equal length.  x[-1..-(nt-1)] readable, nt = ceil(ntaps/L).
 * Returns #outputs produced from n_in new inputs (those whose newest input
 * index < n_in). */
ORC_API int64_t orc_resample_ccc(const float* x, int64_t n_in, const float* taps, int ntaps,
                                 int L, int M, float* y, int is_complex)
{
    int nt = (ntaps + L - 1) / L;
    int64_t m = 0;
    for (;; m++) {
        int64_t num = m * (int64_t)M;
        int64_t i = num / L;
        int p = (int)(num % L);
        if (i >= n_in) break;
        if (is_complex) {
            float ar = 0, ai = 0;
            for (int j = 0; j < nt; j++) { /* oldest first: tap index k = nt-1-j */
                int k = nt - 1 - j;
                int ti = p + k * L;
                float t = ti < ntaps ? taps[ti] : 0.0f;
                const float* s = x + 2 * (i - k);
                ar = fmaf(t, s[0], ar);
                ai = fmaf(t, s[1], ai);
            }
            y[2 * m] = ar;
            y[2 * m + 1] = ai;
        } else {
            float a = 0;
            for (int j = 0; j < nt; j++) {
                int k = nt - 1 - j;
                int ti = p + k * L;
                float t = ti < ntaps ? taps[ti] : 0.0f;
                a = fmaf(t, x[i - k], a);
            }
            y[m] = a;
        }
    }
    return m;
}

/* ----------------------------------------------------------- whole chains
 * OOK receive chain as one call, block after block through temporaries the
 * way the GNU Radio scheduler runs it (TS33C/live_decoder.grc:3292-3377):
 * [nco64 rotate . FIR(ctaps|taps, D)] -> mag2 -> gain -> threshold -> slice ->
 * keep_one_in_n.  Input is a whole capture from stream start (zero history).
 * Also the timed body of bench.py's cpu_baseline ("port", 1 core). */
ORC_API int64_t orc_ook_chain(const float* iq, int64_t n, const float* taps, int ntaps,
                              int taps_complex, int D, uint64_t nco_inc, const float* nco_tab,
                              float gain, float lo, float hi, int keep_n, int8_t* bits,
                              float* mag_out /* optional nout floats */)
{
    /* gr::sync_decimator: output m (newest input m*D) is produced once the
     * block can consume D items for it, so a finite capture gives n/D. */
    int64_t nout = n / D;
    if (nout <= 0) return 0;
    int H = ntaps - 1;
    float* xb = (float*)calloc((size_t)(n + H) * 2, sizeof(float));
    memcpy(xb + 2 * H, iq, (size_t)n * 2 * sizeof(float));
    float* y = (float*)malloc((size_t)nout * 2 * sizeof(float));
    if (taps_complex)
        orc_fir_ccc(xb + 2 * H, nout, taps, ntaps, D, 0, y);
    else
        orc_fir_ccf(xb + 2 * H, nout, taps, ntaps, D, 0, y);
    if (nco_tab) orc_rotate_nco(y, nout, nco_inc, 0, nco_tab);
    float* m2 = (float*)malloc((size_t)nout * sizeof(float));
    orc_mag2(y, nout, m2);
    if (mag_out) memcpy(mag_out, m2, (size_t)nout * sizeof(float));
    int st = 0, ctr = keep_n;
    int64_t nb = orc_slice(m2, nout, gain, lo, hi, keep_n, &st, &ctr, bits);
    free(xb);
    free(y);
    free(m2);
    return nb;
}

/* -------------------------------------------------- polyphase channelizer
 * BASELINE.json configs[4]: one wideband capture split into 64 equal sub-bands.
 * The reference has no channelizer block; the semantic
 * anchor is a bank of 64 freq_xlating_fir_filter_ccc(decim = 64, taps,
 * center_freq = c*fs/64) blocks [GR-ext], restated above (orc_xlating_taps +
 * orc_fir_ccc + rotation, which is the identity for these centre frequencies):
 *     y_c[m] = sum_k taps[k] e^{+j 2 pi c k / 64} x[64 m - k].
 * With k = p + 64 q this is a 64-point DFT over the polyphase branch sums
 *     v_p[m] = sum_q taps[p + 64 q] x[64 (m - q) - p],   y_c[m] = sum_p W^{c p} v_p[m],
 * W = e^{+j 2 pi / 64}.  tests/test_oracle_dsp.py checks this function against the
 * bank within 1e-5 relative; the HIP kernel reproduces THIS function bit for bit.
 *
 * Numeric contract: branch sums with one fmaf per tap, oldest sample first
 * (q descending).  DFT as 4 x 16 Cooley-Tukey, p = 16 a + b, c = e + 4 d:
 *   1. for every b a radix-4 butterfly over a (W^16 = +j, no products):
 *        s02 = v0+v2, d02 = v0-v2, s13 = v1+v3, d13 = v1-v3,
 *        u0 = s02+s13, u2 = s02-s13, u1 = d02 + j d13, u3 = d02 - j d13;
 *   2. z[e][b] = u_e * tw[(e b) mod 64] for e = 1..3 (z[0][b] = u0), complex product
 *        (fmaf(dr, wr, -(di*wi)), fmaf(dr, wi, di*wr));
 *   3. for every e a 16-point radix-2 decimation-in-frequency network over b,
 *        stages half = 8,4,2,1: t[l] = a + b, t[l+half] = (a - b) * W16^k with
 *        k = (l mod half) * 8/half, W16^k = tw[4k]; k = 0 -> no product, k = 4 ->
 *        (-di, dr); position pos then holds d = bitrev4(pos).
 * tw[k] = (float cos, float sin)(2 pi k / 64).
 * x: interleaved complex from stream start (samples before it are zero);
 * y: channel-major, y[2*(c*nout + m)]. */
ORC_API void orc_pfb_twiddles(int M, float* tw)
{
    for (int k = 0; k < M; k++) {
        tw[2 * k] = (float)cos(2.0 * M_PI * k / M);
        tw[2 * k + 1] = (float)sin(2.0 * M_PI * k / M);
    }
}

static void orc_cmul(float dr, float di, float wr, float wi, float* orr, float* oi)
{
    *orr = fmaf(dr, wr, -(di * wi));
    *oi = fmaf(dr, wi, di * wr);
}

ORC_API int orc_pfb_channelize(const float* x, int64_t n, const float* taps, int ntaps, int M,
                               const float* tw, int64_t nout, float* y)
{
    if (M != 64) return -1;
    const int P = (ntaps + M - 1) / M;
    float vr[64], vi[64], zr[4][16], zi[4][16];
    for (int64_t m = 0; m < nout; m++) {
        for (int p = 0; p < M; p++) {
            float ar = 0.0f, ai = 0.0f;
            for (int q = P - 1; q >= 0; q--) {
                int k = p + q * M;
                int64_t idx = (m - q) * (int64_t)M - p;
                if (k >= ntaps || idx < 0 || idx >= n) continue; /* zero tap / zero sample: fmaf leaves acc unchanged */
                ar = fmaf(taps[k], x[2 * idx], ar);
                ai = fmaf(taps[k], x[2 * idx + 1], ai);
            }
            vr[p] = ar;
            vi[p] = ai;
        }
        for (int b = 0; b < 16; b++) {
            float s02r = vr[b] + vr[32 + b], s02i = vi[b] + vi[32 + b];
            float d02r = vr[b] - vr[32 + b], d02i = vi[b] - vi[32 + b];
            float s13r = vr[16 + b] + vr[48 + b], s13i = vi[16 + b] + vi[48 + b];
            float d13r = vr[16 + b] - vr[48 + b], d13i = vi[16 + b] - vi[48 + b];
            float ur[4], ui[4];
            ur[0] = s02r + s13r; ui[0] = s02i + s13i;
            ur[2] = s02r - s13r; ui[2] = s02i - s13i;
            ur[1] = d02r - d13i; ui[1] = d02i + d13r;
            ur[3] = d02r + d13i; ui[3] = d02i - d13r;
            zr[0][b] = ur[0];
            zi[0][b] = ui[0];
            for (int e = 1; e < 4; e++) {
                int k = (e * b) & 63;
                orc_cmul(ur[e], ui[e], tw[2 * k], tw[2 * k + 1], &zr[e][b], &zi[e][b]);
            }
        }
        for (int e = 0; e < 4; e++) {
            float* tr = zr[e];
            float* ti = zi[e];
            for (int half = 8; half >= 1; half >>= 1)
                for (int l = 0; l < 16; l++) {
                    if (l & half) continue;
                    float ar = tr[l], ai = ti[l], br = tr[l + half], bi = ti[l + half];
                    tr[l] = ar + br;
                    ti[l] = ai + bi;
                    float dr = ar - br, di = ai - bi;
                    int k = (l & (half - 1)) * (8 / half);
                    if (k == 0) {
                        tr[l + half] = dr;
                        ti[l + half] = di;
                    } else if (k == 4) {
                        tr[l + half] = -di;
                        ti[l + half] = dr;
                    } else
                        orc_cmul(dr, di, tw[8 * k], tw[8 * k + 1], &tr[l + half], &ti[l + half]);
                }
            for (int pos = 0; pos < 16; pos++) {
                int d = ((pos & 1) << 3) | ((pos & 2) << 1) | ((pos & 4) >> 1) | ((pos & 8) >> 3);
                int c = e + 4 * d;
                y[2 * (c * nout + m)] = tr[pos];
                y[2 * (c * nout + m) + 1] = ti[pos];
            }
        }
    }
    return 0;
}
