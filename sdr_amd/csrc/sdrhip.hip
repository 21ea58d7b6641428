// libsdrhip.so -- host side of the C ABI declared in include/sdrhip.h.
// gfx950 only.  Filter design (firdes), table set-up, buffer ownership, launch
// geometry and the streaming state that the GNU Radio scheduler / the Python
// decoder object keep between work() calls.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "sdr_common.h"
#include "frontend_kernels.h"
#include "frontend_mfma.h"
#include "frontend_win.h"
#include "backend_kernels.h"
#include "backend_main.h"
#include "fm_kernels.h"
#include "squelch_kernels.h"
#include "pfb_kernels.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define SDR_PROF_RING 256
#define SDR_SEG_MIN_ROW (32 * 1024)    // sliced rows this long take the segment- and piece-parallel back end (r02: 128 -> 32 KiB, the channelizer rows)
struct sdr_handle {
    sdr_chain_cfg cfg;
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;

    std::vector<float> proto_taps;  // as given (real, or re/im pairs)
    bool proto_complex = false;
    int ntaps = 1;
    bool rt_complex = false;
    float* d_rtaps = nullptr;
    FrontKTaps ktaps;        // the same reversed taps for the kernel-argument path (ntaps <= 64)
    int ktaps_n = 0;
    std::vector<float> post_taps;
    float* d_post_rtaps = nullptr;
    float2* d_nco_tab = nullptr;
    float* d_atan_tab = nullptr;
    double* d_pow = nullptr;
    double* d_log2 = nullptr;
    unsigned long long nco_inc = 0;

    float2* d_iq = nullptr;       // staging for host inputs
    size_t iq_cap = 0;
    float2* d_hist[2] = {nullptr, nullptr};
    int hist_sel = 0, hist_len = 0, hist_cap = 0;
    int8_t* d_bits = nullptr;
    long long bits_stride = 0;
    float* d_tap = nullptr;       // demod floats (or complex pairs)
    float* d_tap2 = nullptr;      // post-filter output
    unsigned char* d_pack = nullptr;  // rows_to_host staging
    size_t pack_cap = 0;
    float* d_post_hist[2] = {nullptr, nullptr};  // streaming history of the post-demod filter
    int post_hist_sel = 0;
    long long tap_stride = 0;
    TileSum* d_tsum = nullptr;
    int tsum_stride = 0;
    int32_t* d_runs = nullptr;
    int32_t* d_runs_seg = nullptr;  // rows of SDR_SEG_MIN_ROW bytes and more: per-segment run starts (k_runs_seg)
    int32_t* d_seg_cnt = nullptr;
    int32_t* d_runs_n = nullptr;    // runs per capture left by the last full back-end pass
    int32_t* d_pieces_hdr = nullptr; // long rows: 8 ints per capture for k_decode_pieces
    bool runs_valid = false;        // ... and whether they describe the current sliced bytes
    bool reuse_next = false;        // set by sdr_redecode around its launch_back
    int lvl_override = 0;           // set by sdr_process_runs around its launch_back (BackParams.lvl_override)
    long long runs_cap = 0;
    bool no_quad_inst = false;   // env SDR_NO_QUAD_INST: discriminator chains stay on the generic k_frontend_win
    double* d_cur = nullptr;
    DecState* d_st = nullptr;
    // sdr_redecode_multi: tables for rows x decoders virtual captures
    double* d_m_cur = nullptr;
    DecState* d_m_st = nullptr;
    sdr_frame* d_m_frames = nullptr;
    long long m_cap = 0;
    int m_ndec = 0, m_rows = 0;
    RuleState* d_rule = nullptr;    // callback-rule decoders: state between streaming calls (one per streaming row)
    size_t rule_rows = 1;
    sdr_frame* d_frames = nullptr;

    // streaming state tracked on the host (pure functions of the sample counts)
    unsigned long long in_count = 0, out_count = 0;
    long long keep_ctr = 1;
    long long raw_n = 0;
    bool stream_started = false;

    // pwr_squelch_cc between channel filter and demodulator (sdr_set_squelch)
    bool squelch_on = false;
    int sq_gate = 1;
    double sq_alpha = 0, sq_thr = 0;
    float2* d_sq_y = nullptr;     // channel-filter output
    float2* d_sq_z = nullptr;     // gated items (+1 leading history slot per capture)
    long long sq_stride = 0;
    int32_t* d_sq_cnt = nullptr;
    SquelchState* d_sq_st = nullptr;

    sdr_handle* pre = nullptr;  // upstream channel-filter stage (sdr_set_prefilter)
    int pfb_M = 0, pfb_P = 0;   // !=0: channelizer handle (sdr_pfb_create): rows = (capture, channel), front end = k_pfb
    float* d_pfb_taps = nullptr;
    float2* d_pfb_tw = nullptr;
    int in_fmt = SDR_IN_FC32;   // sample format of this stage's input (sdr_set_input_format)
    size_t ss() const { return in_fmt == SDR_IN_CU8 ? 2 : sizeof(float2); }  // bytes per input sample
    int pipe_mode = 1;   // k_frontend_pipe: 0 never, 1 for cu8 input (default), 2 for complex64 too (env SDR_PIPE; measured slower there)
    int pipe_tpw = 1;    // tiles per workgroup in that kernel (env SDR_TPW); measured on cu8: 1 -> 0.353 ms, 2 -> 0.414, 4 -> 0.447, 8 -> 0.543
    int front_mode = 0;  // 0 auto, 1 LDS/VALU kernel only, 2 MFMA kernel whenever it fits, 3 sliding-window kernel whenever it fits (env SDR_FRONTEND=auto|valu|mfma|win)
    // ablation switches (env SDR_NO_KTAPS / SDR_NO_RAW8 / SDR_NO_REUSE / SDR_NO_SEG / SDR_NO_PIECES, SDR_TILE):
    // read ONCE in sdr_create, never on the call path
    bool no_ktaps = false, no_raw8 = false, no_reuse = false, no_seg = false, no_pieces = false;
    int mfma2_tpw = 0;
    bool mfma2_ok = false;    // k_frontend_mfma2 may be launched (its > 64 KiB of dynamic LDS was granted; SDR_NO_MFMA2 unset)
    int tile_override = 0;
    long long seg_min_row = SDR_SEG_MIN_ROW;   // rows this long take the segment- and piece-parallel back end (env SDR_SEG_MIN)

    // threshold-only discriminator (FrontParams.quad_fast): set up once at create
    bool quad_fast = false;
    float q_c_hi = 0, q_s_hi = 0, q_c_lo = 0, q_s_lo = 0, q_band = 0;

    // internal modes used by the FM chain (fm_host.h)
    struct TailBatch* tail_defer = nullptr;   // set by the FM object around its calls: history updates are collected, not launched
    bool prerotate = false;   // center_freq applies to the INPUT samples (xlating taps=[1] then FIR)
    bool skip_slice = false;  // demod floats only: no slicer, no back end

    // last call
    int last_ncap = 0;
    long long last_nb = 0, last_nout = 0, last_bits_off = 0;
    long long last_keep_c0 = 0;
    unsigned long long last_out0 = 0;   // absolute output index of the last call's output 0
    int carry_next = 0, carry_last = -1; // threshold_ff carry-in of the next / last streaming call as far as the host knows it (-1 unknown)
    double dec_rate = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // per-launch profiling ring
    bool profile = false;        // events are recorded in the call in progress (decided by prof_begin)
    int prof_every = 0;          // sdr_profile_enable(h, N): every N-th call is timed; 0 = off
    unsigned long long prof_calls = 0;
    std::vector<hipEvent_t> pev;   // 3 events per slot: front start, front end/back start, back end
    unsigned long long pcount = 0;

    // overlapped batches (sdr_set_overlap): the back end of a batch, and what reads its results on the device
    // (sdr_redecode*, sdr_export_*), run on a second stream beside the front end of the next batch.  The front
    // end alternates between two (sliced bytes, tile summaries) pairs; everything else the back end touches
    // is only ever touched on that second stream, in order.
    bool overlap = false;
    hipStream_t bstream = nullptr;
    bool ov_call = false;        // the current call puts its back-end side on bstream
    bool ov_pending = false;     // bstream holds work h->stream has not been ordered behind
    bool ov_front_covered = false;   // bstream already waits for everything the library has queued on h->stream
    int ov_sel = 0, ov_last = 0; // pair the last batch wrote; pair whose event was recorded last
    int8_t* d_bits2 = nullptr;
    TileSum* d_tsum2 = nullptr;
    hipEvent_t ov_front = nullptr, ov_back[2] = {nullptr, nullptr};
    bool ov_back_rec[2] = {false, false};
};

static thread_local std::string g_err;

static int fail(sdr_handle* h, int code, const std::string& msg)
{
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

#define HIPCHK(h, call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(h, SDR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));     \
    } while (0)

// ------------------------------------------------------------------ overlapped batches
// Entry points that read or replace what a back end writes call ov_join first: h->stream waits for the last
// event of the result stream, after which the handle behaves as if it had one stream.
// one wave busy for `cycles` shader clocks (bounded by the clock: every wave ends); sdr_set_overlap's queue probe
__global__ void k_probe_spin(long long cycles, int* sink)
{
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) {
    }
    if (sink && threadIdx.x == 0) *sink = 0;
}

// SDR_OV_TRACE=1 (diagnostics, read once): every cross-stream record / wait of the overlap pattern on stderr
static bool ov_trace()
{
    static const bool on = getenv("SDR_OV_TRACE") != nullptr;
    return on;
}
#define OV_TRACE(...)                                 \
    do {                                              \
        if (ov_trace()) fprintf(stderr, __VA_ARGS__); \
    } while (0)

// A second stream of the same priority as `main_stream` that sits on ANOTHER hardware queue (sdr_set_overlap,
// sdr_fm_set_overlap; why not a higher priority and why the probe: see sdr_set_overlap).
static int fail(sdr_handle* h, int code, const std::string& msg);
static int make_side_stream(sdr_handle* h, hipStream_t main_stream, hipStream_t* out)
{
    hipStream_t rejected[4];
    int nrej = 0;
    *out = nullptr;
    for (int attempt = 0; attempt < 4 && !*out; ++attempt) {
        hipStream_t cand = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
        if (e != hipSuccess) return fail(h, SDR_E_HIP, std::string("hipStreamCreateWithFlags: ") + hipGetErrorString(e));
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, main_stream, 5000000LL, (int*)nullptr);
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, cand, 0LL, (int*)nullptr);
        e = hipStreamSynchronize(cand);
        const bool apart = e == hipSuccess && hipStreamQuery(main_stream) == hipErrorNotReady;
        (void)hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(main_stream);
        if (e != hipSuccess) {
            (void)hipStreamDestroy(cand);
            for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
            return fail(h, SDR_E_HIP, std::string("side-stream probe: ") + hipGetErrorString(e));
        }
        OV_TRACE("[ov] side-stream candidate %d: %s\n", attempt, apart ? "on another hardware queue" : "shares the main stream's queue");
        if (apart || attempt == 3)
            *out = cand;
        else
            rejected[nrej++] = cand;   // kept alive until the choice is made: a destroyed stream's queue would be handed out again
    }
    for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
    return SDR_OK;
}

static int ov_join(sdr_handle* h)
{
    if (h && h->ov_pending) {
        (void)hipSetDevice(h->device);
        OV_TRACE("[ov %p] join: stream waits back[%d]\n", (void*)h, h->ov_last);
        HIPCHK(h, hipStreamWaitEvent(h->stream, h->ov_back[h->ov_last], 0));
        h->ov_pending = false;
    }
    if (h) h->ov_front_covered = false;   // the caller is about to queue work on h->stream outside an overlapped call
    return SDR_OK;
}
#define OV_JOIN(h)              \
    do {                        \
        int j_ = ov_join(h);    \
        if (j_) return j_;      \
    } while (0)

// first event of a processing call's triple: decides whether THIS call -- and the sdr_redecode* calls behind it, which
// keep the decision -- is timed (every prof_every-th one).  A timing event is a packet of its own on the queue, ~3 us
// each: three per step were 10 us of the 0.44-ms headline step
static inline bool prof_begin(sdr_handle* h)
{
    h->profile = h->prof_every > 0 && (h->prof_calls++ % (unsigned long long)h->prof_every) == 0;
    return h->profile;
}

struct OvScope {   // marks the calls whose back-end side goes to the result stream
    sdr_handle* h;
    bool prev;
    OvScope(sdr_handle* h_, bool on) : h(h_), prev(h_->ov_call) { h->ov_call = on; }
    ~OvScope() { h->ov_call = prev; }
};

// the stream the back-end side of this call runs on; inside an overlapped call it is ordered behind
// everything queued on h->stream so far (the front end just launched)
// user_bufs: the call writes buffers of the caller (sdr_export_*), whose earlier uses are ordered on h->stream -- always
// order behind it.  A call that touches the handle's own tables only (back end, sdr_redecode*) needs the wait once
// per front end: a second event record + cross-queue wait per step is a barrier packet the result queue sits
// behind for ~20 us although its dependency is long satisfied (kernel timeline of the channelizer workload:
// 26 us idle in front of the sdr_redecode_multi launch of every step).
static int result_stream_begin(sdr_handle* h, hipStream_t* bs, bool user_bufs = false)
{
    *bs = h->stream;
    if (!h->ov_call) return SDR_OK;
    if (user_bufs || !h->ov_front_covered) {
        OV_TRACE("[ov %p] begin: record front on stream, bstream waits front\n", (void*)h);
        HIPCHK(h, hipEventRecord(h->ov_front, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->bstream, h->ov_front, 0));
        h->ov_front_covered = true;
    }
    *bs = h->bstream;
    return SDR_OK;
}
static int result_stream_end(sdr_handle* h)
{
    if (!h->ov_call) return SDR_OK;
    OV_TRACE("[ov %p] end: record back[%d] on bstream\n", (void*)h, h->ov_sel);
    HIPCHK(h, hipEventRecord(h->ov_back[h->ov_sel], h->bstream));
    h->ov_back_rec[h->ov_sel] = true;
    h->ov_last = h->ov_sel;
    h->ov_pending = true;
    return SDR_OK;
}
// the front end of an overlapped batch writes the pair the back end before last read
static int ov_next_buffers(sdr_handle* h)
{
    std::swap(h->d_bits, h->d_bits2);
    std::swap(h->d_tsum, h->d_tsum2);
    h->ov_sel ^= 1;
    h->ov_front_covered = false;   // a front end is about to be queued on h->stream
    OV_TRACE("[ov %p] next buffers: pair %d%s\n", (void*)h, h->ov_sel, h->ov_back_rec[h->ov_sel] ? ", stream waits its back event" : "");
    if (h->ov_back_rec[h->ov_sel]) HIPCHK(h, hipStreamWaitEvent(h->stream, h->ov_back[h->ov_sel], 0));
    return SDR_OK;
}

// ------------------------------------------------------------------ firdes
// gr-filter/lib/firdes.cc low_pass + gr-fft/lib/window.cc (GNU Radio 3.7.11),
// restated: float windows, float taps, double normalisation.
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

static bool make_window(int win, int ntaps, double beta, std::vector<float>& w)
{
    w.resize(ntaps);
    const double M = ntaps - 1;
    for (int n = 0; n < ntaps; ++n) {
        switch (win) {
        case SDR_WIN_HAMMING: w[n] = 0.54 - 0.46 * cos((2 * M_PI * n) / M); break;
        case SDR_WIN_HANN: w[n] = 0.5 - 0.5 * cos((2 * M_PI * n) / M); break;
        case SDR_WIN_BLACKMAN:
            w[n] = 0.42 - 0.5 * cos((2 * M_PI * n) / M) + 0.08 * cos((4 * M_PI * n) / M);
            break;
        case SDR_WIN_RECTANGULAR: w[n] = 1; break;
        case SDR_WIN_KAISER: {
            double t = 2 * n * (1.0 / M) - 1;
            w[n] = izero(beta * sqrt(1.0 - t * t)) * (1.0 / izero(beta));
            break;
        }
        case SDR_WIN_BLACKMAN_HARRIS:
            w[n] = 0.35875 - 0.48829 * cos((2 * M_PI * n) / M) + 0.14128 * cos((4 * M_PI * n) / M) -
                   0.01168 * cos((6 * M_PI * n) / M);
            break;
        default: return false;
        }
    }
    return true;
}

extern "C" int sdr_firdes_low_pass(double gain, double fs, double fc, double tw, int win, double beta,
                                   float* taps, int cap)
{
    double a;
    switch (win) {
    case SDR_WIN_HAMMING: a = 53; break;
    case SDR_WIN_HANN: a = 44; break;
    case SDR_WIN_BLACKMAN: a = 74; break;
    case SDR_WIN_RECTANGULAR: a = 21; break;
    case SDR_WIN_KAISER: a = beta / 0.1102 + 8.7; break;
    case SDR_WIN_BLACKMAN_HARRIS: a = 92; break;
    default: return SDR_E_ARG;
    }
    if (!(fs > 0) || !(tw > 0) || !(fc > 0)) return SDR_E_ARG;
    int ntaps = (int)(a * fs / (22.0 * tw));
    if ((ntaps & 1) == 0) ntaps++;
    if (ntaps > cap || !taps) return SDR_E_CAPACITY;
    std::vector<float> w;
    make_window(win, ntaps, beta, w);
    const int M = (ntaps - 1) / 2;
    const double fwT0 = 2 * M_PI * fc / fs;
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
    return ntaps;
}

extern "C" int sdr_resampler_taps(int interp, int decim, double fbw, float* taps, int cap)
{
    if (interp < 1 || decim < 1) return SDR_E_ARG;
    int a = interp, b = decim;
    while (b) {
        int t = a % b;
        a = b;
        b = t;
    }
    interp /= a;
    decim /= a;
    if (fbw <= 0) fbw = 0.4;
    const double beta = 7.0, halfband = 0.5, rate = (double)interp / (double)decim;
    double tw, mid;
    if (rate >= 1.0) {
        tw = halfband - fbw;
        mid = halfband - tw / 2.0;
    } else {
        tw = rate * (halfband - fbw);
        mid = rate * halfband - tw / 2.0;
    }
    return sdr_firdes_low_pass(interp, interp, mid, tw, SDR_WIN_KAISER, beta, taps, cap);
}

// ------------------------------------------------------------------ tables
static void host_nco_table(std::vector<float2>& t)
{
    t.resize(SDR_NCO_TAB);
    for (int i = 0; i < SDR_NCO_TAB; ++i) {
        t[i].x = (float)cos(2 * M_PI * i / 1024.0);
        t[i].y = (float)sin(2 * M_PI * i / 1024.0);
    }
}

static void host_atan_table(std::vector<float>& t)
{
    // gnuradio fast_atan2f table: 7-significant-digit literals of atan(i/255)
    t.resize(258);
    char buf[64];
    for (int i = 0; i <= 255; ++i) {
        snprintf(buf, sizeof buf, "%.6e", atan((double)i / 255.0));
        t[i] = strtof(buf, nullptr);
    }
    t[256] = t[255];
    t[257] = t[255];
}

static void host_pow_tables(std::vector<double>& p, std::vector<double>& l)
{
    p.resize(SDR_MAX_POW_TABLE);
    l.resize(SDR_MAX_POW_TABLE);
    p[0] = 0;
    l[0] = 0;
    for (int n = 1; n < SDR_MAX_POW_TABLE; ++n) {
        p[n] = pow((double)n, -1.0 / 3.0);  // x.size ** (-1.0 / 3.0)
        l[n] = log2((double)n);             // np.log2(x.size)
    }
}

static unsigned long long host_nco_inc(double fc, double fs, int D)
{
    double f = -fc / fs * (double)D;
    f -= floor(f);
    if (f >= 1.0) f = 0.0;
    return (unsigned long long)ldexp(f, 64);
}

// ------------------------------------------------------------------ helpers
// SDR_POISON=1 (test switch): every device buffer starts as 0xA5 bytes instead of whatever hipMalloc hands out
// (zeros, in a fresh process), so that a kernel reading memory nothing has written shows up in the test suite;
// and every buffer gets a 512-byte guard band in front of it and one behind it that must still be 0xA5 when the
// buffer is freed (sdr_selftest_guard_hits() counts the buffers whose bands were not: writes outside a buffer).
#define SDR_GUARD_BYTES 512
static std::mutex g_guard_mu;
static std::unordered_map<void*, size_t> g_guard_size;
static int g_guard_hits = 0, g_guard_checked = 0;
static bool poison_on()
{
    static const bool on = getenv("SDR_POISON") != nullptr;
    return on;
}
static hipError_t dev_alloc(void** p, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (!poison_on()) return hipMalloc(p, bytes);
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, bytes + 2 * SDR_GUARD_BYTES);
    if (e != hipSuccess) return e;
    e = hipMemset(base, 0xA5, bytes + 2 * SDR_GUARD_BYTES);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    *p = static_cast<unsigned char*>(base) + SDR_GUARD_BYTES;   // a band in front (512 keeps hipMalloc's alignment) and one behind
    std::lock_guard<std::mutex> lk(g_guard_mu);
    g_guard_size[*p] = bytes;
    return e;
}
static hipError_t dev_free(void* p)
{
    if (!p || !poison_on()) return hipFree(p);
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(g_guard_mu);
        auto it = g_guard_size.find(p);
        if (it != g_guard_size.end()) {
            bytes = it->second;
            g_guard_size.erase(it);
        }
    }
    if (!bytes) return hipFree(p);   // not from dev_alloc
    unsigned char* const base = static_cast<unsigned char*>(p) - SDR_GUARD_BYTES;
    unsigned char g[2][SDR_GUARD_BYTES];
    (void)hipDeviceSynchronize();
    if (hipMemcpy(g[0], base, SDR_GUARD_BYTES, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(g[1], static_cast<unsigned char*>(p) + bytes, SDR_GUARD_BYTES, hipMemcpyDeviceToHost) == hipSuccess) {
        int bad[2] = {0, 0};
        for (int k = 0; k < 2; ++k)
            for (unsigned char b : g[k]) bad[k] += b != 0xA5;
        std::lock_guard<std::mutex> lk(g_guard_mu);
        g_guard_checked++;
        if (bad[0] || bad[1]) {
            g_guard_hits++;
            fprintf(stderr, "sdrhip: guard bytes around a %zu-byte device buffer were overwritten (%d in front, %d behind)\n", bytes, bad[0], bad[1]);
        }
    }
    return hipFree(base);
}

// hipMemset on device memory returns before the fill has run (it is queued on the null stream, which the
// handles' non-blocking streams do not wait for): a state reset must have landed before the next call queues
// work on the handle's stream, so wait for it.  (Seen as 1 in 8 wrong audio after sdr_fm_reset: the late fill
// wiped the history the first chunk had just written.)
static hipError_t memset_done(void* p, int v, size_t bytes)
{
    hipError_t e = hipMemset(p, v, bytes);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

static int upload_taps(sdr_handle* h)
{
    const int n = h->ntaps;
    const double fc = h->cfg.center_freq, fs = h->cfg.sample_rate;
    std::vector<float> rt;
    if (fc != 0.0 && !h->prerotate) {
        // freq_xlating_fir_filter: ctaps[k] = taps[k] * exp(j k 2 pi fc/fs)
        h->rt_complex = true;
        rt.resize(2 * n);
        const double th = 2 * M_PI * fc / fs;
        for (int k = 0; k < n; ++k) {
            float cr, ci;
            if (h->proto_complex) {
                // complex prototype: full complex product in double, rounded once
                double pr = h->proto_taps[2 * k], pi = h->proto_taps[2 * k + 1];
                cr = (float)(pr * cos(k * th) - pi * sin(k * th));
                ci = (float)(pr * sin(k * th) + pi * cos(k * th));
            } else {
                cr = (float)((double)h->proto_taps[k] * cos(k * th));
                ci = (float)((double)h->proto_taps[k] * sin(k * th));
            }
            rt[2 * (n - 1 - k)] = cr;
            rt[2 * (n - 1 - k) + 1] = ci;
        }
    } else if (h->proto_complex) {
        h->rt_complex = true;
        rt.resize(2 * n);
        for (int k = 0; k < n; ++k) {
            rt[2 * (n - 1 - k)] = h->proto_taps[2 * k];
            rt[2 * (n - 1 - k) + 1] = h->proto_taps[2 * k + 1];
        }
    } else {
        h->rt_complex = false;
        rt.resize(n);
        for (int k = 0; k < n; ++k) rt[n - 1 - k] = h->proto_taps[k];
    }
    memset(&h->ktaps, 0, sizeof h->ktaps);
    h->ktaps_n = 0;
    if (n <= SDR_KTAPS && !h->no_ktaps) {
        for (int k = 0; k < n; ++k) {
            h->ktaps.t[k].x = h->rt_complex ? rt[2 * k] : rt[k];
            h->ktaps.t[k].y = h->rt_complex ? rt[2 * k + 1] : 0.0f;
        }
        h->ktaps_n = n;
    }
    if (h->d_rtaps) (void)dev_free(h->d_rtaps);
    h->d_rtaps = nullptr;
    HIPCHK(h, dev_alloc((void**)&h->d_rtaps, rt.size() * sizeof(float)));
    HIPCHK(h, hipMemcpy(h->d_rtaps, rt.data(), rt.size() * sizeof(float), hipMemcpyHostToDevice));
    h->nco_inc = (fc != 0.0) ? host_nco_inc(fc, fs, h->prerotate ? 1 : h->cfg.decim) : 0;
    // streaming history: ntaps-1 + 2*D samples (previous output for QUAD, GR's
    // "emit once D items are consumed" rule)
    h->hist_len = n - 1 + 2 * h->cfg.decim;
    if (h->hist_len > h->hist_cap) {
        for (int i = 0; i < 2; ++i) {
            if (h->d_hist[i]) (void)dev_free(h->d_hist[i]);
            h->d_hist[i] = nullptr;
            HIPCHK(h, dev_alloc((void**)&h->d_hist[i], (size_t)h->hist_len * sizeof(float2)));
        }
        h->hist_cap = h->hist_len;
    }
    for (int i = 0; i < 2; ++i)
        HIPCHK(h, memset_done(h->d_hist[i], 0, (size_t)h->hist_cap * sizeof(float2)));
    return SDR_OK;
}

static double decoder_reset_limit(int dec)
{
    return dec == SDR_DEC_TS33C ? 1550.0 : 500.0;
}

static bool is_stream_decoder(int dec) { return dec >= SDR_DEC_TS33C && dec <= SDR_DEC_TFA_GRC; }

// launch geometry of the front end: threads (= FIR outputs) per tile
static void pick_tile(const sdr_handle* h, int& T, size_t& lds, bool& use_lds)
{
    const int D = h->cfg.decim, nt = h->ntaps;
    const int cands[3] = {256, 128, 64};
    use_lds = true;
    // +2: 16-B aligned tile start
    auto bytes = [&](int t) { return ((size_t)(t - 1) * D + nt + 2) * sizeof(float2); };
    if (h->tile_override) {  // tuning override: 64 | 128 | 256
        T = h->tile_override;
        lds = bytes(T);
        if ((T == 64 || T == 128 || T == 256) && lds <= 60 * 1024) return;
    }
    for (int i = 0; i < 3; ++i) {
        T = cands[i];
        lds = bytes(T);
        if (lds <= 52 * 1024) return;  // 3 two-wave workgroups per CU measured 4 % faster than 6 one-wave ones
    }
    if (lds <= 60 * 1024) return;
    use_lds = false;
    T = 256;
    lds = 0;
}

// k_frontend_mfma (frontend_mfma.h): LDS bytes for nw wavefronts, 0 if it does not fit
static size_t mfma_lds_bytes(int D, int ntaps, int nw)
{
    const long long TB = 256LL * nw, BK = 16LL * D;
    const long long ksteps = (15LL * D + ntaps + 3) >> 2;
    const long long ntl = (15LL * D + 4 * (ksteps + 8) + 3) & ~3LL;
    const long long npos = (TB - 1) * D + ntaps + 40 + 2;
    const long long smp = 2 * npos + 4 * (npos / BK + 2);
    const long long bytes = (ntl + smp) * 4;
    return bytes <= 60 * 1024 ? (size_t)bytes : 0;
}

// The matrix-core kernel pays when the filter is fp32-FMA bound: real taps and
// enough flops per input sample that the LDS-walking kernel is not HBM bound.
// k_frontend_win (frontend_win.h): decimation 1, real or complex (xlating) taps, complex64 or cu8 input.  Every lane owns WIN_R
// consecutive outputs; chosen over the matrix-core form there (measured).
static bool pick_win(const sdr_handle* h, long long nout, int& T, size_t& lds)
{
    if (h->front_mode == 1 || h->front_mode == 2 || h->cfg.decim != 1 || h->prerotate) return false;
    if (h->front_mode == 0 && h->ntaps < 8) return false;
    T = nout >= 8LL * 256 * WIN_R ? 256 : 64;
    if (h->tile_override == 64 || h->tile_override == 128 || h->tile_override == 256) T = h->tile_override;
    lds = (size_t)win_tap_floats(h->ntaps, h->rt_complex) * 4 + (size_t)win_sample_slots(T, h->ntaps) * sizeof(float2);
    return lds <= 60 * 1024;
}

static bool pick_mfma(const sdr_handle* h, int& T, size_t& lds)
{
    if (h->front_mode == 1 || h->front_mode == 3 || h->rt_complex || h->in_fmt != SDR_IN_FC32) return false;
    const int D = h->cfg.decim, nt = h->ntaps;
    if (h->front_mode == 0 && 4.0 * nt / D < 24.0) return false;
    const int nw0 = (h->tile_override == 64 || h->tile_override == 128) ? h->tile_override / 64 : 4;   // tuning override
    for (int nw = nw0; nw >= 1; nw >>= 1) {
        size_t b = mfma_lds_bytes(D, nt, nw);
        if (b) {
            T = 64 * nw;
            lds = b;
            return true;
        }
    }
    return false;
}

static int pfb_upload_taps(sdr_handle* h);
static int scratch_init(sdr_handle* h);

// Threshold-only discriminator (frontend_win.h): the arctangent values at which fl(gain * a) crosses the
// thresholds, found by walking floats -- a -> fl(gain*a) is monotone for gain > 0, so "gain*a > hi" is
// "a >= a_hi" for the smallest float a_hi whose product exceeds hi (and likewise below lo).
static void quad_fast_setup(sdr_handle* h)
{
    h->quad_fast = false;
    const sdr_chain_cfg& c = h->cfg;
    if (c.demod != SDR_DEMOD_QUAD || c.post_ntaps > 0 || getenv("SDR_NO_QUADFAST")) return;
    const float g = c.demod_gain, lo = c.thr_lo, hi = c.thr_hi;
    if (!(g > 0.0f) || !std::isfinite(g) || !std::isfinite(lo) || !std::isfinite(hi) || !(lo <= hi)) return;
    auto above = [&](float a) { return g * a > hi; };
    auto below = [&](float a) { return g * a < lo; };
    float a_hi = (float)((double)hi / (double)g), a_lo = (float)((double)lo / (double)g);
    for (int i = 0; i < 64 && above(a_hi); ++i) a_hi = std::nextafterf(a_hi, -INFINITY);   // largest float NOT above ...
    for (int i = 0; i < 64 && !above(a_hi); ++i) a_hi = std::nextafterf(a_hi, INFINITY);   // ... then the smallest above
    for (int i = 0; i < 64 && below(a_lo); ++i) a_lo = std::nextafterf(a_lo, INFINITY);
    for (int i = 0; i < 64 && !below(a_lo); ++i) a_lo = std::nextafterf(a_lo, -INFINITY);
    if (!above(a_hi) || above(std::nextafterf(a_hi, -INFINITY)) || !below(a_lo) || below(std::nextafterf(a_lo, INFINITY))) return;
    const double guard = 1e-4, pi = M_PI;
    // both directions well inside (-pi, pi): beyond it a threshold is never / always met and the plain form is kept
    if (!(std::fabs((double)a_hi) < pi - 0.01) || !(std::fabs((double)a_lo) < pi - 0.01)) return;
    h->q_c_hi = (float)std::cos((double)a_hi);
    h->q_s_hi = (float)std::sin((double)a_hi);
    h->q_c_lo = (float)std::cos((double)a_lo);
    h->q_s_lo = (float)std::sin((double)a_lo);
    // a direction within 1e-6 of the real axis would make the "upper half plane" choice of its side rounding-dependent
    if (std::fabs(h->q_s_hi) < 1e-6f && a_hi != 0.0f) return;
    if (std::fabs(h->q_s_lo) < 1e-6f && a_lo != 0.0f) return;
    h->q_band = (float)guard;
    h->quad_fast = true;
}

// ------------------------------------------------------------------ create
extern "C" int sdr_version(void) { return 101; }

// stream 0's decoder object as StreamDecoder.__init__ leaves it, threshold_ff at its init value
static int reset_dec_state(sdr_handle* h)
{
    DecState s0;
    memset(&s0, 0, sizeof s0);
    s0.thr_state = h->cfg.thr_init;
    s0.first_evt = -1;
    s0.pulse_short_limit = 775.0;
    s0.gap_short_limit = 775.0;
    // a plain chain streams through row 0 only; a channelizer through its 64 channel rows
    const size_t rows = h->pfb_M ? (size_t)h->cfg.max_batch : 1;
    std::vector<DecState> all(rows, s0);
    HIPCHK(h, hipMemcpy(h->d_st, all.data(), rows * sizeof s0, hipMemcpyHostToDevice));
    HIPCHK(h, memset_done(h->d_rule, 0, h->rule_rows * sizeof(RuleState)));
    h->carry_next = h->cfg.thr_init;
    h->carry_last = -1;
    return SDR_OK;
}

extern "C" const char* sdr_last_error(sdr_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

extern "C" int sdr_destroy(sdr_handle* h)
{
    if (!h) return SDR_OK;
    (void)hipSetDevice(h->device);
    if (h->pre) {
        (void)hipStreamSynchronize(h->stream);
        sdr_destroy(h->pre);
        h->pre = nullptr;
    }
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    if (h->bstream) {
        (void)hipStreamSynchronize(h->bstream);
        (void)hipStreamSynchronize(h->stream);
        (void)hipStreamDestroy(h->bstream);
        for (hipEvent_t e : {h->ov_front, h->ov_back[0], h->ov_back[1]})
            if (e) (void)hipEventDestroy(e);
    }
    void* ptrs[] = {h->d_bits2, h->d_tsum2, h->d_rtaps, h->d_post_rtaps, h->d_nco_tab, h->d_atan_tab, h->d_pow, h->d_log2, h->d_iq,
                    h->d_hist[0], h->d_hist[1], h->d_bits, h->d_tap, h->d_tap2, h->d_tsum, h->d_runs, h->d_runs_seg, h->d_seg_cnt, h->d_runs_n, h->d_pieces_hdr,
                    h->d_cur, h->d_st, h->d_rule, h->d_frames, h->d_m_cur, h->d_m_st, h->d_m_frames, h->d_sq_y, h->d_sq_z, h->d_sq_cnt, h->d_sq_st,
                    h->d_post_hist[0], h->d_post_hist[1], h->d_pack, h->d_pfb_taps, h->d_pfb_tw};
    for (void* p : ptrs)
        if (p) (void)dev_free(p);
    for (hipEvent_t e : h->pev) (void)hipEventDestroy(e);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return SDR_OK;
}

extern "C" int sdr_create(const sdr_chain_cfg* cfg, sdr_handle** out)
{
    if (!cfg || !out) return fail(nullptr, SDR_E_ARG, "sdr_create: null argument");
    *out = nullptr;
    if (cfg->decim < 1 || cfg->keep_n < 1 || cfg->max_batch < 1 || cfg->max_samples < 1 ||
        !(cfg->sample_rate > 0) || cfg->demod < SDR_DEMOD_NONE || cfg->demod > SDR_DEMOD_QUAD ||
        cfg->decoder < SDR_DEC_NONE || cfg->decoder > SDR_DEC_VELUX || cfg->post_ntaps < 0)
        return fail(nullptr, SDR_E_ARG, "sdr_create: bad configuration");
    if (cfg->max_samples / (uint64_t)cfg->decim >= 0x7fffffffull - 1024ull)
        return fail(nullptr, SDR_E_CAPACITY, "sdr_create: max_samples/decim must stay below 2^31 filter outputs per capture or call "
                                             "(tile summaries and run tables index them with 32 bits); cut the capture into segments (sdr_seek)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, SDR_E_NODEVICE, "sdr_create: no HIP device (this library has no CPU path)");
    sdr_handle* h = new sdr_handle();
    h->cfg = *cfg;
    if (h->cfg.max_frames < 1) h->cfg.max_frames = 16;
    if (hipGetDevice(&h->device) != hipSuccess) {
        delete h;
        return fail(nullptr, SDR_E_HIP, "hipGetDevice failed");
    }
    if (const char* fm = getenv("SDR_FRONTEND"))
        h->front_mode = !strcmp(fm, "valu") ? 1 : (!strcmp(fm, "mfma") ? 2 : (!strcmp(fm, "win") ? 3 : 0));
    if (const char* pe = getenv("SDR_PIPE")) h->pipe_mode = atoi(pe);
    if (const char* pe = getenv("SDR_TPW")) h->pipe_tpw = atoi(pe) > 0 ? atoi(pe) : 1;
    h->no_ktaps = getenv("SDR_NO_KTAPS") != nullptr;
    h->no_quad_inst = getenv("SDR_NO_QUAD_INST") != nullptr;
    h->no_raw8 = getenv("SDR_NO_RAW8") != nullptr;
    h->no_reuse = getenv("SDR_NO_REUSE") != nullptr;
    h->no_seg = getenv("SDR_NO_SEG") != nullptr;
    h->no_pieces = getenv("SDR_NO_PIECES") != nullptr;
    if (const char* e = getenv("SDR_MFMA2_TPW")) h->mfma2_tpw = atoi(e);
    if (!getenv("SDR_NO_MFMA2"))   // two sample buffers of a 1024-output tile: up to 88 KiB of dynamic LDS, above the 64 KiB default
        h->mfma2_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_frontend_mfma2), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          88 * 1024) == hipSuccess;
    if (const char* te = getenv("SDR_TILE")) h->tile_override = atoi(te);
    if (const char* te = getenv("SDR_SEG_MIN")) h->seg_min_row = atoll(te) >= 16384 ? atoll(te) : 16384;
    (void)hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, h->device);
    if (h->num_cus < 1) h->num_cus = 256;
    int rc = SDR_OK;
    auto body = [&]() -> int {
        HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
        h->stream = h->own_stream;
        HIPCHK(h, hipEventCreate(&h->ev0));
        HIPCHK(h, hipEventCreate(&h->ev1));
        // tables
        std::vector<float2> nco;
        host_nco_table(nco);
        HIPCHK(h, dev_alloc((void**)&h->d_nco_tab, nco.size() * sizeof(float2)));
        HIPCHK(h, hipMemcpy(h->d_nco_tab, nco.data(), nco.size() * sizeof(float2), hipMemcpyHostToDevice));
        std::vector<float> at;
        host_atan_table(at);
        HIPCHK(h, dev_alloc((void**)&h->d_atan_tab, at.size() * sizeof(float)));
        HIPCHK(h, hipMemcpy(h->d_atan_tab, at.data(), at.size() * sizeof(float), hipMemcpyHostToDevice));
        std::vector<double> pw, lg;
        host_pow_tables(pw, lg);
        HIPCHK(h, dev_alloc((void**)&h->d_pow, pw.size() * sizeof(double)));
        HIPCHK(h, dev_alloc((void**)&h->d_log2, lg.size() * sizeof(double)));
        HIPCHK(h, hipMemcpy(h->d_pow, pw.data(), pw.size() * sizeof(double), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->d_log2, lg.data(), lg.size() * sizeof(double), hipMemcpyHostToDevice));
        // default taps = [1]
        h->proto_taps = {1.0f};
        h->proto_complex = false;
        h->ntaps = 1;
        int r = upload_taps(h);
        if (r) return r;
        // buffers
        const long long B = h->cfg.max_batch;
        const long long nout_max = ((long long)(h->cfg.max_samples / h->cfg.decim) + 3) & ~1LL;  // even: rows stay 16-B aligned
        const long long nb_max = nout_max / h->cfg.keep_n + 2;
        h->dec_rate = h->cfg.decoder_rate > 0 ? h->cfg.decoder_rate
                                              : h->cfg.sample_rate / h->cfg.decim / h->cfg.keep_n;
        // rawBuffer may hold up to reset_limit worth of pending samples plus one call
        long long pend = (long long)(decoder_reset_limit(h->cfg.decoder) * 1e-6 * h->dec_rate) + 64;
        // batches decode every capture from a fresh rawBuffer; 64 rows may be a one-capture channelizer handle,
        // which streams (pfb_stream accumulates a stream decoder's rawBuffer in every channel row)
        if (B > 1 && B != PFB_M) pend = 0;
        h->bits_stride = ((nb_max + pend + 63) / 64) * 64;
        HIPCHK(h, dev_alloc((void**)&h->d_bits, (size_t)(B * h->bits_stride)));
        HIPCHK(h, memset_done(h->d_bits, 0, (size_t)(B * h->bits_stride)));
        if (h->cfg.tap_demod || h->cfg.post_ntaps > 0) {
            h->tap_stride = nout_max * (h->cfg.demod == SDR_DEMOD_NONE ? 2 : 1);
            HIPCHK(h, dev_alloc((void**)&h->d_tap, (size_t)(B * h->tap_stride) * sizeof(float)));
            if (h->cfg.post_ntaps > 0 && h->cfg.tap_demod)
                HIPCHK(h, dev_alloc((void**)&h->d_tap2, (size_t)(B * h->tap_stride) * sizeof(float)));
        }
        h->tsum_stride = (int)(nout_max / 63 + 2);  // smallest tile emits 63 outputs (QUAD, T=64)
        HIPCHK(h, dev_alloc((void**)&h->d_tsum, (size_t)(B * h->tsum_stride) * sizeof(TileSum)));
        h->runs_cap = h->bits_stride;
        HIPCHK(h, dev_alloc((void**)&h->d_runs, (size_t)(B * (h->runs_cap + 2)) * sizeof(int32_t)));
        HIPCHK(h, dev_alloc((void**)&h->d_runs_n, (size_t)B * sizeof(int32_t)));
        if (h->bits_stride >= h->seg_min_row) {
            HIPCHK(h, dev_alloc((void**)&h->d_runs_seg, (size_t)(B * h->bits_stride) * sizeof(int32_t)));
            HIPCHK(h, dev_alloc((void**)&h->d_seg_cnt, (size_t)(B * BK_THREADS) * sizeof(int32_t)));
            HIPCHK(h, dev_alloc((void**)&h->d_pieces_hdr, (size_t)(B * 8) * sizeof(int32_t)));
        }
        HIPCHK(h, dev_alloc((void**)&h->d_cur, (size_t)(B * SDR_CUR_CAP) * sizeof(double)));
        HIPCHK(h, dev_alloc((void**)&h->d_st, (size_t)B * sizeof(DecState)));
        HIPCHK(h, memset_done(h->d_st, 0, (size_t)B * sizeof(DecState)));
        h->rule_rows = (B == 64) ? 64 : 1;   // 64: a one-capture channelizer handle may stream (sdr_pfb_create)
        HIPCHK(h, dev_alloc((void**)&h->d_rule, h->rule_rows * sizeof(RuleState)));
        {
            int r2 = reset_dec_state(h);
            if (r2) return r2;
        }
        HIPCHK(h, dev_alloc((void**)&h->d_frames, (size_t)(B * h->cfg.max_frames) * sizeof(sdr_frame)));
        // the record has 4 bytes of alignment padding no field assignment reaches: keep them defined
        HIPCHK(h, memset_done(h->d_frames, 0, (size_t)(B * h->cfg.max_frames) * sizeof(sdr_frame)));
        h->keep_ctr = h->cfg.keep_n;
        quad_fast_setup(h);
        return scratch_init(h);
    };
    rc = body();
    if (rc != SDR_OK) {
        g_err = h->err;
        sdr_destroy(h);
        return rc;
    }
    *out = h;
    return SDR_OK;
}

static void reset_stream_counters(sdr_handle* h)
{
    h->in_count = h->out_count = 0;
    h->keep_ctr = h->cfg.keep_n;
    h->raw_n = 0;
    h->last_ncap = 0;
    h->last_nb = h->last_nout = h->last_bits_off = 0;
}

extern "C" int sdr_set_taps(sdr_handle* h, const float* taps, int n, int is_complex)
{
    if (!h || !taps || n < 1 || n > 65536) return fail(h, SDR_E_ARG, "sdr_set_taps: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->proto_taps.assign(taps, taps + (is_complex ? 2 * n : n));
    h->proto_complex = is_complex != 0;
    h->ntaps = n;
    if (h->pfb_M) return pfb_upload_taps(h);
    // new taps restart the stream (history zeroed by upload_taps, counters and decoder state here):
    // a history of zero BYTES would not be the zeros before stream start for cu8 input
    int r = upload_taps(h);
    if (r) return r;
    reset_stream_counters(h);
    return reset_dec_state(h);
}

extern "C" int sdr_set_post_taps(sdr_handle* h, const float* taps, int n)
{
    if (!h || !taps || n < 1 || n != h->cfg.post_ntaps)
        return fail(h, SDR_E_ARG, "sdr_set_post_taps: n must equal cfg.post_ntaps");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    std::vector<float> rt(n);
    for (int k = 0; k < n; ++k) rt[n - 1 - k] = taps[k];
    if (!h->d_post_rtaps) HIPCHK(h, dev_alloc((void**)&h->d_post_rtaps, n * sizeof(float)));
    for (int i = 0; i < 2; ++i) {
        if (!h->d_post_hist[i]) HIPCHK(h, dev_alloc((void**)&h->d_post_hist[i], (size_t)(n + 1) * sizeof(float)));
        HIPCHK(h, memset_done(h->d_post_hist[i], 0, (size_t)(n + 1) * sizeof(float)));
    }
    HIPCHK(h, hipMemcpy(h->d_post_rtaps, rt.data(), n * sizeof(float), hipMemcpyHostToDevice));
    h->post_taps.assign(taps, taps + n);
    reset_stream_counters(h);   // as sdr_set_taps: the stream restarts
    return reset_dec_state(h);
}

extern "C" int sdr_set_prefilter(sdr_handle* h, const float* taps, int n, int decim)
{
    if (!h || !taps || n < 1 || n > 65536 || decim < 1) return fail(h, SDR_E_ARG, "sdr_set_prefilter: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->pre) {
        sdr_destroy(h->pre);
        h->pre = nullptr;
    }
    sdr_chain_cfg pc;
    memset(&pc, 0, sizeof pc);
    pc.sample_rate = h->cfg.sample_rate * decim;
    pc.decim = decim;
    pc.demod = SDR_DEMOD_NONE;
    pc.demod_gain = 1.0f;
    pc.keep_n = 1;
    pc.decoder = SDR_DEC_NONE;
    pc.tap_demod = 1;
    pc.max_batch = h->cfg.max_batch;
    pc.max_frames = 1;
    pc.max_samples = h->cfg.max_samples * (uint64_t)decim;
    sdr_handle* p = nullptr;
    int r = sdr_create(&pc, &p);
    if (r) return fail(h, r, "sdr_set_prefilter: " + g_err);
    r = sdr_set_taps(p, taps, n, 0);
    if (r) {
        std::string e = p->err;
        sdr_destroy(p);
        return fail(h, r, "sdr_set_prefilter: " + e);
    }
    p->stream = h->stream;
    p->in_fmt = h->in_fmt;  // the upstream stage reads the caller's samples
    h->in_fmt = SDR_IN_FC32;
    h->pre = p;
    return SDR_OK;
}

extern "C" int sdr_set_input_format(sdr_handle* h, int fmt)
{
    if (!h || (fmt != SDR_IN_FC32 && fmt != SDR_IN_CU8)) return fail(h, SDR_E_ARG, "sdr_set_input_format: unknown format");
    if (h->prerotate) return fail(h, SDR_E_ARG, "sdr_set_input_format: not available on this chain");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    sdr_handle* t = h->pre ? h->pre : h;
    t->in_fmt = fmt;
    reset_stream_counters(t);
    if (t != h) reset_stream_counters(h);
    return reset_dec_state(h);
}

extern "C" int sdr_set_squelch(sdr_handle* h, double threshold_db, double alpha, int gate)
{
    if (!h || !(alpha > 0.0) || alpha > 1.0) return fail(h, SDR_E_ARG, "sdr_set_squelch: alpha must be in (0, 1]");
    if (h->cfg.post_ntaps > 0 || h->cfg.demod == SDR_DEMOD_NONE || h->prerotate || h->skip_slice)
        return fail(h, SDR_E_ARG, "sdr_set_squelch: needs a MAG2/QUAD chain without a post-demod filter");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const long long B = h->cfg.max_batch;
    if (!h->d_sq_y) {
        h->sq_stride = (long long)(h->cfg.max_samples / h->cfg.decim) + 4;
        HIPCHK(h, dev_alloc((void**)&h->d_sq_y, (size_t)(B * h->sq_stride) * sizeof(float2)));
        HIPCHK(h, dev_alloc((void**)&h->d_sq_z, (size_t)(B * h->sq_stride) * sizeof(float2)));
        HIPCHK(h, dev_alloc((void**)&h->d_sq_cnt, (size_t)B * sizeof(int32_t)));
        HIPCHK(h, dev_alloc((void**)&h->d_sq_st, (size_t)B * sizeof(SquelchState)));
    }
    HIPCHK(h, memset_done(h->d_sq_st, 0, (size_t)B * sizeof(SquelchState)));
    HIPCHK(h, memset_done(h->d_sq_cnt, 0, (size_t)B * sizeof(int32_t)));
    h->sq_alpha = alpha;
    h->sq_thr = std::pow(10.0, threshold_db / 10.0);
    h->sq_gate = gate ? 1 : 0;
    h->squelch_on = true;
    return SDR_OK;
}

extern "C" int sdr_set_stream(sdr_handle* h, void* s)
{
    if (!h) return SDR_E_ARG;
    h->stream = s ? (hipStream_t)s : h->own_stream;
    h->ov_front_covered = false;
    if (h->pre) h->pre->stream = h->stream;
    return SDR_OK;
}

extern "C" int sdr_synchronize(sdr_handle* h)
{
    if (!h) return SDR_E_ARG;
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SDR_OK;
}

extern "C" int sdr_set_overlap(sdr_handle* h, int on)
{
    if (!h) return SDR_E_ARG;
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (on && !h->bstream) {
        const size_t B = (size_t)h->cfg.max_batch;
        {
            // The result stream has the SAME priority as the handle's stream.  Round 2 created it at the highest
            // priority (a hardware queue of its own, never the one h->stream sits on); round 3 found that on this
            // platform a high-priority stream deadlocks the process once HIP's GPU_MAX_HW_QUEUES (4) normal-priority
            // hardware queues are all open: the fifth, high-priority queue sits in its barrier packet waiting for an
            // event of a normal-priority queue that is then never scheduled (tools/probe/two_queue_pmc.hip shows it
            // without any library code; the same wait pattern at one priority runs).  At one priority HIP hands its
            // hardware queues out in turn, so a new stream may land on the queue h->stream uses -- correct (one queue
            // is in submission order) but nothing overlaps.  So candidates are probed: a candidate whose tiny kernel
            // finishes while a ~2 ms one-wave spin kernel still occupies h->stream's queue sits on another queue.
            int r = make_side_stream(h, h->stream, &h->bstream);
            if (r) return r;
        }
        HIPCHK(h, hipEventCreateWithFlags(&h->ov_front, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) HIPCHK(h, hipEventCreateWithFlags(&h->ov_back[i], hipEventDisableTiming));
        HIPCHK(h, dev_alloc((void**)&h->d_bits2, B * (size_t)h->bits_stride));
        HIPCHK(h, memset_done(h->d_bits2, 0, B * (size_t)h->bits_stride));
        HIPCHK(h, dev_alloc((void**)&h->d_tsum2, B * (size_t)h->tsum_stride * sizeof(TileSum)));
    }
    h->overlap = on != 0;
    return SDR_OK;
}

// Diagnostics: where the two streams of an overlapped handle stand, without blocking (hipStreamQuery /
// hipEventQuery).  out[0] = handle stream idle, out[1] = result stream idle, out[2] = ov_front reached,
// out[3], out[4] = ov_back[0], ov_back[1] reached (1 yes, 0 not yet, -1 never recorded / no second stream),
// out[5] = pair the last batch wrote, out[6] = pair whose event was recorded last, out[7] = result stream holds
// work the handle's stream has not been ordered behind.
extern "C" int sdr_overlap_state(sdr_handle* h, int32_t out[8])
{
    if (!h || !out) return SDR_E_ARG;
    (void)hipSetDevice(h->device);
    auto q = [](hipError_t e) { return e == hipSuccess ? 1 : (e == hipErrorNotReady ? 0 : -2); };
    out[0] = q(hipStreamQuery(h->stream));
    out[1] = h->bstream ? q(hipStreamQuery(h->bstream)) : -1;
    out[2] = h->ov_front ? q(hipEventQuery(h->ov_front)) : -1;
    for (int i = 0; i < 2; ++i) out[3 + i] = (h->ov_back[i] && h->ov_back_rec[i]) ? q(hipEventQuery(h->ov_back[i])) : -1;
    out[5] = h->ov_sel;
    out[6] = h->ov_last;
    out[7] = h->ov_pending ? 1 : 0;
    (void)hipGetLastError();   // hipErrorNotReady is an answer, not a failure
    return SDR_OK;
}

extern "C" void* sdr_result_stream(sdr_handle* h)
{
    if (!h) return nullptr;
    return (void*)(h->overlap ? h->bstream : h->stream);
}

extern "C" int sdr_timer_start(sdr_handle* h)
{
    if (!h) return SDR_E_ARG;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    return SDR_OK;
}
extern "C" int sdr_timer_stop(sdr_handle* h)
{
    if (!h) return SDR_E_ARG;
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    return SDR_OK;
}
extern "C" int sdr_timer_elapsed_ms(sdr_handle* h, float* ms)
{
    if (!h || !ms) return SDR_E_ARG;
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return SDR_OK;
}

// ------------------------------------------------------------------ streaming history
// nxt[i] = (cur ++ x)[n + i], i in [0, H): what a block remembers for its next call -- the last H units of its
// history followed by the n new ones -- in one small launch on the call's stream (it was one or two
// hipMemcpyAsync device-to-device copies per memory: each a command of its own with a copy engine's hand-over
// around it; the FM chain updates four memories per call).  Units of 2 bytes: cu8 samples, floats, complex64.
__global__ void k_tail_update(const unsigned short* __restrict__ cur, const unsigned short* __restrict__ x, long long H,
                              long long n, unsigned short* __restrict__ nxt)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < H; i += (long long)gridDim.x * blockDim.x) {
        const long long j = i + n;
        nxt[i] = j < H ? cur[j] : x[j - H];
    }
}
// Several memories in one launch (blockIdx.y picks one): a flowgraph object that runs several blocks per call
// (sdr_fm_process) collects its blocks' updates here and flushes them once, at the end of the call -- every
// source stays intact until then.
struct TailJob {
    const unsigned short *cur, *x;
    long long H, n;
    unsigned short* nxt;
};
struct TailBatch {
    TailJob job[6];
    int count;
};
__global__ void k_tail_update_batch(TailBatch B)
{
    const TailJob J = B.job[blockIdx.y];
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < J.H; i += (long long)gridDim.x * blockDim.x) {
        const long long j = i + J.n;
        J.nxt[i] = j < J.H ? J.cur[j] : J.x[j - J.H];
    }
}
static hipError_t tail_flush(hipStream_t stream, TailBatch* tb)
{
    if (!tb || tb->count == 0) return hipSuccess;
    long long hmax = 0;
    for (int i = 0; i < tb->count; ++i) hmax = tb->job[i].H > hmax ? tb->job[i].H : hmax;
    long long blocks = (hmax + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_tail_update_batch, dim3((unsigned)blocks, (unsigned)tb->count), dim3(256), 0, stream, *tb);
    tb->count = 0;
    return hipGetLastError();
}
static hipError_t tail_update(hipStream_t stream, const void* cur, const void* x, size_t h_bytes, size_t n_bytes, void* nxt,
                              TailBatch* defer = nullptr)
{
    if (h_bytes == 0 || n_bytes == 0) return hipSuccess;
    if (defer && defer->count < 6) {
        defer->job[defer->count++] = TailJob{(const unsigned short*)cur, (const unsigned short*)x, (long long)(h_bytes / 2),
                                             (long long)(n_bytes / 2), (unsigned short*)nxt};
        return hipSuccess;
    }
    const long long H = (long long)(h_bytes / 2), n = (long long)(n_bytes / 2);
    long long blocks = (H + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_tail_update, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned short*)cur,
                       (const unsigned short*)x, H, n, (unsigned short*)nxt);
    return hipGetLastError();
}

// ------------------------------------------------------------------ launches
static int ensure_iq_staging(sdr_handle* h, size_t samples)
{
    if (samples <= h->iq_cap) return SDR_OK;
    if (h->d_iq) (void)dev_free(h->d_iq);
    h->d_iq = nullptr;
    h->iq_cap = 0;
    HIPCHK(h, dev_alloc((void**)&h->d_iq, samples * sizeof(float2)));
    h->iq_cap = samples;
    return SDR_OK;
}

struct CallGeom {
    long long nout;     // FIR/demod outputs per capture
    long long in0;
    long long keep_c0;
    long long ne;       // bytes emitted per capture
    int T, tile_out, tiles;
    size_t lds;
    bool use_lds;
    bool mfma;   // k_frontend_mfma: T = 64*nw threads, 256*nw FIR outputs per tile
    bool win = false;  // k_frontend_win: T threads, T*WIN_R FIR outputs per tile
    int mfma2 = 0;     // >0: k_frontend_mfma2 (two teams, no demodulation), this many tiles per workgroup
    size_t mfma2_lds = 0;
    int pipe_nch = 0;  // >0: k_frontend_pipe with this many 16-B chunks per lane per tile
    int pipe_tpw = 0;  // tiles per workgroup
};

static void fill_geom_gated(const sdr_handle* h, long long nitems, long long keep_c0, CallGeom& g);
static int pfb_batch(sdr_handle* h, const void* iq, size_t n, size_t stride, int ncap, int is_dev, sdr_out* out);
static int pfb_stream(sdr_handle* h, const void* iq, size_t n, int is_dev, sdr_out* out);
static int launch_back(sdr_handle* h, int ncap, const CallGeom& g, long long patch_off, long long nb_total, int fresh,
                       int use_state_thr, int run_decoder);

// channel filter (+ demod + slicer unless the output goes to the squelch stage)
static int launch_front(sdr_handle* h, const float2* x, long long stride, long long n, int ncap,
                        const float2* hist, int hist_len, const CallGeom& g, long long patch_off, bool to_squelch)
{
    const sdr_chain_cfg& c = h->cfg;
    const bool post = c.post_ntaps > 0;
    FrontParams P;
    memset(&P, 0, sizeof P);
    P.x = x;
    P.hist = hist;
    P.stride = stride;
    P.n = n;
    P.nout = g.nout;
    P.in0 = g.in0;
    P.hist_len = hist_len;
    P.in_fmt = h->in_fmt;
    P.vec_ok = ((((uintptr_t)x) & 15) == 0 && (stride % (h->in_fmt == SDR_IN_CU8 ? 8 : 2)) == 0) ? 1 : 0;
    P.D = c.decim;
    P.ntaps = h->ntaps;
    P.taps_complex = h->rt_complex;
    P.rtaps = h->d_rtaps;
    P.ktaps_n = h->ktaps_n;
    P.nco_inc = h->nco_inc;
    P.nco_m0 = h->out_count;
    P.nco_tab = (c.center_freq != 0.0 && !h->prerotate) ? h->d_nco_tab : nullptr;
    if (c.center_freq != 0.0 && h->prerotate) {
        P.pre_tab = h->d_nco_tab;
        P.pre_inc = h->nco_inc;
        P.pre_idx0 = h->in_count;
    }
    P.atan_tab = h->d_atan_tab;
    P.demod = c.demod;
    P.gain = post ? 1.0f : c.demod_gain;
    P.lo = c.thr_lo;
    P.hi = c.thr_hi;
    P.do_slice = (c.demod != SDR_DEMOD_NONE && !post && !h->skip_slice) ? 1 : 0;
    P.keep_n = c.keep_n;
    P.keep_c0 = g.keep_c0;
    P.bits = h->d_bits + patch_off;
    P.bits_stride = h->bits_stride;
    P.tap = (c.tap_demod || post || c.demod == SDR_DEMOD_NONE) ? h->d_tap : nullptr;
    P.tap_stride = h->tap_stride;
    P.tsum = h->d_tsum;
    P.tiles_per_capture = g.tiles;
    P.tile_out = g.tile_out;
    if (h->quad_fast && P.do_slice && !P.tap && !to_squelch && P.gain == c.demod_gain) {
        P.quad_fast = 1;
        P.q_c_hi = h->q_c_hi;
        P.q_s_hi = h->q_s_hi;
        P.q_c_lo = h->q_c_lo;
        P.q_s_lo = h->q_s_lo;
        P.q_band = h->q_band;
    }
    if (to_squelch) {
        P.demod = SDR_DEMOD_NONE;
        P.do_slice = 0;
        P.tap = reinterpret_cast<float*>(h->d_sq_y);
        P.tap_stride = 2 * h->sq_stride;
    } else if (c.demod == SDR_DEMOD_NONE && !h->d_tap)
        return fail(h, SDR_E_ARG, "demod NONE needs cfg.tap_demod (complex output buffer)");
    dim3 grid(g.tiles, ncap), block(g.T);
    if (g.win && P.taps_complex)
        if (P.demod == SDR_DEMOD_MAG2)
            hipLaunchKernelGGL((k_frontend_win<true, SDR_DEMOD_MAG2>), grid, block, g.lds, h->stream, P);
        else
            hipLaunchKernelGGL((k_frontend_win<true, 0>), grid, block, g.lds, h->stream, P);
    else if (g.win)
        if (P.demod == SDR_DEMOD_MAG2)
            hipLaunchKernelGGL((k_frontend_win<false, SDR_DEMOD_MAG2>), grid, block, g.lds, h->stream, P);
        else if (P.demod == SDR_DEMOD_QUAD && P.do_slice && !P.tap && !h->no_quad_inst)
            hipLaunchKernelGGL((k_frontend_win<false, WIN_DM_QUAD_SLICE>), grid, block, g.lds, h->stream, P);
        else
            hipLaunchKernelGGL((k_frontend_win<false, 0>), grid, block, g.lds, h->stream, P);
    else if (g.mfma && g.mfma2) {
        dim3 dgrid((g.tiles + g.mfma2 - 1) / g.mfma2, ncap);
        hipLaunchKernelGGL(k_frontend_mfma2, dgrid, dim3(512), g.mfma2_lds, h->stream, P, g.mfma2);
    } else if (g.mfma)
        hipLaunchKernelGGL(k_frontend_mfma, grid, block, g.lds, h->stream, P);
    else if (h->in_fmt == SDR_IN_CU8 && P.vec_ok && !P.pre_tab && P.ktaps_n > 0 && g.use_lds && !h->no_raw8) {
        // bytes stay bytes in LDS: (count + up to 7 samples of alignment) rounded up to 16-B chunks
        const long long count = (long long)(g.T - 1) * c.decim + h->ntaps;
        const size_t lds = (size_t)((count + 7 + 7) / 8) * 16;
        hipLaunchKernelGGL(k_frontend_raw8, grid, block, lds, h->stream, P, h->ktaps);
    } else if (g.pipe_nch && P.vec_ok && !P.pre_tab) {
        dim3 pgrid((g.tiles + g.pipe_tpw - 1) / g.pipe_tpw, ncap);
        const bool cu8 = h->in_fmt == SDR_IN_CU8;
#define SDR_PIPE_LAUNCH(C, N) hipLaunchKernelGGL((k_frontend_pipe<C, N>), pgrid, block, g.lds, h->stream, P, h->ktaps, g.pipe_tpw)
        if (cu8 && g.pipe_nch == 8) SDR_PIPE_LAUNCH(true, 8);
        else if (cu8 && g.pipe_nch == 16) SDR_PIPE_LAUNCH(true, 16);
        else if (cu8) SDR_PIPE_LAUNCH(true, 32);
        else if (g.pipe_nch == 8) SDR_PIPE_LAUNCH(false, 8);
        else if (g.pipe_nch == 16) SDR_PIPE_LAUNCH(false, 16);
        else SDR_PIPE_LAUNCH(false, 32);
#undef SDR_PIPE_LAUNCH
    } else if (h->in_fmt == SDR_IN_CU8) {
        if (g.use_lds)
            hipLaunchKernelGGL((k_frontend<true, true>), grid, block, g.lds, h->stream, P, h->ktaps);
        else
            hipLaunchKernelGGL((k_frontend<false, true>), grid, block, 0, h->stream, P, h->ktaps);
    } else if (g.use_lds && P.demod != SDR_DEMOD_QUAD)
        hipLaunchKernelGGL((k_frontend<true, false, false>), grid, block, g.lds, h->stream, P, h->ktaps);
    else if (g.use_lds)
        hipLaunchKernelGGL((k_frontend<true, false>), grid, block, g.lds, h->stream, P, h->ktaps);
    else
        hipLaunchKernelGGL((k_frontend<false, false>), grid, block, 0, h->stream, P, h->ktaps);
    HIPCHK(h, hipGetLastError());
    if (post && !to_squelch) {
        // the post filter has its own tiling: one output per thread, 256 per tile
        CallGeom gp;
        fill_geom_gated(h, g.nout, g.keep_c0, gp);
        PostParams Q;
        memset(&Q, 0, sizeof Q);
        Q.v = h->d_tap;
        Q.hist = hist ? h->d_post_hist[h->post_hist_sel] : nullptr;  // streaming call
        Q.v_stride = h->tap_stride;
        Q.nout = g.nout;
        Q.ntaps = c.post_ntaps;
        Q.rtaps = h->d_post_rtaps;
        Q.gain = c.demod_gain;
        Q.lo = c.thr_lo;
        Q.hi = c.thr_hi;
        Q.keep_n = c.keep_n;
        Q.keep_c0 = g.keep_c0;
        Q.bits = h->d_bits + patch_off;
        Q.bits_stride = h->bits_stride;
        Q.tap = h->d_tap2;
        Q.tap_stride = h->tap_stride;
        Q.tsum = h->d_tsum;
        Q.tiles_per_capture = gp.tiles;
        Q.tile_out = gp.tile_out;
        size_t lds = (size_t)(gp.T + c.post_ntaps) * sizeof(float);
        hipLaunchKernelGGL(k_post_fir_slice, dim3(gp.tiles, ncap), dim3(gp.T), lds, h->stream, Q);
        HIPCHK(h, hipGetLastError());
    }
    return SDR_OK;
}

// squelch stage: channel filter -> complex items -> k_squelch -> gated items + per-capture counts
static int launch_squelch_stage(sdr_handle* h, const float2* x, long long stride, long long n, int ncap,
                                const float2* hist, int hist_len, const CallGeom& g, int fresh)
{
    if (g.nout > h->sq_stride - 2) return fail(h, SDR_E_CAPACITY, "squelch buffer too small");
    if (g.nout > 0) {
        int r = launch_front(h, x, stride, n, ncap, hist, hist_len, g, 0, true);
        if (r) return r;
    }
    SquelchParams S;
    memset(&S, 0, sizeof S);
    S.y = h->d_sq_y;
    S.y_stride = h->sq_stride;
    S.n = g.nout;
    S.alpha = h->sq_alpha;
    S.thr = h->sq_thr;
    S.gate = h->sq_gate;
    S.fresh = fresh;
    S.z = h->d_sq_z + 1;
    S.z_stride = h->sq_stride;
    S.cnt = h->d_sq_cnt;
    S.st = h->d_sq_st;
    hipLaunchKernelGGL(k_squelch, dim3(ncap), dim3(SQ_THREADS), 0, h->stream, S);
    HIPCHK(h, hipGetLastError());
    return SDR_OK;
}

// With the squelch enabled the caller has run launch_squelch_stage first and g
// describes the gated stream (g.nout = item count, or its upper bound in a batch).
static int launch_front_and_back(sdr_handle* h, const float2* x, long long stride, long long n, int ncap,
                                 const float2* hist, int hist_len, const CallGeom& g, long long patch_off,
                                 long long nb_total, int fresh, int use_state_thr, int run_decoder)
{
    const sdr_chain_cfg& c = h->cfg;
    const bool post = c.post_ntaps > 0;
    if (post && (int)h->post_taps.size() != c.post_ntaps)
        return fail(h, SDR_E_ARG, "post-demod filter configured but sdr_set_post_taps not called");
    if (g.tiles > h->tsum_stride) return fail(h, SDR_E_CAPACITY, "tile table too small");
    if (prof_begin(h)) HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING)], h->stream));
    if (g.nout > 0 && !h->squelch_on) {
        int r = launch_front(h, x, stride, n, ncap, hist, hist_len, g, patch_off, false);
        if (r) return r;
    } else if (g.nout > 0) {
        DemodParams Q;
        memset(&Q, 0, sizeof Q);
        Q.z = h->d_sq_z + 1;
        Q.z_stride = h->sq_stride;
        Q.cnt = h->d_sq_cnt;
        Q.nmax = g.nout;
        Q.atan_tab = h->d_atan_tab;
        Q.demod = c.demod;
        Q.gain = c.demod_gain;
        Q.lo = c.thr_lo;
        Q.hi = c.thr_hi;
        Q.do_slice = 1;
        Q.keep_n = c.keep_n;
        Q.keep_c0 = g.keep_c0;
        Q.bits = h->d_bits + patch_off;
        Q.bits_stride = h->bits_stride;
        Q.tap = c.tap_demod ? h->d_tap : nullptr;
        Q.tap_stride = h->tap_stride;
        Q.tsum = h->d_tsum;
        Q.tiles_per_capture = g.tiles;
        Q.tile_out = g.tile_out;
        hipLaunchKernelGGL(k_demod_slice, dim3(g.tiles, ncap), dim3(g.T), 0, h->stream, Q);
        HIPCHK(h, hipGetLastError());
    }
    return launch_back(h, ncap, g, patch_off, nb_total, fresh, use_state_thr, run_decoder);
}

// run-length + decode over `ncap` rows of sliced bytes (k_backend); closes the profiling slot
static int launch_back(sdr_handle* h, int ncap, const CallGeom& g, long long patch_off, long long nb_total, int fresh,
                       int use_state_thr, int run_decoder)
{
    const sdr_chain_cfg& c = h->cfg;
    const bool post = c.post_ntaps > 0;
    if (h->profile) HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING) + 1], h->stream));
    if (c.demod == SDR_DEMOD_NONE || h->skip_slice) {
        if (h->profile) {
            HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING) + 2], h->stream));
            h->pcount++;
        }
        return SDR_OK;
    }
    hipStream_t bs;
    {
        int rb = result_stream_begin(h, &bs);
        if (rb) return rb;
    }
    BackParams Bp;
    memset(&Bp, 0, sizeof Bp);
    Bp.bits = h->d_bits;
    Bp.bits_stride = h->bits_stride;
    Bp.nb = nb_total;
    Bp.tsum = g.nout > 0 ? h->d_tsum : nullptr;
    Bp.tiles_per_capture = g.tiles;
    Bp.tile_out = g.tile_out;
    if (post) {  // the slicer ran in k_post_fir_slice with its own tiling
        CallGeom gp;
        fill_geom_gated(h, g.nout, g.keep_c0, gp);
        Bp.tiles_per_capture = gp.tiles;
        Bp.tile_out = gp.tile_out;
    }
    Bp.nout = g.nout;
    Bp.nout_var = (h->squelch_on && fresh) ? h->d_sq_cnt : nullptr;
    Bp.keep_n = c.keep_n;
    Bp.keep_c0 = g.keep_c0;
    Bp.patch_off = patch_off;
    Bp.thr_init = c.thr_init;
    Bp.use_state_thr = use_state_thr;
    Bp.run_decoder = run_decoder;
    Bp.decoder = c.decoder;
    Bp.fresh = fresh;
    Bp.dec_rate = h->dec_rate;
    Bp.runs = h->d_runs;
    Bp.runs_cap = h->runs_cap;
    Bp.cur = h->d_cur;
    Bp.st = h->d_st;
    Bp.frames = h->d_frames;
    Bp.max_frames = c.max_frames;
    Bp.pow_m13 = h->d_pow;
    Bp.log2n = h->d_log2;
    Bp.runs_n = h->d_runs_n;
    Bp.rule = (!fresh && (size_t)ncap <= h->rule_rows && !is_stream_decoder(c.decoder) && c.decoder != SDR_DEC_NONE) ? h->d_rule : nullptr;
    const bool full_pass = run_decoder && c.decoder != SDR_DEC_NONE && !Bp.nout_var;
    if (h->reuse_next && h->runs_valid && full_pass && !h->no_reuse) {
        Bp.reuse_runs = 1;
        Bp.lvl_override = h->lvl_override;
        hipLaunchKernelGGL(k_backend, dim3(ncap), dim3(BK_THREADS), 0, bs, Bp);
    } else if (h->d_runs_seg && !Bp.nout_var && run_decoder && c.decoder != SDR_DEC_NONE && nb_total >= h->seg_min_row &&
        !h->no_seg) {
        // long rows: carries first (they patch the bytes), then the run starts of every 64-KiB-or-more
        // segment by its own workgroup, then the decoders on the gathered run table
        const long long round = BK_THREADS * 64;
        long long seg = 4 * round;
        if ((nb_total + seg - 1) / seg > BK_THREADS) seg = ((nb_total + BK_THREADS - 1) / BK_THREADS + round - 1) / round * round;
        Bp.seg_bytes = seg;
        Bp.nseg = (int)((nb_total + seg - 1) / seg);
        Bp.runs_seg = h->d_runs_seg;
        Bp.seg_cnt = h->d_seg_cnt;
        Bp.phase_sel = 1;
        const bool pieces = is_stream_decoder(c.decoder) && !h->no_pieces;
        if (pieces) {   // the phase-1 launch zeroes the piece headers
            Bp.pieces_b = h->d_runs_seg;
            Bp.pieces_hdr = h->d_pieces_hdr;
        }
        hipLaunchKernelGGL(k_backend, dim3(ncap), dim3(BK_THREADS), 0, bs, Bp);
        hipLaunchKernelGGL(k_runs_seg, dim3(Bp.nseg, ncap), dim3(BK_THREADS), 0, bs, Bp);
        Bp.phase_sel = 2;
        Bp.tsum = nullptr;
        Bp.fresh = 0;
        hipLaunchKernelGGL(k_backend, dim3(ncap), dim3(BK_THREADS), 0, bs, Bp);
        if (pieces) hipLaunchKernelGGL(k_decode_pieces, dim3(256, ncap), dim3(64), 0, bs, Bp);
    } else
        hipLaunchKernelGGL(k_backend, dim3(ncap), dim3(BK_THREADS), 0, bs, Bp);
    HIPCHK(h, hipGetLastError());
    if (!Bp.reuse_runs) h->runs_valid = full_pass;
    if (h->profile) {
        HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING) + 2], bs));
        h->pcount++;
    }
    return result_stream_end(h);
}

static void fill_geom(const sdr_handle* h, long long nout, long long in0, long long keep_c0, CallGeom& g,
                      bool to_squelch = false)
{
    g.nout = nout;
    g.in0 = in0;
    g.keep_c0 = keep_c0;
    const long long kn = h->cfg.keep_n;
    g.ne = nout > keep_c0 ? (nout - keep_c0 + kn - 1) / kn : 0;
    const int q = (h->cfg.demod == SDR_DEMOD_QUAD && !to_squelch) ? 1 : 0;
    g.mfma2 = 0;
    g.win = pick_win(h, nout, g.T, g.lds);
    g.mfma = !g.win && pick_mfma(h, g.T, g.lds);
    if (g.win) {
        g.use_lds = true;
        g.tile_out = WIN_R * g.T - q;
    } else if (g.mfma) {
        g.use_lds = true;
        g.tile_out = 4 * g.T - q;
        // complex output, no NCO behind the filter, full-size tiles: the two-team form when a tile's chunks fit
        // the registers of one team and two sample buffers fit the LDS
        const int D = h->cfg.decim, nt = h->ntaps;
        const bool plain = (h->cfg.demod == SDR_DEMOD_NONE || to_squelch) && (h->cfg.center_freq == 0.0 || h->prerotate);
        const long long tiles = (nout + 1023) / 1024;
        if (plain && g.T == 256 && h->mfma2_ok && tiles >= 2 && (mfma2_npos(D, nt) + 2) / 2 <= MFMA2_NCH * 256 &&
            mfma2_lds_bytes(D, nt) <= 88 * 1024) {
            g.mfma2_lds = mfma2_lds_bytes(D, nt);
            long long tpw = tiles / 512;                   // two workgroups per CU's worth before tiles are chained further
            if (h->mfma2_tpw > 0) tpw = h->mfma2_tpw;      // SDR_MFMA2_TPW (tuning), read at create
            g.mfma2 = (int)(tpw < 2 ? 2 : (tpw > 64 ? 64 : tpw));
        }
    } else {
        pick_tile(h, g.T, g.lds, g.use_lds);
        g.tile_out = g.T - q;
    }
    g.tiles = (int)((nout + g.tile_out - 1) / g.tile_out);
    g.pipe_nch = g.pipe_tpw = 0;
    if (!g.mfma && !g.win && (h->pipe_mode >= 2 || (h->pipe_mode == 1 && h->in_fmt == SDR_IN_CU8))) {
        // k_frontend_pipe: float tile only (cu8 needs no raw area), so the tile size is chosen
        // as for complex64 input; chunks per lane must fit one of the instantiations
        const int D = h->cfg.decim, nt = h->ntaps, spc = h->in_fmt == SDR_IN_CU8 ? 8 : 2;
        for (int T : {256, 128, 64}) {
            const long long count = (long long)(T - 1) * D + nt;
            const size_t lds = (size_t)(count + 2 * spc) * sizeof(float2);
            if (lds > 52 * 1024) continue;
            const long long per = ((count + 2 * spc - 2) / spc + T - 1) / T;
            const int nch = per <= 8 ? 8 : (per <= 16 ? 16 : (per <= 32 ? 32 : 0));
            if (!nch) continue;
            const int tile_out = T - q;
            const int tiles = (int)((nout + tile_out - 1) / tile_out);
            if (tiles < 2) break;
            g.T = T;
            g.lds = lds;
            g.use_lds = true;
            g.tile_out = tile_out;
            g.tiles = tiles;
            g.pipe_nch = nch;
            g.pipe_tpw = h->pipe_tpw < tiles ? h->pipe_tpw : tiles;
            break;
        }
    }
}

// geometry of k_demod_slice on the gated stream (one item per thread)
static void fill_geom_gated(const sdr_handle* h, long long nitems, long long keep_c0, CallGeom& g)
{
    memset(&g, 0, sizeof g);
    g.nout = nitems;
    g.keep_c0 = keep_c0;
    const long long kn = h->cfg.keep_n;
    g.ne = nitems > keep_c0 ? (nitems - keep_c0 + kn - 1) / kn : 0;
    g.T = 256;
    g.tile_out = 256;
    g.tiles = (int)((nitems + 255) / 256);
    g.use_lds = true;
}

static int finish_out(sdr_handle* h, sdr_out* out, unsigned long long n_in, long long nbits)
{
    if (!out) return SDR_OK;
    out->n_in = n_in;
    out->n_bits = (uint64_t)nbits;
    out->n_frames = 0;
    out->overflow = 0;
    out->pad_ = 0;
    return SDR_OK;
}

extern "C" int sdr_batch_process(sdr_handle* h, const void* iq, size_t n, size_t stride, int ncap,
                                 int is_dev, sdr_out* out)
{
    if (!h || !iq || ncap < 1) return fail(h, SDR_E_ARG, "sdr_batch_process: bad argument");
    if (h->pfb_M) return pfb_batch(h, iq, n, stride, ncap, is_dev, out);
    if (h->pre) {
        // upstream stage first; this stage then reads its complex output rows in place
        if (ncap > h->cfg.max_batch) return fail(h, SDR_E_CAPACITY, "sdr_batch_process: batch larger than the handle was created for");
        int r = sdr_batch_process(h->pre, iq, n, stride, ncap, is_dev, nullptr);
        if (r) return fail(h, r, "prefilter: " + h->pre->err);
        sdr_handle* keep = h->pre;
        h->pre = nullptr;
        r = sdr_batch_process(h, keep->d_tap, (size_t)keep->last_nout, (size_t)(keep->tap_stride / 2), ncap, 1, out);
        h->pre = keep;
        if (r == SDR_OK && out) out->n_in = (uint64_t)n * ncap;
        return r;
    }
    if (ncap > h->cfg.max_batch || n > h->cfg.max_samples)
        return fail(h, SDR_E_CAPACITY, "sdr_batch_process: batch larger than the handle was created for");
    if (stride < n) return fail(h, SDR_E_ARG, "sdr_batch_process: stride < n_per_capture");
    (void)hipSetDevice(h->device);
    // behind a gating squelch the back end reads per-capture counts the next front end would overwrite
    const bool ov = h->overlap && !h->squelch_on;
    if (ov) {
        int r = ov_next_buffers(h);
        if (r) return r;
    } else
        OV_JOIN(h);
    OvScope ovs(h, ov);
    const float2* x = (const float2*)iq;
    if (!is_dev) {
        int r = ensure_iq_staging(h, (size_t)ncap * stride);
        if (r) return r;
        HIPCHK(h, hipMemcpyAsync(h->d_iq, iq, (size_t)((ncap - 1) * stride + n) * h->ss(),
                                 hipMemcpyHostToDevice, h->stream));
        x = h->d_iq;
    }
    const sdr_chain_cfg& c = h->cfg;
    CallGeom g;
    // gr::sync_decimator: a finite capture of n items yields n/D outputs
    fill_geom(h, (long long)(n / c.decim), 0, c.keep_n - 1, g, h->squelch_on);
    unsigned long long saved = h->out_count;
    h->out_count = 0;  // every capture starts at NCO phase 0
    int r = SDR_OK;
    if (h->squelch_on) {
        // gated stream: per-capture lengths stay on the device (BackParams.nout_var);
        // g becomes the upper-bound geometry of the demod stage
        r = launch_squelch_stage(h, x, (long long)stride, (long long)n, ncap, nullptr, 0, g, 1);
        fill_geom_gated(h, (long long)(n / c.decim), c.keep_n - 1, g);
    }
    int run = 1;
    if (is_stream_decoder(c.decoder))
        run = ((double)g.ne * 1e6 / h->dec_rate > decoder_reset_limit(c.decoder)) ? 1 : 0;
    if (r == SDR_OK)
        r = launch_front_and_back(h, x, (long long)stride, (long long)n, ncap, nullptr, 0, g, 0, g.ne, 1, 0, run);
    h->out_count = saved;
    if (r) return r;
    h->last_ncap = ncap;
    h->last_nb = g.ne;
    h->last_nout = g.nout;
    h->last_bits_off = 0;
    h->last_keep_c0 = g.keep_c0;
    h->last_out0 = 0;
    return finish_out(h, out, (unsigned long long)n * ncap, g.ne);
}

// Streaming history as the kernels see it: only the samples that exist (the zeros
// before stream start are not representable in cu8 and are supplied by the kernels'
// own padding); hist[valid + idx] is sample idx < 0 of the call.
static int hist_valid(const sdr_handle* h)
{
    return h->in_count < (unsigned long long)h->hist_len ? (int)h->in_count : h->hist_len;
}
static const float2* hist_ptr(const sdr_handle* h)
{
    const char* b = reinterpret_cast<const char*>(h->d_hist[h->hist_sel]);
    return reinterpret_cast<const float2*>(b + (size_t)(h->hist_len - hist_valid(h)) * h->ss());
}

static int stream_decode_step(sdr_handle* h, const CallGeom& g, const float2* x, long long n,
                              bool have_iq, sdr_out* out, unsigned long long n_in)
{
    const sdr_chain_cfg& c = h->cfg;
    const long long total = h->raw_n + g.ne;
    if (total > h->bits_stride) return fail(h, SDR_E_CAPACITY, "rawBuffer capacity exceeded");
    int run = 0;
    if (is_stream_decoder(c.decoder))
        run = ((double)total * 1e6 / h->dec_rate > decoder_reset_limit(c.decoder)) ? 1 : 0;
    else if (c.decoder != SDR_DEC_NONE)
        run = 1;
    const long long off = (c.decoder == SDR_DEC_NONE || !is_stream_decoder(c.decoder)) ? 0 : h->raw_n;
    const long long nb_total = (off == 0) ? g.ne : total;
    int r;
    if (have_iq)
        r = launch_front_and_back(h, x, 0, n, 1, hist_ptr(h), hist_valid(h), g, off, nb_total, 0, 1, run);
    else {
        CallGeom g0 = g;
        g0.nout = 0;
        g0.tiles = 0;
        r = launch_front_and_back(h, nullptr, 0, 0, 1, nullptr, 0, g0, off, nb_total, 0, 1, run);
    }
    if (r) return r;
    h->last_ncap = 1;
    h->last_nb = g.ne;
    h->last_nout = g.nout;
    h->last_bits_off = off;
    if (is_stream_decoder(c.decoder)) h->raw_n = run ? 0 : total;
    return finish_out(h, out, n_in, g.ne);
}

extern "C" int sdr_process_iq(sdr_handle* h, const void* iq, size_t n, int is_dev, sdr_out* out)
{
    if (!h || (!iq && n)) return fail(h, SDR_E_ARG, "sdr_process_iq: bad argument");
    OV_JOIN(h);
    if (h->pfb_M) return pfb_stream(h, iq, n, is_dev, out);
    if (h->pre) {
        int r = sdr_process_iq(h->pre, iq, n, is_dev, nullptr);
        if (r) return fail(h, r, "prefilter: " + h->pre->err);
        sdr_handle* keep = h->pre;
        h->pre = nullptr;
        r = sdr_process_iq(h, keep->d_tap, (size_t)keep->last_nout, 1, out);
        h->pre = keep;
        if (r == SDR_OK && out) out->n_in = n;
        return r;
    }
    if (n > h->cfg.max_samples) return fail(h, SDR_E_CAPACITY, "sdr_process_iq: chunk larger than max_samples");
    (void)hipSetDevice(h->device);
    const sdr_chain_cfg& c = h->cfg;
    const float2* x = (const float2*)iq;
    if (!is_dev && n) {
        int r = ensure_iq_staging(h, n);
        if (r) return r;
        HIPCHK(h, hipMemcpyAsync(h->d_iq, iq, n * h->ss(), hipMemcpyHostToDevice, h->stream));
        x = h->d_iq;
    }
    const unsigned long long A = h->in_count;
    const long long D = c.decim;
    const unsigned long long m_end = (A + n) / D;  // outputs emitted once D items are consumed
    CallGeom g;
    const long long nout = (long long)(m_end - h->out_count);
    const long long in0 = (long long)(h->out_count * D) - (long long)A;
    fill_geom(h, nout, in0, h->keep_ctr - 1, g, h->squelch_on);
    int r;
    long long nkept = nout;  // items that reach the slicer
    if (h->squelch_on) {
        // the gate makes the item count data dependent: one 4-byte read-back per call
        r = launch_squelch_stage(h, x, 0, (long long)n, 1, hist_ptr(h), hist_valid(h), g, 0);
        if (r) return r;
        int32_t cnt = 0;
        HIPCHK(h, hipMemcpyAsync(&cnt, h->d_sq_cnt, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        nkept = cnt;
        fill_geom_gated(h, nkept, h->keep_ctr - 1, g);
    }
    r = stream_decode_step(h, g, x, (long long)n, true, out, n);
    if (r) return r;
    h->last_keep_c0 = g.keep_c0;
    h->last_out0 = h->out_count;
    h->carry_last = h->carry_next;
    h->carry_next = -1;
    // history for the next call: last hist_len samples of (hist ++ x)
    if (n) {
        const size_t H = (size_t)h->hist_len, ss = h->ss();
        const char* xb = reinterpret_cast<const char*>(x);
        char* cur = reinterpret_cast<char*>(h->d_hist[h->hist_sel]);
        char* nxt = reinterpret_cast<char*>(h->d_hist[h->hist_sel ^ 1]);
        HIPCHK(h, tail_update(h->stream, cur, xb, H * ss, n * ss, nxt, h->tail_defer));
        h->hist_sel ^= 1;
    }
    // history of the post-demod filter: last post_ntaps-1 demodulated floats
    if (c.post_ntaps > 1 && nout > 0) {
        const long long Hp = c.post_ntaps - 1;
        float* cur = h->d_post_hist[h->post_hist_sel];
        float* nxt = h->d_post_hist[h->post_hist_sel ^ 1];
        HIPCHK(h, tail_update(h->stream, cur, h->d_tap, (size_t)Hp * sizeof(float), (size_t)nout * sizeof(float), nxt, h->tail_defer));
        h->post_hist_sel ^= 1;
    }
    // counters
    h->in_count = A + n;
    h->out_count = m_end;
    if (g.ne == 0)
        h->keep_ctr -= nkept;
    else {
        long long last = g.keep_c0 + (g.ne - 1) * c.keep_n;
        h->keep_ctr = c.keep_n - (nkept - 1 - last);
    }
    return SDR_OK;
}

extern "C" int sdr_process_bits(sdr_handle* h, const int8_t* bits, size_t n, int is_dev, sdr_out* out)
{
    if (!h || (!bits && n)) return fail(h, SDR_E_ARG, "sdr_process_bits: bad argument");
    if (h->cfg.decoder == SDR_DEC_NONE) return fail(h, SDR_E_ARG, "sdr_process_bits: no decoder configured");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    const long long off = is_stream_decoder(h->cfg.decoder) ? h->raw_n : 0;
    if (off + (long long)n > h->bits_stride) return fail(h, SDR_E_CAPACITY, "sdr_process_bits: chunk too large");
    if (n)
        HIPCHK(h, hipMemcpyAsync(h->d_bits + off, bits, n, is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                 h->stream));
    CallGeom g;
    memset(&g, 0, sizeof g);
    g.ne = (long long)n;
    g.T = 256;
    g.tile_out = 256;
    return stream_decode_step(h, g, nullptr, 0, false, out, 0);
}

// ------------------------------------------------------------------ results
// Rows of a pitched device buffer -> a dense host array.  Not hipMemcpy2D: under
// rocprofv3 --kernel-trace on ROCm 7.2 its device-to-host form returned without
// delivering the data (every fetch read zeros).  Rows are packed on the device,
// then moved with one plain hipMemcpy.
__global__ void k_pack_rows(const unsigned char* __restrict__ src, size_t spitch, size_t width, size_t rows,
                            unsigned char* __restrict__ dst)
{
    const size_t total = width * rows;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / width, c = i - r * width;
        dst[i] = src[r * spitch + c];
    }
}

__global__ void k_copy_rows(const unsigned char* __restrict__ src, size_t spitch, unsigned char* __restrict__ dst,
                            size_t dpitch, size_t width, size_t rows)
{
    const size_t total = width * rows;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / width, c = i - r * width;
        dst[r * dpitch + c] = src[r * spitch + c];
    }
}

static int rows_to_host(sdr_handle* h, void* dst, const void* src, size_t spitch, size_t width, size_t rows)
{
    if (rows == 0 || width == 0) return SDR_OK;
    if (rows == 1 || spitch == width) {
        HIPCHK(h, hipMemcpy(dst, src, width * rows, hipMemcpyDeviceToHost));
        return SDR_OK;
    }
    const size_t total = width * rows;
    if (total > h->pack_cap) {
        if (h->d_pack) (void)dev_free(h->d_pack);
        h->d_pack = nullptr;
        h->pack_cap = 0;
        HIPCHK(h, dev_alloc((void**)&h->d_pack, total));
        h->pack_cap = total;
    }
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(256), 0, h->stream, (const unsigned char*)src, spitch,
                       width, rows, h->d_pack);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(dst, h->d_pack, total, hipMemcpyDeviceToHost));
    return SDR_OK;
}

extern "C" int sdr_fetch_frames(sdr_handle* h, sdr_frame* outp, size_t cap, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_fetch_frames: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int B = h->last_ncap;
    *n = 0;
    if (B == 0) return SDR_OK;
    std::vector<DecState> st(B);
    HIPCHK(h, hipMemcpy(st.data(), h->d_st, (size_t)B * sizeof(DecState), hipMemcpyDeviceToHost));
    size_t total = 0;
    int overflow = 0;
    for (int i = 0; i < B; ++i) {
        total += st[i].n_frames;
        overflow |= st[i].overflow;
    }
    *n = total;
    if (outp && cap) {
        std::vector<sdr_frame> all((size_t)B * h->cfg.max_frames);
        HIPCHK(h, hipMemcpy(all.data(), h->d_frames, all.size() * sizeof(sdr_frame), hipMemcpyDeviceToHost));
        size_t k = 0;
        for (int i = 0; i < B && k < cap; ++i)
            for (int j = 0; j < st[i].n_frames && k < cap; ++j) outp[k++] = all[(size_t)i * h->cfg.max_frames + j];
    }
    if (overflow) return fail(h, SDR_E_OVERFLOW, "device table overflow (runs/frames/symbols), code " + std::to_string(overflow));
    return SDR_OK;
}

extern "C" int sdr_fetch_bits(sdr_handle* h, int8_t* outp, size_t cap, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_fetch_bits: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t total = (size_t)h->last_ncap * (size_t)h->last_nb;
    *n = total;
    if (!outp || cap < total || total == 0) return SDR_OK;
    return rows_to_host(h, outp, h->d_bits + h->last_bits_off, (size_t)h->bits_stride, (size_t)h->last_nb,
                        (size_t)h->last_ncap);
}

extern "C" int sdr_fetch_demod(sdr_handle* h, float* outp, size_t cap, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_fetch_demod: bad argument");
    if (!h->d_tap) return fail(h, SDR_E_ARG, "sdr_fetch_demod: handle created without tap_demod");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t w = (size_t)h->last_nout * (h->cfg.demod == SDR_DEMOD_NONE ? 2 : 1);
    const size_t total = (size_t)h->last_ncap * w;
    *n = total;
    if (!outp || cap < total || total == 0) return SDR_OK;
    const float* src = (h->cfg.post_ntaps > 0 && h->d_tap2) ? h->d_tap2 : h->d_tap;
    return rows_to_host(h, outp, src, (size_t)h->tap_stride * sizeof(float), w * sizeof(float), (size_t)h->last_ncap);
}

extern "C" int sdr_fetch_squelch_counts(sdr_handle* h, int32_t* outp, size_t cap, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_fetch_squelch_counts: bad argument");
    if (!h->squelch_on) return fail(h, SDR_E_ARG, "sdr_fetch_squelch_counts: no squelch configured");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n = (size_t)h->last_ncap;
    if (!outp || cap < *n || *n == 0) return SDR_OK;
    HIPCHK(h, hipMemcpy(outp, h->d_sq_cnt, *n * sizeof(int32_t), hipMemcpyDeviceToHost));
    return SDR_OK;
}

extern "C" int sdr_decoder_state(sdr_handle* h, int32_t state[4])
{
    if (!h || !state) return SDR_E_ARG;
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DecState s;
    HIPCHK(h, hipMemcpy(&s, h->d_st, sizeof s, hipMemcpyDeviceToHost));
    state[0] = s.state;
    state[1] = s.n_cur;
    state[2] = s.prev_type;
    state[3] = s.chunk_id;
    return SDR_OK;
}

// ---- the decoder object's buffers, for the Python mirror's lazy attributes and its INFO/TRACE output --------
// ref: TS33C/stream_decoder.py:27-28 (currentSymbols, rawBuffer), :65-74 (edge_positions)
extern "C" int sdr_fetch_runs(sdr_handle* h, int row, int32_t* outp, size_t cap, size_t* n)
{
    if (!h || !n || row < 0) return fail(h, SDR_E_ARG, "sdr_fetch_runs: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n = 0;
    if (!h->runs_valid || row >= h->last_ncap) return SDR_OK;   // the last call only filled the rawBuffer (or had no decoder)
    int32_t rn = 0;
    HIPCHK(h, hipMemcpy(&rn, h->d_runs_n + row, sizeof rn, hipMemcpyDeviceToHost));
    const size_t R = (size_t)((unsigned)rn & 0x7fffffffu);
    if (R == 0) return SDR_OK;
    *n = R + 1;   // R run starts + the end of the buffer
    if (!outp || cap < R + 1) return SDR_OK;
    HIPCHK(h, hipMemcpy(outp, h->d_runs + (size_t)row * (size_t)(h->runs_cap + 2), (R + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    if ((unsigned)rn >> 31) return fail(h, SDR_E_OVERFLOW, "sdr_fetch_runs: the run table overflowed");
    return SDR_OK;
}

extern "C" int sdr_fetch_pending_symbols(sdr_handle* h, double* outp, size_t cap, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_fetch_pending_symbols: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DecState s;
    HIPCHK(h, hipMemcpy(&s, h->d_st, sizeof s, hipMemcpyDeviceToHost));
    const size_t k = s.n_cur > 0 ? (size_t)(s.n_cur < SDR_CUR_CAP ? s.n_cur : SDR_CUR_CAP) : 0;
    *n = k;
    if (!outp || cap < k || k == 0) return SDR_OK;
    HIPCHK(h, hipMemcpy(outp, h->d_cur, k * sizeof(double), hipMemcpyDeviceToHost));
    return SDR_OK;
}

extern "C" int sdr_fetch_raw_buffer(sdr_handle* h, int8_t* outp, size_t cap, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_fetch_raw_buffer: bad argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t k = h->raw_n > 0 ? (size_t)h->raw_n : 0;
    *n = k;
    if (!outp || cap < k || k == 0) return SDR_OK;
    HIPCHK(h, hipMemcpy(outp, h->d_bits, k, hipMemcpyDeviceToHost));
    return SDR_OK;
}

extern "C" int sdr_profile_enable(sdr_handle* h, int on)
{
    if (!h) return SDR_E_ARG;
    (void)hipSetDevice(h->device);
    if (on && h->pev.empty()) {
        h->pev.resize(3 * SDR_PROF_RING);
        for (auto& e : h->pev) HIPCHK(h, hipEventCreate(&e));
    }
    h->prof_every = on > 0 ? on : 0;
    h->prof_calls = 0;
    h->profile = false;
    h->pcount = 0;
    return SDR_OK;
}

extern "C" int sdr_profile_read(sdr_handle* h, float* front_ms, float* back_ms, int cap, int* n)
{
    if (!h || !n) return SDR_E_ARG;
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unsigned long long have = h->pcount < SDR_PROF_RING ? h->pcount : SDR_PROF_RING;
    if ((unsigned long long)cap < have) have = cap;
    *n = (int)have;
    for (unsigned long long i = 0; i < have; ++i) {
        unsigned long long slot = (h->pcount - have + i) % SDR_PROF_RING;
        float a = 0, b = 0;
        HIPCHK(h, hipEventElapsedTime(&a, h->pev[3 * slot], h->pev[3 * slot + 1]));
        HIPCHK(h, hipEventElapsedTime(&b, h->pev[3 * slot + 1], h->pev[3 * slot + 2]));
        if (front_ms) front_ms[i] = a;
        if (back_ms) back_ms[i] = b;
    }
    return SDR_OK;
}

__global__ void k_export_frames(const sdr_frame* __restrict__ src, const DecState* __restrict__ st, int ncap,
                                int max_frames, sdr_frame* __restrict__ dst, int32_t* __restrict__ counts)
{
    // one sdr_frame = 24 dwords; thread per dword, zero the unused slots
    const int words = sizeof(sdr_frame) / 4;
    long long total = (long long)ncap * max_frames * words;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        long long rec = i / words;
        int cap = (int)(rec / max_frames), k = (int)(rec % max_frames);
        unsigned int v = (k < st[cap].n_frames) ? reinterpret_cast<const unsigned int*>(src)[i] : 0u;
        reinterpret_cast<unsigned int*>(dst)[i] = v;
        if (i < ncap) counts[i] = st[i].n_frames;
    }
}

extern "C" int sdr_export_frames(sdr_handle* h, void* frames_dev, size_t frames_bytes, void* counts_dev,
                                 size_t counts_bytes)
{
    if (!h || !frames_dev || !counts_dev) return fail(h, SDR_E_ARG, "sdr_export_frames: null buffer");
    const int B = h->last_ncap;
    if (frames_bytes < (size_t)B * h->cfg.max_frames * sizeof(sdr_frame) || counts_bytes < (size_t)B * 4)
        return fail(h, SDR_E_CAPACITY, "sdr_export_frames: destination too small");
    if (B == 0) return SDR_OK;
    (void)hipSetDevice(h->device);
    long long total = (long long)B * h->cfg.max_frames * (sizeof(sdr_frame) / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    // with sdr_set_overlap on, device-side result readers run on the result stream (sdr_result_stream)
    OvScope ovs(h, h->overlap);
    hipStream_t bs;
    {
        int rb = result_stream_begin(h, &bs, true);
        if (rb) return rb;
    }
    hipLaunchKernelGGL(k_export_frames, dim3(blocks), dim3(256), 0, bs, h->d_frames, h->d_st, B,
                       h->cfg.max_frames, (sdr_frame*)frames_dev, (int32_t*)counts_dev);
    HIPCHK(h, hipGetLastError());
    return result_stream_end(h);
}

// ------------------------------------------------------------------ segments
// A chain with more than one memory (prefilter stage, post-demod filter, rotated input) is positioned by
// running it over the halo: its state at abs_index depends on the last seek_span() input samples only
// (every stage is FIR), so a chain started `span` samples early from empty memories has, at abs_index, the
// state of the stream from sample 0.
static bool seek_by_warmup(const sdr_handle* h) { return h->pre || h->cfg.post_ntaps > 0 || h->prerotate; }
static long long seek_span(const sdr_handle* h)
{
    const long long D = h->cfg.decim;
    // hist_len covers the channel filter and the discriminator's previous output; each of the post filter's
    // post_ntaps-1 remembered outputs is D inputs further back; + one decimation period of alignment
    long long L = h->hist_len + (h->cfg.post_ntaps > 0 ? (long long)h->cfg.post_ntaps * D : 0) + D;
    if (h->pre) L = h->pre->hist_len + (L + 1) * h->pre->cfg.decim;
    return L;
}

extern "C" int sdr_history_len(sdr_handle* h)
{
    if (!h) return SDR_E_ARG;
    return seek_by_warmup(h) ? (int)seek_span(h) : h->hist_len;
}

static int seek_warm(sdr_handle* h, uint64_t abs_index, const void* halo, size_t n_halo, int is_dev)
{
    const unsigned long long L = (unsigned long long)seek_span(h);
    const size_t need = (size_t)(abs_index < L ? abs_index : L);
    if (n_halo < need || (need && !halo))
        return fail(h, SDR_E_ARG, "sdr_seek: halo shorter than min(abs_index, sdr_history_len)");
    (void)hipSetDevice(h->device);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // every stage at absolute index a0 = abs_index - need with empty memories
    const unsigned long long a0 = abs_index - need;
    unsigned long long a_main = a0;
    if (sdr_handle* p = h->pre) {
        for (int i = 0; i < 2; ++i) HIPCHK(h, hipMemsetAsync(p->d_hist[i], 0, (size_t)p->hist_cap * sizeof(float2), h->stream));
        p->in_count = a0;
        p->out_count = a0 / (unsigned long long)p->cfg.decim;
        p->keep_ctr = 1;
        a_main = p->out_count;
    }
    for (int i = 0; i < 2; ++i) {
        HIPCHK(h, hipMemsetAsync(h->d_hist[i], 0, (size_t)h->hist_cap * sizeof(float2), h->stream));
        if (h->d_post_hist[i])
            HIPCHK(h, hipMemsetAsync(h->d_post_hist[i], 0, (size_t)(h->cfg.post_ntaps + 1) * sizeof(float), h->stream));
    }
    const unsigned long long D = (unsigned long long)h->cfg.decim, kn = (unsigned long long)h->cfg.keep_n;
    h->in_count = a_main;
    h->out_count = a_main / D;
    h->keep_ctr = (long long)(kn - (h->out_count % kn));
    h->raw_n = 0;
    int r = reset_dec_state(h);
    if (r) return r;
    // warm-up over the halo; what it emits (bits, frames, decoder state) is discarded below
    const sdr_handle* first = h->pre ? h->pre : h;
    const size_t ss = first->ss(), chunk = (size_t)first->cfg.max_samples;
    const char* src = (const char*)halo + (n_halo - need) * ss;
    for (size_t off = 0; off < need; off += chunk) {
        const size_t k = need - off < chunk ? need - off : chunk;
        r = sdr_process_iq(h, src + off * ss, k, is_dev, nullptr);
        if (r) return r;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->raw_n = 0;
    h->last_ncap = 0;
    h->last_nb = h->last_nout = h->last_bits_off = 0;
    return reset_dec_state(h);
}

extern "C" int sdr_seek(sdr_handle* h, uint64_t abs_index, const void* halo, size_t n_halo, int is_dev)
{
    if (!h) return fail(h, SDR_E_ARG, "sdr_seek: null handle");
    OV_JOIN(h);
    if (h->squelch_on)
        return fail(h, SDR_E_ARG, "sdr_seek: a squelch chain cannot be positioned from a halo (the power average is an IIR "
                                  "and the gate makes output positions data dependent)");
    if (seek_by_warmup(h)) return seek_warm(h, abs_index, halo, n_halo, is_dev);
    const size_t H = (size_t)h->hist_len;
    const size_t need = abs_index < H ? (size_t)abs_index : H;
    if (n_halo < need || (need && !halo))
        return fail(h, SDR_E_ARG, "sdr_seek: halo shorter than min(abs_index, sdr_history_len)");
    (void)hipSetDevice(h->device);
    char* cur = reinterpret_cast<char*>(h->d_hist[h->hist_sel]);
    const size_t ss = h->ss();
    HIPCHK(h, hipMemsetAsync(cur, 0, (size_t)h->hist_cap * sizeof(float2), h->stream));
    if (need)
        HIPCHK(h, hipMemcpyAsync(cur + (H - need) * ss, (const char*)halo + (n_halo - need) * ss, need * ss,
                                 is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const unsigned long long D = (unsigned long long)h->cfg.decim, kn = (unsigned long long)h->cfg.keep_n;
    h->in_count = abs_index;
    h->out_count = abs_index / D;  // outputs are emitted once D items are consumed
    h->keep_ctr = (long long)(kn - (h->out_count % kn));
    h->raw_n = 0;
    h->last_ncap = 0;
    h->last_nb = h->last_nout = h->last_bits_off = 0;
    return reset_dec_state(h);
}

extern "C" int sdr_segment_info_get(sdr_handle* h, sdr_segment_info* info)
{
    if (!h || !info) return fail(h, SDR_E_ARG, "sdr_segment_info_get: null argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    DecState s;
    HIPCHK(h, hipMemcpy(&s, h->d_st, sizeof s, hipMemcpyDeviceToHost));
    memset(info, 0, sizeof *info);
    const long long kn = h->cfg.keep_n, nb = h->last_nb, c0 = h->last_keep_c0;
    info->n_bits = (uint64_t)nb;
    info->first_bit_index = (uint64_t)((h->last_out0 + (unsigned long long)c0 + 1ull) / (unsigned long long)kn) - 1ull;
    long long und = nb;
    if (s.first_evt >= 0) {
        und = s.first_evt <= c0 ? 0 : ((long long)s.first_evt - c0 + kn - 1) / kn;
        if (und > nb) und = nb;
    }
    info->n_undetermined = (uint64_t)und;
    info->has_event = s.first_evt >= 0;
    info->last_state = s.thr_state;
    info->carry_in = h->carry_last;
    if (h->carry_next < 0) h->carry_next = s.thr_state;
    return SDR_OK;
}

extern "C" int sdr_segment_info_rows(sdr_handle* h, sdr_segment_info* info, size_t cap, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_segment_info_rows: null argument");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const size_t rows = (size_t)h->last_ncap;
    *n = rows;
    if (!info || cap < rows || rows == 0) return SDR_OK;
    std::vector<DecState> st(rows);
    HIPCHK(h, hipMemcpy(st.data(), h->d_st, rows * sizeof(DecState), hipMemcpyDeviceToHost));
    const long long kn = h->cfg.keep_n, nb = h->last_nb, c0 = h->last_keep_c0;
    for (size_t r = 0; r < rows; ++r) {
        sdr_segment_info& o = info[r];
        memset(&o, 0, sizeof o);
        o.n_bits = (uint64_t)nb;
        o.first_bit_index = (uint64_t)((h->last_out0 + (unsigned long long)c0 + 1ull) / (unsigned long long)kn) - 1ull;
        long long und = nb;
        if (st[r].first_evt >= 0) {
            und = st[r].first_evt <= c0 ? 0 : ((long long)st[r].first_evt - c0 + kn - 1) / kn;
            if (und > nb) und = nb;
        }
        o.n_undetermined = (uint64_t)und;
        o.has_event = st[r].first_evt >= 0;
        o.last_state = st[r].thr_state;
        o.carry_in = h->carry_last;
    }
    return SDR_OK;
}

// Decoder-only batch entry: `rows` independent sliced streams of n bytes (row r at bits + r*stride), each
// decoded from a fresh decoder object as ONE work() call -- sdr_batch_process without the front end.
extern "C" int sdr_batch_process_bits(sdr_handle* h, const int8_t* bits, size_t n, size_t stride, int rows, int is_dev,
                                      sdr_out* out)
{
    if (!h || !bits || rows < 1) return fail(h, SDR_E_ARG, "sdr_batch_process_bits: bad argument");
    if (h->cfg.decoder == SDR_DEC_NONE) return fail(h, SDR_E_ARG, "sdr_batch_process_bits: no decoder configured");
    if (rows > h->cfg.max_batch || (long long)n > h->bits_stride)
        return fail(h, SDR_E_CAPACITY, "sdr_batch_process_bits: more rows or longer rows than the handle was created for");
    if (stride < n) return fail(h, SDR_E_ARG, "sdr_batch_process_bits: stride < n");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    if (n) {
        if (rows == 1 || (stride == n && (size_t)h->bits_stride == n))
            HIPCHK(h, hipMemcpyAsync(h->d_bits, bits, n * (size_t)rows, is_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                     h->stream));
        else if (is_dev) {
            size_t total = n * (size_t)rows, blocks = (total + 255) / 256;
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)blocks), dim3(256), 0, h->stream, (const unsigned char*)bits, stride,
                               (unsigned char*)h->d_bits, (size_t)h->bits_stride, n, (size_t)rows);
            HIPCHK(h, hipGetLastError());
        } else
            for (int r = 0; r < rows; ++r)
                HIPCHK(h, hipMemcpyAsync(h->d_bits + (size_t)r * h->bits_stride, bits + (size_t)r * stride, n,
                                         hipMemcpyHostToDevice, h->stream));
    }
    CallGeom g;
    memset(&g, 0, sizeof g);
    g.ne = (long long)n;
    g.T = 256;
    g.tile_out = 256;
    int run = 1;
    if (is_stream_decoder(h->cfg.decoder))
        run = ((double)g.ne * 1e6 / h->dec_rate > decoder_reset_limit(h->cfg.decoder)) ? 1 : 0;
    if (prof_begin(h)) HIPCHK(h, hipEventRecord(h->pev[3 * (h->pcount % SDR_PROF_RING)], h->stream));
    int r = launch_back(h, rows, g, 0, g.ne, 1, 0, run);
    if (r) return r;
    h->last_ncap = rows;
    h->last_nb = g.ne;
    h->last_nout = 0;
    h->last_bits_off = 0;
    h->raw_n = 0;
    return finish_out(h, out, 0, g.ne);
}

extern "C" int sdr_export_bits(sdr_handle* h, void* bits_dev, size_t cap_bytes, size_t* n)
{
    if (!h || !n) return fail(h, SDR_E_ARG, "sdr_export_bits: null argument");
    const size_t rows = (size_t)h->last_ncap, width = (size_t)h->last_nb, total = rows * width;
    *n = total;
    if (total == 0) return SDR_OK;
    if (!bits_dev || cap_bytes < total) return fail(h, SDR_E_CAPACITY, "sdr_export_bits: destination too small");
    (void)hipSetDevice(h->device);
    const int8_t* src = h->d_bits + h->last_bits_off;
    OvScope ovs(h, h->overlap);   // as sdr_export_frames
    hipStream_t bs;
    {
        int rb = result_stream_begin(h, &bs, true);
        if (rb) return rb;
    }
    if (rows == 1 || (size_t)h->bits_stride == width) {
        HIPCHK(h, hipMemcpyAsync(bits_dev, src, total, hipMemcpyDeviceToDevice, bs));
        return result_stream_end(h);
    }
    size_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(256), 0, bs, (const unsigned char*)src,
                       (size_t)h->bits_stride, width, rows, (unsigned char*)bits_dev);
    HIPCHK(h, hipGetLastError());
    return result_stream_end(h);
}

// ------------------------------------------------------------------ one long capture, frames-only exchange
static_assert(sizeof(sdr_segment_pieces) == 56, "sdr_segment_pieces is mirrored field by field in sdr_amd/_lib.py");
// SURVEY 8e / north_star "RCCL ... used only to gather decoded frames and counts".  The stream decoder
// (TS33C/stream_decoder.py:76-110) cuts the symbol stream behind every width above reset_limit and decodes each
// closed piece from an empty currentSymbols, so a rank can decode the pieces that lie wholly inside its segment
// by itself.  What it cannot decide alone -- the runs up to and including its first over-limit run (the "head")
// and the runs behind its last over-limit run that does not touch the segment's end (the "tail") -- leaves the
// rank as two short lists of run-start positions; everything between them collapses into ONE run, which is
// over-limit by construction and therefore a piece of its own that decodes to nothing.  The reduced tables of all
// ranks, joined (sdr_amd/segments.py: stitch_tables), go through sdr_process_runs once.
extern "C" int sdr_segment_decode(sdr_handle* h, int carry_in, int32_t* head, size_t head_cap, int32_t* tail, size_t tail_cap,
                                  sdr_segment_pieces* info)
{
    if (!h || !info || !head || !tail || head_cap < 4 || tail_cap < 4)
        return fail(h, SDR_E_ARG, "sdr_segment_decode: bad argument");
    if (!is_stream_decoder(h->cfg.decoder)) return fail(h, SDR_E_ARG, "sdr_segment_decode: needs a stream decoder (TS33C / TFA)");
    if (h->pfb_M || h->last_ncap != 1 || h->last_bits_off != 0)
        return fail(h, SDR_E_ARG, "sdr_segment_decode: the segment must be ONE sdr_process_iq call behind sdr_seek");
    if (carry_in != 0 && carry_in != 1) return fail(h, SDR_E_ARG, "sdr_segment_decode: carry_in must be 0 or 1");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memset(info, 0, sizeof *info);
    const long long nb = h->last_nb;
    info->n_bits = nb;
    h->raw_n = 0;   // the segment is consumed here, not buffered for a later call
    if (nb == 0) return SDR_OK;
    DecState s;
    HIPCHK(h, hipMemcpy(&s, h->d_st, sizeof s, hipMemcpyDeviceToHost));
    // bytes in front of the segment's first threshold event hold the state the stream had when the segment began
    const long long kn = h->cfg.keep_n, c0 = h->last_keep_c0;
    long long und = nb;
    if (s.first_evt >= 0) {
        und = s.first_evt <= c0 ? 0 : ((long long)s.first_evt - c0 + kn - 1) / kn;
        if (und > nb) und = nb;
    }
    const int assumed = h->carry_last >= 0 ? h->carry_last : h->cfg.thr_init;
    if (carry_in != assumed && und > 0) HIPCHK(h, hipMemsetAsync(h->d_bits, carry_in, (size_t)und, h->stream));
    // the segment's bytes as a stream of their own: fresh decoder object, ONE work() call, decoder forced to run
    CallGeom g;
    memset(&g, 0, sizeof g);
    g.ne = nb;
    g.T = 256;
    g.tile_out = 256;
    int r = launch_back(h, 1, g, 0, nb, 1, 0, 1);
    if (r) return r;
    OV_JOIN(h);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(&s, h->d_st, sizeof s, hipMemcpyDeviceToHost));
    int32_t rn = 0;
    HIPCHK(h, hipMemcpy(&rn, h->d_runs_n, sizeof rn, hipMemcpyDeviceToHost));
    const long long R = (long long)((unsigned)rn & 0x7fffffffu);
    info->n_runs = (int32_t)R;
    info->n_frames = s.n_frames;
    if (((unsigned)rn >> 31) || s.overflow) info->flags |= SDR_SEG_TABLE_OVERFLOW;
    int8_t lv[2] = {0, 0};
    HIPCHK(h, hipMemcpy(&lv[0], h->d_bits, 1, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(&lv[1], h->d_bits + nb - 1, 1, hipMemcpyDeviceToHost));
    info->first_level = lv[0] != 0;
    info->last_level = lv[1] != 0;
    if (R < 1 || (info->flags & SDR_SEG_TABLE_OVERFLOW)) {
        info->flags |= SDR_SEG_TABLE_OVERFLOW;
        return SDR_OK;
    }
    const double limit = decoder_reset_limit(h->cfg.decoder), rate = h->dec_rate;
    // head: pos[0 .. i1 + 1], i1 = the first over-limit run
    const long long nh = R + 1 < (long long)head_cap ? R + 1 : (long long)head_cap;
    HIPCHK(h, hipMemcpy(head, h->d_runs, (size_t)nh * sizeof(int32_t), hipMemcpyDeviceToHost));
    long long i1 = -1;
    for (long long k = 0; k + 1 < nh && i1 < 0; ++k)
        if (run_width(head, k, rate) > limit) i1 = k;
    if (i1 < 0 || i1 == R - 1) {
        // no over-limit run, or only the segment's last one (whose true width the next segment knows): all of it is head
        if (nh < R + 1) {
            info->flags |= SDR_SEG_HEAD_TOO_LONG;
            return SDR_OK;
        }
        info->n_head = (int32_t)(R + 1);
        return SDR_OK;
    }
    info->n_head = (int32_t)(i1 + 2);
    info->keep_from = head[i1 + 1];
    // tail: pos[j + 1 .. R], j = the last over-limit run in [i1, R - 2]
    const long long span = R - i1;   // positions pos[i1 + 1 .. R]
    const long long nt = span < (long long)tail_cap ? span : (long long)tail_cap;
    HIPCHK(h, hipMemcpy(tail, h->d_runs + (R + 1 - nt), (size_t)nt * sizeof(int32_t), hipMemcpyDeviceToHost));
    // tail[t] = pos[R + 1 - nt + t]; run k (k >= R + 1 - nt) has its width in tail[k - (R + 1 - nt)] and the entry behind it
    long long j = -1;
    for (long long k = R - 2; k >= R + 1 - nt && j < 0; --k)
        if (run_width(tail, k - (R + 1 - nt), rate) > limit) j = k;
    if (j < 0) {
        if (nt < span) {
            info->flags |= SDR_SEG_TAIL_TOO_LONG;
            return SDR_OK;
        }
        j = i1;   // the window reaches back to the head's separator: nothing lies between head and tail
    }
    const long long first = j + 1 - (R + 1 - nt);   // index of pos[j + 1] inside tail[]
    info->n_tail = (int32_t)(R - j);
    if (first > 0) memmove(tail, tail + first, (size_t)info->n_tail * sizeof(int32_t));
    info->keep_to = tail[0];
    return SDR_OK;
}

// A sliced stream given as its run starts instead of its bytes: pos[0] = 0 < pos[1] < ... < pos[R] = the stream's
// length, the first and the last sample's level.  ONE work() call of the handle's stream decoder on it
// (TS33C/stream_decoder.py:39-110: everything behind line 63 is a function of the edge positions, rawBuffer[0]
// and rawBuffer[-2] only); frame end positions are positions of this table.
extern "C" int sdr_process_runs(sdr_handle* h, const int32_t* pos, size_t n_pos, int first_level, int last_level, sdr_out* out)
{
    if (!h || !pos || n_pos < 2) return fail(h, SDR_E_ARG, "sdr_process_runs: bad argument");
    if (!is_stream_decoder(h->cfg.decoder)) return fail(h, SDR_E_ARG, "sdr_process_runs: needs a stream decoder (TS33C / TFA)");
    if (h->pfb_M || h->raw_n != 0) return fail(h, SDR_E_ARG, "sdr_process_runs: the decoder holds buffered bytes (or is a channelizer)");
    if (h->no_reuse) return fail(h, SDR_E_ARG, "sdr_process_runs: run-table reuse is switched off (SDR_NO_REUSE)");
    const long long R = (long long)n_pos - 1;
    if (R > h->runs_cap) return fail(h, SDR_E_CAPACITY, "sdr_process_runs: more runs than the handle's run table holds");
    if (pos[0] != 0) return fail(h, SDR_E_ARG, "sdr_process_runs: pos[0] must be 0");
    for (long long k = 0; k < R; ++k)
        if (pos[k + 1] <= pos[k]) return fail(h, SDR_E_ARG, "sdr_process_runs: positions must increase");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);
    HIPCHK(h, hipMemcpyAsync(h->d_runs, pos, n_pos * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    const int32_t rn = (int32_t)R;
    HIPCHK(h, hipMemcpyAsync(h->d_runs_n, &rn, sizeof rn, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));   // rn lives on this frame
    const long long total = pos[R];
    CallGeom g;
    memset(&g, 0, sizeof g);
    g.ne = total;
    g.T = 256;
    g.tile_out = 256;
    const int run = ((double)total * 1e6 / h->dec_rate > decoder_reset_limit(h->cfg.decoder)) ? 1 : 0;
    if (!run) return fail(h, SDR_E_ARG, "sdr_process_runs: shorter than reset_limit (the reference would only buffer it)");
    h->reuse_next = true;
    h->runs_valid = true;
    h->lvl_override = 1 | (first_level ? 2 : 0) | (last_level ? 4 : 0);
    int r = launch_back(h, 1, g, 0, total, 0, 0, run);
    h->reuse_next = false;
    h->runs_valid = false;   // no bytes behind this table: nothing for sdr_redecode to work on
    h->lvl_override = 0;
    if (r) return r;
    h->last_ncap = 1;
    h->last_nb = 0;
    h->last_nout = 0;
    h->last_bits_off = 0;
    return finish_out(h, out, 0, total);
}

// ------------------------------------------------------------------ allocation-free batch entry point
// sdr_batch_process_static: sdr_batch_process for callers that cannot call hipMalloc (HIP graph
// capture, tight serving loops).  Everything the kernels need beyond the caller's input and output
// buffers -- sliced rows, tile summaries, run tables, pending symbols, decoder states, frame table --
// lives in ONE file-scope device array per GPU instead of the handle's hipMalloc'ed buffers.
// Host code never passes the array's name to a kernel: the device address is looked up once per
// device (scratch_init, from sdr_create) with hipGetSymbolAddress and kept in g_scratch_ptr.
#define SDR_STATIC_SCRATCH_BYTES (8u << 20)
#define SDR_STATIC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[SDR_STATIC_SCRATCH_BYTES];
static unsigned char* g_scratch_ptr[SDR_STATIC_MAX_DEVICES];   // per device: that GPU's copy of the array
// Every call on a GPU shares that GPU's array.  Calls on ONE stream are ordered by the stream; a call on another
// stream (another handle, another thread) is queued behind the last user's event, so back-to-back calls from
// different streams serialise on the device instead of racing.  Events are created once per device in
// scratch_init (nothing is allocated on the call path).  Not covered: a stream under graph capture (an event
// recorded there belongs to the graph and cannot order work outside it) -- replays of a captured call and direct
// calls that overlap on one GPU are the caller's to order.
static hipEvent_t g_scratch_ev[SDR_STATIC_MAX_DEVICES];
static hipStream_t g_scratch_last[SDR_STATIC_MAX_DEVICES];
static bool g_scratch_used[SDR_STATIC_MAX_DEVICES];
static std::mutex g_scratch_mu;

static int scratch_init(sdr_handle* h)
{
    const int dev = h->device;
    if (dev < 0 || dev >= SDR_STATIC_MAX_DEVICES) return SDR_OK;   // the static entry point refuses such a device
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev]) return SDR_OK;
    void* ptr = nullptr;
    HIPCHK(h, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));    // the handle's device is current (sdr_create)
    if (!ptr) return fail(h, SDR_E_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    HIPCHK(h, hipEventCreateWithFlags(&g_scratch_ev[dev], hipEventDisableTiming));
    g_scratch_ptr[dev] = static_cast<unsigned char*>(ptr);
    return SDR_OK;
}

extern "C" size_t sdr_static_scratch_bytes(void) { return SDR_STATIC_SCRATCH_BYTES; }

extern "C" int sdr_batch_process_static(sdr_handle* h, const void* iq_dev, size_t n, size_t stride, int ncap,
                                        int8_t* bits_dev, size_t bits_cap, sdr_frame* frames_dev, size_t frames_cap,
                                        int32_t* counts_dev, sdr_out* out)
{
    if (!h || !iq_dev || ncap < 1 || !bits_dev || !frames_dev || !counts_dev)
        return fail(h, SDR_E_ARG, "sdr_batch_process_static: null argument (all buffers are caller-owned DEVICE memory)");
    if (h->pfb_M || h->pre || h->squelch_on || h->cfg.post_ntaps > 0 || h->prerotate || h->skip_slice ||
        h->cfg.demod == SDR_DEMOD_NONE || h->cfg.tap_demod)
        return fail(h, SDR_E_ARG, "sdr_batch_process_static: plain chains only (channel filter -> MAG2/QUAD -> slicer -> decoder; "
                                  "no prefilter, squelch, post-demod filter, channelizer or demod tap)");
    if (stride < n) return fail(h, SDR_E_ARG, "sdr_batch_process_static: stride < n_per_capture");
    if (h->device < 0 || h->device >= SDR_STATIC_MAX_DEVICES || !g_scratch_ptr[h->device])
        return fail(h, SDR_E_HIP, "sdr_batch_process_static: no static scratch address for this device");
    (void)hipSetDevice(h->device);
    OV_JOIN(h);   // this entry point stays on h->stream, whatever sdr_set_overlap says
    const sdr_chain_cfg& c = h->cfg;
    CallGeom g;
    fill_geom(h, (long long)(n / c.decim), 0, c.keep_n - 1, g, false);
    if (g.nout >= 0x7fffffffLL - 1024) return fail(h, SDR_E_CAPACITY, "sdr_batch_process_static: more than 2^31 filter outputs per capture");
    // carve this call's tables out of the device's static array (256-B aligned pieces)
    const long long B = ncap;
    const long long bstride = ((g.ne + 63) / 64 + 1) * 64;
    const long long rcap = bstride;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t o_bits = carve((size_t)(B * bstride));
    const size_t o_tsum = carve((size_t)(B * (g.tiles + 1)) * sizeof(TileSum));
    const size_t o_runs = carve((size_t)(B * (rcap + 2)) * sizeof(int32_t));
    const size_t o_runs_n = carve((size_t)B * sizeof(int32_t));
    const size_t o_cur = carve((size_t)(B * SDR_CUR_CAP) * sizeof(double));
    const size_t o_st = carve((size_t)B * sizeof(DecState));
    const size_t o_frames = carve((size_t)(B * c.max_frames) * sizeof(sdr_frame));
    if (off > SDR_STATIC_SCRATCH_BYTES || bstride >= h->seg_min_row)
        return fail(h, SDR_E_CAPACITY, "sdr_batch_process_static: this batch needs " + std::to_string(off) + " bytes of scratch, the static array holds " +
                                       std::to_string(SDR_STATIC_SCRATCH_BYTES) + " (and rows must stay below " + std::to_string(h->seg_min_row) +
                                       " sliced bytes: longer ones take the segment tables); use sdr_batch_process");
    if (bits_cap < (size_t)(B * g.ne) || frames_cap < (size_t)(B * c.max_frames))
        return fail(h, SDR_E_CAPACITY, "sdr_batch_process_static: output buffer too small");
    unsigned char* const base = g_scratch_ptr[h->device];   // never the array's name: that is a host-side stand-in
    // the array's last user on this GPU, if it was another stream: this call's work waits for it on the device
    std::lock_guard<std::mutex> scratch_lk(g_scratch_mu);
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(h->stream, &cap_st);
    const bool capturing = cap_st != hipStreamCaptureStatusNone;
    if (!capturing && g_scratch_used[h->device] && g_scratch_last[h->device] != h->stream)
        HIPCHK(h, hipStreamWaitEvent(h->stream, g_scratch_ev[h->device], 0));
    // what must start at zero: row padding behind the sliced bytes, the frame table (record padding bytes)
    HIPCHK(h, hipMemsetAsync(base + o_bits, 0, (size_t)(B * bstride), h->stream));
    HIPCHK(h, hipMemsetAsync(base + o_frames, 0, (size_t)(B * c.max_frames) * sizeof(sdr_frame), h->stream));
    // run the handle's own launch path on the static tables
    struct Saved {
        int8_t* bits; long long bstride; TileSum* tsum; int tstride; int32_t *runs, *runs_seg, *runs_n; long long rcap;
        double* cur; DecState* st; sdr_frame* frames; unsigned long long out_count; int prof_every;
    } sv{h->d_bits, h->bits_stride, h->d_tsum, h->tsum_stride, h->d_runs, h->d_runs_seg, h->d_runs_n, h->runs_cap,
         h->d_cur, h->d_st, h->d_frames, h->out_count, h->prof_every};
    h->d_bits = reinterpret_cast<int8_t*>(base + o_bits);
    h->bits_stride = bstride;
    h->d_tsum = reinterpret_cast<TileSum*>(base + o_tsum);
    h->tsum_stride = g.tiles + 1;
    h->d_runs = reinterpret_cast<int32_t*>(base + o_runs);
    h->d_runs_seg = nullptr;
    h->d_runs_n = reinterpret_cast<int32_t*>(base + o_runs_n);
    h->runs_cap = rcap;
    h->d_cur = reinterpret_cast<double*>(base + o_cur);
    h->d_st = reinterpret_cast<DecState*>(base + o_st);
    h->d_frames = reinterpret_cast<sdr_frame*>(base + o_frames);
    h->out_count = 0;
    h->prof_every = 0;
    h->profile = false;
    int run = 1;
    if (is_stream_decoder(c.decoder))
        run = ((double)g.ne * 1e6 / h->dec_rate > decoder_reset_limit(c.decoder)) ? 1 : 0;
    int r = launch_front_and_back(h, (const float2*)iq_dev, (long long)stride, (long long)n, ncap, nullptr, 0, g, 0, g.ne, 1, 0, run);
    if (r == SDR_OK && g.ne > 0) {
        size_t total = (size_t)(B * g.ne), blocks = (total + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_pack_rows, dim3((unsigned)blocks), dim3(256), 0, h->stream, (const unsigned char*)h->d_bits,
                           (size_t)bstride, (size_t)g.ne, (size_t)B, (unsigned char*)bits_dev);
    }
    if (r == SDR_OK) {
        long long total = B * c.max_frames * (long long)(sizeof(sdr_frame) / 4);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_export_frames, dim3(blocks), dim3(256), 0, h->stream, h->d_frames, h->d_st, ncap, c.max_frames,
                           frames_dev, counts_dev);
    }
    h->d_bits = sv.bits; h->bits_stride = sv.bstride; h->d_tsum = sv.tsum; h->tsum_stride = sv.tstride;
    h->d_runs = sv.runs; h->d_runs_seg = sv.runs_seg; h->d_runs_n = sv.runs_n; h->runs_cap = sv.rcap;
    h->d_cur = sv.cur; h->d_st = sv.st; h->d_frames = sv.frames; h->out_count = sv.out_count; h->prof_every = sv.prof_every;
    h->runs_valid = false;   // the handle's run table does not describe what its next sdr_redecode would expect
    if (!capturing) {   // whatever was queued (also by a call that failed half way) is what the next user waits for
        if (hipEventRecord(g_scratch_ev[h->device], h->stream) == hipSuccess) {
            g_scratch_used[h->device] = true;
            g_scratch_last[h->device] = h->stream;
        }
    }
    if (r) return r;
    HIPCHK(h, hipGetLastError());
    return finish_out(h, out, (unsigned long long)n * ncap, g.ne);
}

#include "pfb_host.h"
#include "fm_host.h"
#include "preamble_host.h"
#include "selftest.h"
