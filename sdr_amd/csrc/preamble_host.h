// Host side of the somfy preamble-finder block (include/sdrhip.h sdr_preamble_*).
#pragma once
#include "preamble.h"

struct sdr_preamble {
    int device = 0;
    hipStream_t stream = nullptr;
    int8_t* d_in = nullptr;
    int8_t* d_out = nullptr;
    size_t cap = 0;
    PreambleState* d_st = nullptr;
    std::string err;
};

static int pfail(sdr_preamble* h, int code, const std::string& msg)
{
    if (h) h->err = msg;
    g_err = msg;
    return code;
}
#define PCHK(h, call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess)                                                                \
            return pfail(h, SDR_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_));   \
    } while (0)

extern "C" const char* sdr_preamble_last_error(sdr_preamble* h) { return h ? h->err.c_str() : g_err.c_str(); }

extern "C" int sdr_preamble_destroy(sdr_preamble* h)
{
    if (!h) return SDR_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->d_in) (void)dev_free(h->d_in);
    if (h->d_out) (void)dev_free(h->d_out);
    if (h->d_st) (void)dev_free(h->d_st);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SDR_OK;
}

extern "C" int sdr_preamble_create(sdr_preamble** out)
{
    if (!out) return pfail(nullptr, SDR_E_ARG, "sdr_preamble_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return pfail(nullptr, SDR_E_NODEVICE, "sdr_preamble_create: no HIP device (this library has no CPU path)");
    sdr_preamble* h = new sdr_preamble();
    auto body = [&]() -> int {
        PCHK(h, hipGetDevice(&h->device));
        PCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        PCHK(h, dev_alloc((void**)&h->d_st, sizeof(PreambleState)));
        PCHK(h, memset_done(h->d_st, 0, sizeof(PreambleState)));
        return SDR_OK;
    };
    int rc = body();
    if (rc != SDR_OK) {
        sdr_preamble_destroy(h);
        return rc;
    }
    *out = h;
    return SDR_OK;
}

extern "C" int sdr_preamble_work(sdr_preamble* h, const int8_t* in, size_t n, int is_dev, int8_t* outp,
                                 sdr_preamble_info* info)
{
    if (!h || ((!in || !outp) && n)) return pfail(h, SDR_E_ARG, "sdr_preamble_work: bad argument");
    static_assert(sizeof(sdr_preamble_info) == sizeof(PreambleState), "info mirrors the device state");
    (void)hipSetDevice(h->device);
    const int8_t* x = in;
    int8_t* y = outp;
    if (!is_dev) {
        if (n > h->cap) {
            if (h->d_in) (void)dev_free(h->d_in);
            if (h->d_out) (void)dev_free(h->d_out);
            h->d_in = h->d_out = nullptr;
            h->cap = 0;
            const size_t want = n < 65536 ? 65536 : n;
            PCHK(h, dev_alloc((void**)&h->d_in, want));
            PCHK(h, dev_alloc((void**)&h->d_out, want));
            h->cap = want;
        }
        if (n) PCHK(h, hipMemcpyAsync(h->d_in, in, n, hipMemcpyHostToDevice, h->stream));
        x = h->d_in;
        y = h->d_out;
    }
    hipLaunchKernelGGL(k_preamble, dim3(1), dim3(PRE_THREADS), 0, h->stream, x, (long long)n, y, h->d_st);
    PCHK(h, hipGetLastError());
    PreambleState s;
    PCHK(h, hipMemcpyAsync(&s, h->d_st, sizeof s, hipMemcpyDeviceToHost, h->stream));
    if (!is_dev && n) PCHK(h, hipMemcpyAsync(outp, h->d_out, n, hipMemcpyDeviceToHost, h->stream));
    PCHK(h, hipStreamSynchronize(h->stream));
    if (info) memcpy(info, &s, sizeof s);
    return SDR_OK;
}
