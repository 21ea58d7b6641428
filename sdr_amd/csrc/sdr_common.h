// Internal declarations shared by the host side (sdr_host.cpp part of
// sdrhip.hip) and the kernels.  Not part of the public ABI (include/sdrhip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sdrhip.h"

#define SDR_MAX_POW_TABLE 4096   // n^(-1/3), log2(n) tables for np.histogram "auto"
#define SDR_CUR_CAP 1024         // pending symbols (currentSymbols) per stream
#define SDR_NCO_TAB 1024

// per-tile hysteresis summary written by the front-end kernels
struct TileSum {
    int32_t first_evt;   // demod-output index (within the capture/call) of the tile's first set/clear event, -1 if none
    int32_t last_state;  // state after the tile's last event (valid iff first_evt >= 0)
};

// decoder state that persists between work() calls (ref: TS33C/stream_decoder.py:20-32)
struct DecState {
    int32_t state;          // 0 idle, 1 frame
    int32_t n_cur;          // len(currentSymbols)
    int32_t prev_type;      // previousSymbolType
    int32_t chunk_id;
    int32_t thr_state;      // threshold_ff state carried between streaming calls
    int32_t n_frames;       // frames written by the last call
    int32_t overflow;       // !=0: runs/frames/symbol table overflow
    int32_t first_evt;      // demod-output index (within the last call) of its first threshold event, -1 if none (sdr_segment_info)
    double pulse_short_limit;
    double gap_short_limit;
};

// pigpio-callback rule decoders (somfy / berner / mumbi / kwmobile / baldr): what the callback object
// keeps between edges, carried between STREAMING calls (ref: somfy/sniff.py:19-26 start_tick, state, bits,
// hw_sync, clock).  Ticks are absolute microseconds since stream start.
struct RuleState {
    long long base;          // decoder-rate samples consumed by the calls so far
    long long start_tick, prev_tick;
    int32_t valid;           // 0: stream start (all defaults)
    int32_t state, nbits, level, n_sync, clock;
    int32_t pad_;
    double sync_sum;
    unsigned char bits[512];
};

#if defined(__HIPCC__)
// Workgroup barrier that orders LDS accesses only: s_waitcnt lgkmcnt(0) (vmcnt and expcnt left alone; gfx9 encoding)
// + s_barrier.  __syncthreads() also fences global memory (vmcnt(0)): a barrier inside a loop that keeps loads of
// the next round in flight, or right behind output stores, then waits for an HBM round trip it does not need.
// Both intrinsics are IntrNoMem to the optimizer, so the pair is bracketed by compiler-only memory clobbers: no LDS
// load or store may be moved across it (no instruction is emitted for them, and vmcnt is still not waited on --
// tests/test_codegen_guard.py checks the generated code).
__device__ __forceinline__ void lds_only_barrier()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#endif

struct FrontParams {
    const float2* x;        // capture 0, sample 0
    const float2* hist;     // streaming: hist_len samples that precede x[0]; nullptr = zeros
    long long stride;       // samples between captures
    long long n;            // samples per capture readable at x
    long long nout;         // demod outputs per capture in this call
    long long in0;          // input index of output 0's newest sample
    int hist_len;
    unsigned long long pre_inc;   // !=0: rotate every INPUT sample by nco64(pre_idx0+idx) while staging (xlating with taps=[1] ahead of the FIR)
    unsigned long long pre_idx0;  // absolute index of x[0]
    const float2* pre_tab;
    int vec_ok;             // x is 16-B aligned and stride is even (a multiple of 8 for cu8): 16-B direct-to-LDS staging allowed
    int in_fmt;             // SDR_IN_FC32: x/hist are complex64 | SDR_IN_CU8: x/hist are (u8 I, u8 Q) pairs, converted while staging
    int D;
    int ntaps;
    int taps_complex;
    const float* rtaps;     // taps reversed (oldest-sample-first order); (re,im) pairs if complex
    int ktaps_n;            // >0: the same reversed taps also sit in ktaps (kernel-argument segment -> scalar loads)
    unsigned long long nco_inc;
    unsigned long long nco_m0;  // absolute output index of output 0 (NCO phase origin)
    const float2* nco_tab;  // nullptr = no rotation
    const float* atan_tab;
    int demod;
    float gain, lo, hi;
    int do_slice;           // 0: only write y/tap (a post-demod stage follows)
    int keep_n;
    long long keep_c0;      // first emitted output index (keep_one_in_n counter - 1)
    int8_t* bits;
    long long bits_stride;
    float* tap;             // optional demod floats (nout per capture); float2 when demod==NONE
    long long tap_stride;
    TileSum* tsum;
    int tiles_per_capture;
    int tile_out;           // demod outputs per tile
    // Discriminator chains that only slice (no demod tap): threshold_ff needs the CLASS of gain*fast_atan2f(ti, tr)
    // (> hi, < lo, between), not its value.  phi_hi / phi_lo = the arctangent values at which fl(gain*a) crosses
    // the thresholds (host: exact float search); a sample whose true angle is further than the guard band from
    // both is classified by two cross-product sign tests, the others take the full arctangent (frontend_win.h).
    int quad_fast;          // 1: fields below are valid
    float q_c_hi, q_s_hi;   // cos, sin of phi_hi
    float q_c_lo, q_s_lo;   // cos, sin of phi_lo
    float q_band;           // the guard band in radians (applied to the 1-norm of (tr, ti))
};

#define SDR_KTAPS 64
struct FrontKTaps {          // second kernel argument of k_frontend<true>: <= 64 reversed taps, (re, im) pairs (im = 0 for real taps)
    float2 t[SDR_KTAPS];
};

struct PostParams {         // real FIR on demodulated floats + slicer (TFA/analyzer2.grc chain)
    const float* v;         // nout floats per capture
    const float* hist;      // streaming: the ntaps-1 floats that precede v[0]; nullptr = zeros
    long long v_stride;
    long long nout;
    int ntaps;
    const float* rtaps;
    float gain, lo, hi;
    int keep_n;
    long long keep_c0;
    int8_t* bits;
    long long bits_stride;
    float* tap;
    long long tap_stride;
    TileSum* tsum;
    int tiles_per_capture;
    int tile_out;
};

// pwr_squelch_cc state carried between streaming calls (row a7)
struct SquelchState {
    double p;           // single-pole IIR power estimate
    float2 last;        // last item that passed (quadrature_demod history after the gate)
    int32_t near_hits;  // calls in which the exact sequential re-run was taken
    int32_t pad_;
};

struct SquelchParams {
    const float2* y;    // channel-filter output, n items per capture
    long long y_stride;
    long long n;
    double alpha, thr;
    int gate;
    int fresh;          // batch: every capture starts from p = 0, last = 0
    float2* z;          // passed items; z[-1] is a valid slot (history item for the demodulator)
    long long z_stride;
    int32_t* cnt;       // items written per capture
    SquelchState* st;
};

struct DemodParams {    // demod + gain + slicer on the gated stream (ragged: cnt[cap] items)
    const float2* z;
    long long z_stride;
    const int32_t* cnt;
    long long nmax;     // grid extent in items; bits beyond cnt are zero-filled up to here
    const float* atan_tab;
    int demod;
    float gain, lo, hi;
    int do_slice;
    int keep_n;
    long long keep_c0;
    int8_t* bits;
    long long bits_stride;
    float* tap;
    long long tap_stride;
    TileSum* tsum;
    int tiles_per_capture;
    int tile_out;
};

struct BackParams {
    int8_t* bits;           // per capture bits_stride bytes
    long long bits_stride;
    long long nb;           // bytes to decode per capture (whole rawBuffer)
    const TileSum* tsum;    // nullptr: bits are final (sdr_process_bits)
    int tiles_per_capture;
    int tile_out;
    long long nout;         // demod outputs this call (for the last tile's extent)
    const int32_t* nout_var;// batch behind a gating squelch: per-capture nout; nb and run_decoder are then derived per capture
    int keep_n;
    long long keep_c0;
    long long patch_off;    // byte offset inside the capture's bits where this call's output starts (streaming append)
    int thr_init;           // used when use_state_thr == 0
    int use_state_thr;      // streaming: carry-in comes from DecState.thr_state
    int run_decoder;        // 0: only resolve carries (rawBuffer still filling)
    int decoder;
    int fresh;              // batch: reset DecState per capture before decoding
    double dec_rate;        // decoder sample_rate
    int32_t* runs;          // per capture runs_cap+2 ints
    long long runs_cap;
    double* cur;            // per capture SDR_CUR_CAP doubles
    DecState* st;           // per capture
    sdr_frame* frames;      // per capture max_frames
    int max_frames;
    const double* pow_m13;  // n^(-1/3)
    const double* log2n;    // log2(n)
    // long rows: run starts found by k_runs_seg, one workgroup per seg_bytes of a row
    int phase_sel;          // 0: all phases; 1: phase 0 only (carries); 2: phases 1-2, runs gathered from runs_seg
    int nseg;
    long long seg_bytes;    // multiple of the 16-KiB round
    int32_t* runs_seg;      // per capture bits_stride ints: segment s keeps its edges from index s*seg_bytes on
    int32_t* seg_cnt;       // per capture nseg counts
    int32_t* runs_n;        // per capture: number of runs the last full pass found (bit 31: table overflowed)
    int reuse_runs;         // 1: phase 1 is skipped, runs and runs_n come from that pass
    // long rows, stream decoders: the closed pieces between the first and the last are decoded by
    // k_decode_pieces, one wavefront each
    int32_t* pieces_b;      // per capture (bits_stride ints, the runs_seg slice): run index behind every over-limit width
    int32_t* pieces_hdr;    // per capture 8 ints: [0] 1 = delegated, [1] over-limit widths, [2] frames of the first piece, [3] frames of the middle pieces
    // sdr_redecode_multi: workgroup v decodes row v % vrows with decoder number v / vrows of dec_pack (4 bits
    // each; NOT an array: indexing a by-value kernel argument dynamically sends the whole struct to scratch
    // memory -- measured: every k_backend launch 2.4x slower); DecState, pending symbols and frame table are
    // indexed by v (ndec == 0: one decoder, v == row)
    int ndec;
    int vrows;
    unsigned int dec_pack;
    RuleState* rule;        // streaming call with a callback-rule decoder: per-row state between calls; nullptr = one call per capture
    int lvl_override;       // sdr_process_runs (run table without bytes): bit 0 = valid, bit 1 = first sample's level, bit 2 = last sample's
};
