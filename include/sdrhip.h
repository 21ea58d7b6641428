/*
 * sdrhip.h -- C ABI of libsdrhip.so: the MI355X (gfx950) receive chain that
 * stands in for the GNU Radio blocks + embedded Python decoders of
 * henrythasler/sdr.  POD structs, plain pointers and sizes only; no C++ or
 * torch types cross this boundary.  Every entry point returns 0 on success or
 * a negative SDR_E_* code and never throws; sdr_last_error() gives the text.
 *
 * One handle = one configured chain (what one GRC flowgraph wires up) with its
 * own HIP stream and device buffers.  Calls on one handle are serialised by
 * the caller; different handles may be used from different threads
 * (GNU Radio runs one thread per block and calls work() serially per block).
 *
 * "ref:" comments name the reference interface each entry point replaces
 * (paths relative to the reference checkout).
 */
#ifndef SDRHIP_H
#define SDRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDR_API __attribute__((visibility("default")))

/* error codes */
enum {
    SDR_OK = 0,
    SDR_E_ARG = -1,      /* bad argument / configuration */
    SDR_E_HIP = -2,      /* HIP runtime error (text in sdr_last_error) */
    SDR_E_CAPACITY = -3, /* input larger than the handle was created for */
    SDR_E_OVERFLOW = -4, /* device-side table (runs / frames) overflowed */
    SDR_E_NODEVICE = -5  /* no usable gfx950 device */
};

/* gr::filter::firdes::win_type values, as the GRC files store them
 * (ref: TS33C/live_decoder.grc:917-988 `win firdes.WIN_HAMMING`). */
enum { SDR_WIN_HAMMING = 0, SDR_WIN_HANN = 1, SDR_WIN_BLACKMAN = 2, SDR_WIN_RECTANGULAR = 3,
       SDR_WIN_KAISER = 4, SDR_WIN_BLACKMAN_HARRIS = 5 };

/* demodulator after the channel filter */
enum {
    SDR_DEMOD_NONE = 0, /* complex out (filter stage only) */
    SDR_DEMOD_MAG2 = 1, /* blocks_complex_to_mag_squared (ref: TS33C/live_decoder.grc:307-349) */
    SDR_DEMOD_QUAD = 2  /* analog_quadrature_demod_cf (ref: TFA/decoder.grc:450-481) */
};

/* sample format of the IQ input (sdr_set_input_format) */
enum {
    SDR_IN_FC32 = 0, /* complex64, what every GNU Radio block of the reference exchanges */
    SDR_IN_CU8 = 1   /* rtl-sdr bytes (u8 I, u8 Q), converted as rtlsdr_source does: (u8 - 127.4) / 128 */
};

/* bit decoder fed by the sliced stream */
enum {
    SDR_DEC_NONE = 0,
    SDR_DEC_TS33C = 1,  /* ref: TS33C/stream_decoder.py:13-220 */
    SDR_DEC_TFA = 2,    /* ref: TFA/stream_decoder.py:13-218 */
    SDR_DEC_TFA_GRC = 3,/* ref: TFA/analyzer2.grc:765-983 (adds the NaN guard) */
    SDR_DEC_SOMFY = 4,  /* ref: somfy/sniff.py:28-117 + somfy/decoder.py:27-58 */
    SDR_DEC_BERNER = 5, /* ref: berner/sniff.py:21-44 */
    SDR_DEC_MUMBI = 6,  /* ref: mumbi/sniff.py:26-65 */
    SDR_DEC_KWMOBILE = 7, /* ref: kwmobile/decoder.py:95-135 (Nexus-type PPM, 36 bits) */
    SDR_DEC_BALDR = 8,  /* ref: baldr/decoder.py:87-142 (PPM, 37 bits, one decode per transmission) */
    SDR_DEC_VELUX = 9   /* ref: velux/decoder.py:30-101 (io-homecontrol: start of frame, UART framing, CRC-16/KERMIT);
                           the sliced stream must hold one byte per symbol */
};

/* frame status (the reference reports these as debug strings) */
enum {
    SDR_FRAME_OK = 0,
    SDR_FRAME_BAD_LENGTH = 1,     /* "Invalid packet length" */
    SDR_FRAME_PARITY = 2,         /* "Parity check failed" */
    SDR_FRAME_UNKNOWN_HEADER = 3, /* "Unknown Sensor or Header" */
    SDR_FRAME_CRC = 4,            /* "CRC-Check failed" */
    SDR_FRAME_SYMBOL_LENGTH = 5,  /* "Symbol length error" (GRC twin only) */
    SDR_FRAME_CHECKSUM = 6,       /* somfy checksum != 0 */
    SDR_FRAME_EXC_RESHAPE = 16,   /* reference raises ValueError (odd sync count, TFA/stream_decoder.py:137) */
    SDR_FRAME_EXC_INDEX = 17,     /* reference raises IndexError (no data start, TFA/stream_decoder.py:143) */
    SDR_FRAME_EXC_NAN = 18,       /* reference raises ValueError int(NaN) (TFA/stream_decoder.py:138) */
    SDR_FRAME_EXC_SHORT = 19      /* reference raises (bitstring unpack: fewer bits than the packet needs, velux/decoder.py:55,63,89) */
};

/* One flowgraph's parameters.  Field names follow the GRC block parameters.
 * ref: the <block> parameter lists of TS33C/live_decoder.grc:3292-3377 wiring. */
typedef struct sdr_chain_cfg {
    double sample_rate;  /* samp_rate of the IQ stream, Hz */
    double center_freq;  /* freq_xlating_fir_filter center_freq, Hz; 0 = plain fir_filter_ccf */
    int32_t decim;       /* channel-filter decimation D (>=1) */
    int32_t demod;       /* SDR_DEMOD_* */
    float demod_gain;    /* multiply_const after mag2, or quadrature_demod gain */
    float thr_lo;        /* threshold_ff low */
    float thr_hi;        /* threshold_ff high */
    int32_t thr_init;    /* threshold_ff initial state */
    int32_t keep_n;      /* keep_one_in_n after the slicer (>=1) */
    int32_t decoder;     /* SDR_DEC_* */
    double decoder_rate; /* sample_rate parameter of the decoder block; 0 = sample_rate/decim/keep_n */
    int32_t post_ntaps;  /* >0: real low_pass_filter on the demodulated floats (TFA/analyzer2.grc:1075-1147); taps via sdr_set_post_taps */
    int32_t tap_demod;   /* !=0: keep the demodulated floats for sdr_fetch_demod (parity tests) */
    int32_t max_batch;   /* captures per sdr_batch_process call (>=1) */
    int32_t max_frames;  /* frame records kept per capture / per stream call */
    uint64_t max_samples;/* IQ samples per capture / per streaming call */
} sdr_chain_cfg;

/* A decoded frame, fixed size.  Integers only; the Python mirror formats the
 * reference's print lines from them. */
typedef struct sdr_frame {
    int32_t capture;   /* capture index inside the batch (0 when streaming) */
    int32_t decoder;   /* SDR_DEC_* */
    int32_t status;    /* SDR_FRAME_* */
    int32_t nbits;     /* raw bits the symbol rule produced */
    int32_t nbytes;    /* valid bytes in `bytes` */
    uint8_t bytes[16]; /* packet (TS33C 10, TFA 11, somfy 7 de-obfuscated, berner 4, mumbi 5, kwmobile 5, baldr 5) */
    uint8_t raw[8];    /* somfy: the 7 on-air (obfuscated) bytes */
    int32_t v[4];      /* TS33C: channel, rolling code, battery_ok, humidity | TFA: id, humidity, battery_ok, 0 | somfy: control, rolling code, address, clock
                          | kwmobile: sensor_id, humidity, battery_ok, channel | baldr: sensor_id, humidity, txMode, channel
                          | velux: the packet (nbytes <= 36: ff 33, control1, message, crc16 LE) runs from bytes[0] on
                            through raw and v (40 contiguous bytes); pad_ = control1, temp_tenths = counter,
                            lim[0] = frame start (symbol index), lim[1] = message length */
    int32_t temp_tenths; /* TS33C/TFA/kwmobile/baldr temperature in 0.1 C */
    int32_t pad_;
    int32_t pad2_;     /* always 0.  Spelled out (the compiler would put 4 bytes of padding here anyway) so that it is
                          assigned like any field: records then compare equal byte for byte */
    double lim[2];     /* TS33C: pulse/gap short limit (us) | TFA: symbol_length (us), 0 */
    int64_t end_pos;   /* index (decoder-rate samples, per call) where the frame ended */
} sdr_frame;

/* per-call summary filled by sdr_process_iq / sdr_batch_process */
typedef struct sdr_out {
    uint64_t n_in;     /* IQ samples consumed */
    uint64_t n_bits;   /* sliced bytes produced (per capture in batch mode) */
    uint64_t n_frames; /* frame records available to sdr_fetch_frames */
    int32_t overflow;  /* !=0: a device table overflowed (also returns SDR_E_OVERFLOW) */
    int32_t pad_;
} sdr_out;

typedef struct sdr_handle sdr_handle;

/* ---- filter design (host) ------------------------------------------------
 * ref: the `low_pass_filter` GRC block = firdes.low_pass(gain, samp_rate,
 * cutoff_freq, width, win, beta), e.g. TS33C/live_decoder.grc:917-988.
 * Returns ntaps (>0), or -(needed) if cap is too small. */
SDR_API int sdr_firdes_low_pass(double gain, double fs, double fc, double tw, int win, double beta,
                                float* taps, int cap);
/* ref: rational_resampler_xxx with empty taps (FM-Radio/FM-Radio.grc:1378-1493). */
SDR_API int sdr_resampler_taps(int interp, int decim, double fractional_bw, float* taps, int cap);

/* ---- chain life cycle -----------------------------------------------------
 * ref: what `top_block.__init__` does when GRC instantiates the blocks. */
SDR_API int sdr_create(const sdr_chain_cfg* cfg, sdr_handle** out);
SDR_API int sdr_destroy(sdr_handle* h);
/* channel-filter taps; n real floats, or n (re,im) pairs when is_complex.
 * Default after sdr_create is the single tap 1 (ref: `taps 1` in
 * FM-Radio/FM-Radio.grc:615-671).  With center_freq != 0 the taps are
 * translated as freq_xlating_fir_filter_ccc does.  Setting taps (channel or
 * post-demod filter) RESTARTS the stream: history, sample counters, keep_one_in_n
 * counter, threshold and decoder state return to their initial values. */
SDR_API int sdr_set_taps(sdr_handle* h, const float* taps, int n, int is_complex);
SDR_API int sdr_set_post_taps(sdr_handle* h, const float* taps, int n);
/* An upstream channel-filter stage (fir_filter_ccf, real taps, decimation `decim`)
 * in front of this chain's own filter, for flowgraphs that cascade two:
 * ref: mumbi/FS300.grc:731-802 low_pass_filter (97 taps) -> :1691-1746
 * rational_resampler_ccc 1/10 (its polyphase filter with interpolation 1 IS a
 * decimating FIR: pass sdr_resampler_taps(1, 10) as this chain's taps, decim 10).
 * cfg.sample_rate / cfg.max_samples of the chain then refer to the stream BETWEEN
 * the two stages; sdr_process_iq / sdr_batch_process take the upstream input. */
SDR_API int sdr_set_prefilter(sdr_handle* h, const float* taps, int n, int decim);
/* Input sample format, SDR_IN_*.  SDR_IN_CU8 = the rtl-sdr on-disk/USB format:
 * what `rtlsdr_source` (ref: TS33C/live_decoder.grc:3292 wiring, block rtlsdr_source_0)
 * receives from the dongle and what the READMEs record (`rtl_sdr -s 2048000
 * sample.cu8`, ref: berner/README.md:21, mumbi/README.md:20); the conversion
 * (u8 - 127.4) / 128 per component happens while the tile is staged, so the chain
 * reads 2 bytes per sample instead of 8.  iq pointers, strides and halos of every
 * entry point then count (I,Q) byte pairs.  Resets the streaming state.  With a
 * prefilter the format applies to the upstream stage's input. */
SDR_API int sdr_set_input_format(sdr_handle* h, int fmt);
/* analog_pwr_squelch_cc (ramp 0) between the channel filter and the demodulator:
 * p = alpha*|y|^2 + (1-alpha)*p (double), an item passes iff !(p < 10^(db/10));
 * gate != 0 drops muted items (the stream gets shorter), gate == 0 zeroes them.
 * ref: TFA/decoder.grc:391-446, TFA/analyzer.grc:544-599 (alpha 1e-4, -40 dB, gate True).
 * Resets the squelch state.  Chains with a post-demod filter are not supported. */
SDR_API int sdr_set_squelch(sdr_handle* h, double threshold_db, double alpha, int gate);
/* items that passed the squelch in the last call, one int32 per capture.  In a
 * batch behind a gating squelch sdr_out.n_bits / the row pitch of sdr_fetch_bits
 * and sdr_fetch_demod are the ungated upper bound; rows are zero (bits) or
 * undefined (demod floats) beyond the capture's own count. */
SDR_API int sdr_fetch_squelch_counts(sdr_handle* h, int32_t* out, size_t cap, size_t* n);
/* run on a caller-owned HIP stream (hipStream_t as void*), e.g. torch's
 * current stream; NULL = the handle's own stream. */
SDR_API int sdr_set_stream(sdr_handle* h, void* hip_stream);
SDR_API int sdr_synchronize(sdr_handle* h);

/* Overlapped batches (throughput mode; no counterpart in the reference, whose flowgraph runs its blocks as
 * concurrent threads -- this is the same pipelining between the filter blocks and the decoder block, on one
 * GPU).  With on != 0 a second, library-owned "result stream" takes the back end (run-length + decoders) of
 * every sdr_batch_process and the device-side result readers (sdr_redecode, sdr_redecode_multi,
 * sdr_export_frames, sdr_export_bits), so that the latency-bound back end of batch k runs beside the
 * HBM-bound front end of batch k+1; front ends alternate between two sets of sliced rows.  Results and state
 * are identical to on == 0.  Ordering: inputs are read in the order of the handle's stream, as before; what
 * the exports write is ordered on sdr_result_stream(h) (record an event there, or call sdr_synchronize).
 * Every other entry point (host fetches, streaming calls, setters, sdr_seek, sdr_batch_process_static) first
 * makes the handle's stream wait for the result stream and then behaves as with on == 0, so it ends the
 * overlap for that step.  A batch behind a gating squelch is not overlapped.  Costs a second copy of the
 * sliced rows and tile summaries.  Calls on one handle stay serialised by the caller. */
SDR_API int sdr_set_overlap(sdr_handle* h, int on);
/* hipStream_t (as void*) that device-side results are ordered on: the result stream while sdr_set_overlap is
 * on, else the handle's stream. */
SDR_API void* sdr_result_stream(sdr_handle* h);
/* Diagnostics, non-blocking (hipStreamQuery / hipEventQuery; no counterpart in the reference): where the two
 * streams of a handle stand.  state[0] handle stream idle, [1] result stream idle, [2] the front-end event
 * reached, [3], [4] the two back-end events reached (1 yes, 0 not yet, -1 never recorded / no result stream),
 * [5] the row set the last batch wrote, [6] the row set whose event was recorded last, [7] the result stream
 * holds work the handle's stream has not been ordered behind.  May be called from another thread while a
 * call on the handle blocks. */
SDR_API int sdr_overlap_state(sdr_handle* h, int32_t state[8]);
SDR_API const char* sdr_last_error(sdr_handle* h);

/* ---- streaming entry points (state carried in the handle) -----------------
 * ref: gr::block::general_work of the C++ blocks + blk.work(input_items,
 * output_items) of the epy decoder (TS33C/live_decoder.grc:665-845): FIR
 * history, NCO phase, hysteresis bit, keep_one_in_n counter and the decoder's
 * rawBuffer/currentSymbols state all persist between calls.  The callback-rule
 * decoders (SDR_DEC_SOMFY .. SDR_DEC_BALDR; ref: the pigpio callbacks of
 * somfy/sniff.py:28-117, berner/sniff.py:21-44, mumbi/sniff.py:26-65) keep their
 * state, collected bits, clock and tick of the last edge (absolute microseconds
 * since stream start) too: a frame cut by a call boundary decodes as in one pass;
 * a streaming call runs the rule edge by edge on one wavefront (the batch entry
 * point keeps the 64-edges-at-a-time form).  SDR_DEC_VELUX alone treats every
 * call as one capture (the reference script decodes one array).
 * iq: n complex64 samples (interleaved re,im), host or device pointer. */
SDR_API int sdr_process_iq(sdr_handle* h, const void* iq, size_t n, int is_device_ptr, sdr_out* out);
/* ref: StreamDecoder.work([chunk]) (TS33C/stream_decoder.py:39): decoder only,
 * bits = n int8 values 0/1 at the decoder rate. */
SDR_API int sdr_process_bits(sdr_handle* h, const int8_t* bits, size_t n, int is_device_ptr, sdr_out* out);

/* Decoder-only batch: `rows` independent sliced streams of n bytes each (row r at bits + r*stride, host or
 * device), every one decoded by a fresh decoder as ONE work() call.
 * ref: TS33C/stream_decoder.py:222-235 main() with a single chunk, per recorded .dat file; used to decode
 * the stitched channel rows of a wideband capture that was channelized in segments (sdr_seek). */
SDR_API int sdr_batch_process_bits(sdr_handle* h, const int8_t* bits, size_t n, size_t stride, int rows,
                                   int is_device_ptr, sdr_out* out);

/* ---- batch entry point -----------------------------------------------------
 * n_captures independent captures of n_per_capture samples each (capture i at
 * iq + i*stride samples), every one processed from stream start by a fresh
 * chain and decoded as ONE work() call on a fresh decoder.
 * ref: BASELINE.json configs[3]; per capture == TS33C/stream_decoder.py main()
 * with a single chunk. */
SDR_API int sdr_batch_process(sdr_handle* h, const void* iq, size_t n_per_capture, size_t stride,
                              int n_captures, int is_device_ptr, sdr_out* out);

/* ---- results ---------------------------------------------------------------
 * All fetches synchronise the handle's stream first. */
SDR_API int sdr_fetch_frames(sdr_handle* h, sdr_frame* out, size_t cap, size_t* n);
/* sliced byte stream of the last call; capture c of a batch starts at
 * out + c*n_bits. */
SDR_API int sdr_fetch_bits(sdr_handle* h, int8_t* out, size_t cap, size_t* n);
/* demodulated floats of the last call (needs cfg.tap_demod); per capture
 * n_in/decim values (complex pairs when demod == SDR_DEMOD_NONE). */
SDR_API int sdr_fetch_demod(sdr_handle* h, float* out, size_t cap, size_t* n);
/* decoder state, for the Python mirror's attributes
 * (ref: TS33C/stream_decoder.py:20-32): state[0]=idle(0)/frame(1),
 * state[1]=pending symbol count, state[2]=previousSymbolType, state[3]=chunk_id. */
SDR_API int sdr_decoder_state(sdr_handle* h, int32_t state[4]);
/* The decoder object's arrays, read back on demand (the Python mirror's lazy attributes and its INFO / TRACE
 * output; all three synchronise):
 *  - sdr_fetch_pending_symbols: `currentSymbols` (ref: TS33C/stream_decoder.py:27,102-107), the widths in
 *    microseconds (float64) collected for a packet that is not complete yet;
 *  - sdr_fetch_raw_buffer: `rawBuffer` (ref: :28,44-47), the bytes held back until the buffer spans more than
 *    reset_limit;
 *  - sdr_fetch_runs: the run table of `row` from the last call that decoded: n = runs + 1 positions, run k of
 *    the decoded buffer is [out[k], out[k+1]); `edge_positions` of the reference (ref: :65-67) is out[] + 1
 *    (it indexes the buffer with the forced edge sample in front).  n = 0 when the last call decoded nothing. */
SDR_API int sdr_fetch_pending_symbols(sdr_handle* h, double* out, size_t cap, size_t* n);
SDR_API int sdr_fetch_raw_buffer(sdr_handle* h, int8_t* out, size_t cap, size_t* n);
SDR_API int sdr_fetch_runs(sdr_handle* h, int row, int32_t* out, size_t cap, size_t* n);

/* ---- timing helper ----------------------------------------------------------
 * HIP events on the stream the kernels run on: sdr_timer_start/stop bracket a
 * region, sdr_timer_elapsed_ms synchronises on the stop event. */
SDR_API int sdr_timer_start(sdr_handle* h);
SDR_API int sdr_timer_stop(sdr_handle* h);
SDR_API int sdr_timer_elapsed_ms(sdr_handle* h, float* ms);

/* Per-launch kernel timing: when enabled, every call brackets its front-end
 * (filter+demod+slice) and back-end (run-length+decode) launches with HIP
 * events on the handle's stream (ring of 256 calls, no host sync).
 * on = N > 1 times every N-th call only (a timing event is a packet of its own
 * on the queue, ~3 us: three per call cost 10 us of a 0.44-ms step).
 * sdr_profile_read synchronises and returns the last n<=cap durations. */
SDR_API int sdr_profile_enable(sdr_handle* h, int on);
SDR_API int sdr_profile_read(sdr_handle* h, float* front_ms, float* back_ms, int cap, int* n);

/* Device-side export for a multi-GPU frame gather: copies the last batch's
 * frame records (n_captures*max_frames sdr_frame, unused slots zeroed) and
 * per-capture frame counts (n_captures int32) into caller-owned DEVICE buffers
 * on the handle's stream, ready for an RCCL all_gather.  No counterpart in the
 * reference (it has no multi-device path); serves BASELINE.json configs[3]. */
SDR_API int sdr_export_frames(sdr_handle* h, void* frames_dev, size_t frames_bytes, void* counts_dev,
                              size_t counts_bytes);

/* ---- allocation-free batch entry point ----------------------------------------
 * No counterpart in the reference.  For callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc): sdr_batch_process for device-resident input with every OUTPUT in
 * caller-owned device buffers and every intermediate table (sliced rows, tile
 * summaries, run tables, pending symbols, decoder states, frame table) in one
 * file-scope device array of sdr_static_scratch_bytes() = 8 MiB per GPU (held by
 * each GPU the library runs on until the process ends).  The call performs no
 * hipMalloc / hipFree and no host synchronisation; work is queued on the handle's
 * stream.  Same kernels and results as sdr_batch_process + sdr_export_bits /
 * sdr_export_frames: bits_dev receives n_captures dense rows of out->n_bits sliced
 * bytes, frames_dev n_captures*cfg.max_frames records (unused slots zeroed),
 * counts_dev n_captures frame counts.  bits_cap in bytes, frames_cap in records.
 * Limits: plain chains only (channel filter -> MAG2/QUAD -> slicer -> decoder; no
 * prefilter, squelch, post-demod filter, channelizer, demod tap); the batch's
 * tables must fit the static array (about 60 KiB per capture of 2^18 samples at
 * decimation 50) and a row must stay below 32 KiB of sliced bytes, else
 * SDR_E_CAPACITY.  All calls on one GPU share that GPU's array, which keeps the
 * last call's contents.  Calls on one stream are ordered by the stream; a call on
 * another stream is queued behind the previous user's event on the device (one event
 * per GPU, created at sdr_create: still nothing allocated on the call path), so
 * direct calls from different handles / streams / threads serialise instead of
 * racing.  NOT covered: a stream under graph capture records no such event -- replays
 * of a captured call that overlap other users of the array on the same GPU race;
 * order them yourself.  The handle's own results (sdr_fetch_*) are left as they were. */
SDR_API int sdr_batch_process_static(sdr_handle* h, const void* iq_dev, size_t n_per_capture, size_t stride,
                                     int n_captures, int8_t* bits_dev, size_t bits_cap, sdr_frame* frames_dev,
                                     size_t frames_cap, int32_t* counts_dev, sdr_out* out);
SDR_API size_t sdr_static_scratch_bytes(void);

/* ---- one long capture in contiguous segments (BASELINE.json north_star: "each
 * GPU gets a contiguous segment with FIR-history overlap") ----------------------
 * No counterpart in the reference (GNU Radio streams one capture through one
 * thread per block); what it must reproduce is the reference's single stream:
 * the sliced bytes of all segments, concatenated, equal those of one
 * sdr_process_iq pass over the whole capture (ref: the block chain of
 * TS33C/live_decoder.grc:3292-3377 run from stream start).
 *
 * sdr_seek positions the handle's streaming state (sample/output counters, NCO
 * phase origin, keep_one_in_n counter, FIR history) as if the first abs_index
 * samples of the capture had been consumed; `halo` = the n_halo samples that
 * precede abs_index (the last sdr_history_len(h) of them are used, missing ones
 * count as the zeros before stream start).  The threshold_ff state is unknown at
 * that point: it restarts from cfg.thr_init, and sdr_segment_info afterwards says
 * which leading bytes of the NEXT sdr_process_iq call depended on it.  Decoder
 * state and rawBuffer are reset.
 * A chain with more than one memory -- a prefilter stage (mumbi/FS300.grc:3157-3242),
 * a post-demod filter (TFA/analyzer2.grc:3151-3232) -- is positioned by running it
 * over the halo from empty memories: every stage is FIR, so its state at abs_index
 * is a function of the last sdr_history_len(h) samples, which for these chains is
 * the sum of the stages' spans in input samples (a few thousand) instead of the
 * channel filter's history.  A squelch chain is refused: its power average is an
 * IIR and its gate makes output positions data dependent. */
SDR_API int sdr_history_len(sdr_handle* h);
SDR_API int sdr_seek(sdr_handle* h, uint64_t abs_index, const void* halo, size_t n_halo, int is_device_ptr);
typedef struct sdr_segment_info {
    uint64_t first_bit_index; /* position in the capture's sliced stream of the last call's first byte */
    uint64_t n_bits;          /* bytes the last call emitted */
    uint64_t n_undetermined;  /* leading bytes emitted before the call's first threshold event: their value is the
                                 state the stream had when the call began (computed here with the handle's carry-in) */
    int32_t has_event;        /* the call contained a set or clear event */
    int32_t last_state;       /* threshold_ff state after the call (given the carry-in used) */
    int32_t carry_in;         /* the carry-in the call was computed with */
    int32_t pad_;
} sdr_segment_info;
/* summary of the last sdr_process_iq call (synchronises) */
SDR_API int sdr_segment_info_get(sdr_handle* h, sdr_segment_info* info);
/* the same summary for every row of the last call (a channelizer handle: 64 channel rows that share
 * first_bit_index / n_bits and differ in the threshold fields); info may be NULL to query *n. */
SDR_API int sdr_segment_info_rows(sdr_handle* h, sdr_segment_info* info, size_t cap, size_t* n);
/* the last call's sliced bytes (capture-major, dense) into a caller-owned DEVICE
 * buffer on the handle's stream, ready for a collective; *n = bytes written. */
SDR_API int sdr_export_bits(sdr_handle* h, void* bits_dev, size_t cap_bytes, size_t* n);

/* ---- the same capture decoded WITHOUT shipping the sliced bytes (north_star: "RCCL ... used only to gather
 * decoded frames and counts").  Stream decoders only (TS33C, TFA).  What makes it exact: the reference's front end
 * (TS33C/stream_decoder.py:76-110 == TFA/stream_decoder.py:96-130) splits the symbol widths behind every width
 * above reset_limit and decodes each closed piece starting from an empty currentSymbols, so every piece that lies
 * wholly inside a segment decodes there to what the single stream gives; only the pieces that touch a segment
 * boundary need the neighbours' runs.
 *
 * sdr_segment_decode: call after sdr_seek + ONE sdr_process_iq (the segment) + sdr_segment_info_get, with the
 * threshold carry-in the exclusive scan over the segments' summaries gave this one (the leading n_undetermined
 * bytes are rewritten if it differs from what the call assumed).  It decodes the segment's bytes as a stream of
 * their own (sdr_fetch_frames / sdr_export_frames afterwards; end_pos counts from the segment's first byte) and
 * returns
 *   head[0 .. n_head)  the run starts up to and including the END of the first over-limit run,
 *   tail[0 .. n_tail)  the run starts from the end of the last over-limit run that does not touch the segment's
 *                      end, up to the segment's length,
 * both counted from the segment's first byte.  Frames with keep_from < end_pos <= keep_to are the segment's own
 * (final); the others the local pass produced belong to pieces that the head or the tail carry, and are dropped.
 * A segment without an over-limit run (or whose only one is its last run) is all head: n_tail = 0.
 * flags != 0: the head or the tail did not fit the caller's arrays, or a device table overflowed -- fall back to
 * the byte exchange (sdr_export_bits) for this capture.  Invalidates sdr_segment_info_get for the call.
 *
 * sdr_process_runs: ONE work() call of the handle's stream decoder on a stream given by its run starts (host
 * array, pos[0] = 0 < ... < pos[n_pos - 1] = length) and the levels of its first and last sample, instead of by
 * its bytes: everything the reference computes behind `edge_positions` (TS33C/stream_decoder.py:63-110) is a
 * function of those.  The joined heads and tails of all segments (sdr_amd/segments.py: stitch_tables) go here;
 * frame end positions come back in the table's coordinates. */
#define SDR_SEG_HEAD_TOO_LONG 1
#define SDR_SEG_TAIL_TOO_LONG 2
#define SDR_SEG_TABLE_OVERFLOW 4
typedef struct sdr_segment_pieces {
    int64_t n_bits;      /* the segment's sliced bytes */
    int64_t keep_from;   /* local frames with keep_from < end_pos <= keep_to are final */
    int64_t keep_to;
    int32_t n_head, n_tail;
    int32_t n_runs;      /* runs of the whole segment */
    int32_t n_frames;    /* frames of the local pass (before the keep window is applied) */
    int32_t first_level, last_level; /* the segment's first / last sliced byte */
    int32_t flags;       /* SDR_SEG_* */
    int32_t pad_;
} sdr_segment_pieces;
SDR_API int sdr_segment_decode(sdr_handle* h, int carry_in, int32_t* head, size_t head_cap, int32_t* tail, size_t tail_cap,
                               sdr_segment_pieces* info);
SDR_API int sdr_process_runs(sdr_handle* h, const int32_t* pos, size_t n_pos, int first_level, int last_level, sdr_out* out);

/* ---- 64-channel polyphase channelizer (BASELINE.json configs[4]) ----------------
 * The reference has no channelizer block; what this stands for is a bank of 64
 * freq_xlating_fir_filter_ccc(decim 64, taps, center_freq c*fs/64) blocks (the
 * block of FM-Radio/FM-Radio.grc:615-671, TFA/analyzer2.grc:1016-1072), each
 * followed by the OOK tail of TS33C/live_decoder.grc:3292-3377 (mag^2 ->
 * multiply_const -> threshold_ff -> slicer -> keep_one_in_n -> decoder).
 * cfg as for sdr_create (decim and center_freq are ignored: decimation = 64;
 * demod NONE or MAG2; keep_n / decoder_rate refer to the fs/64 channel streams).
 * The handle is used like a batch chain: sdr_set_taps (prototype low-pass, real,
 * <= 1024 taps), sdr_batch_process (whole captures from stream start),
 * sdr_fetch_*: row r = capture*64 + channel (channel c is centred at c*fs/64,
 * c >= 32 are the negative frequencies); sdr_frame.capture holds that row.
 * A handle created with max_batch 1 also STREAMS one wideband capture: sdr_process_iq
 * takes chunks of any length (the (P-1)*64+63-sample filter history, the 64-sample output
 * phase, the keep_one_in_n counter and, per channel row, the threshold bit and the
 * decoder / rule state persist), sdr_seek positions it on a segment of a long capture and
 * sdr_segment_info_rows reports the per-channel threshold summary for the stitch. */
SDR_API int sdr_pfb_create(const sdr_chain_cfg* cfg, int channels, sdr_handle** out);
/* Decode the sliced bytes of the last sdr_batch_process again with another
 * decoder (SDR_DEC_*, decoder_rate 0 = unchanged): "all sensor decoders run
 * concurrently per channel".  Replaces the frame table. */
SDR_API int sdr_redecode(sdr_handle* h, int decoder, double decoder_rate);

/* The same for up to 8 decoders in ONE launch (every row x decoder pair is its own workgroup): the frame tables
 * are kept per decoder, sdr_fetch_frames_multi(k) returns those of decoders[k].  sdr_fetch_frames still
 * returns the table of the last sdr_batch_process / sdr_redecode. */
SDR_API int sdr_redecode_multi(sdr_handle* h, const int32_t* decoders, int n, double decoder_rate);
SDR_API int sdr_fetch_frames_multi(sdr_handle* h, int k, sdr_frame* out, size_t cap, size_t* n);

/* ---- FM-Radio chain (BASELINE.json configs[1]) --------------------------------
 * ref: FM-Radio/FM-Radio.grc:2903-2964 wiring: freq_xlating_fir_filter_ccc
 * (taps 1, center_freq -shift) -> rational_resampler_ccc 1/int(samp_rate/500e3)
 * -> low_pass_filter (Blackman) -> analog_wfm_rcv (quadrature_demod ->
 * fir_filter_fff audio_decimation -> fm_deemph 75 us) -> rational_resampler_fff
 * 48/50 -> multiply_const(volume).  The flowgraph is a live stream: sdr_fm_process takes the next
 * chunk (any length), every block's state persists between calls (channel-filter histories and
 * rotator phase, the discriminator's previous sample, the audio FIR's history and decimation
 * phase, the de-emphasis delay line in double precision, the resampler's history and output
 * phase), so any chunking of a stream gives the same audio as one call; the first call after
 * sdr_fm_create / sdr_fm_reset starts at stream start. */
typedef struct sdr_fm_cfg {
    double sample_rate;      /* samp_rate of the IQ stream */
    double center_freq;      /* freq_xlating center_freq (= -shift in the GRC) */
    int32_t resamp_decim;    /* int(samp_rate/500e3) */
    int32_t audio_decim;     /* wfm_rcv audio_decimation (10) */
    double lpf_fs, lpf_fc, lpf_tw; /* low_pass_filter as the GRC declares it (2e6, 400e3, 200e3, Blackman) */
    double quad_rate;        /* wfm_rcv quad_rate as declared (500e3) */
    double tau;              /* de-emphasis time constant (75e-6) */
    int32_t audio_interp;    /* rational_resampler_fff interpolation (48) */
    int32_t audio_decimation;/* rational_resampler_fff decimation (50) */
    float volume;            /* multiply_const */
    int32_t pad_;
    uint64_t max_samples;    /* IQ samples per call */
} sdr_fm_cfg;

typedef struct sdr_fm sdr_fm;
SDR_API int sdr_fm_create(const sdr_fm_cfg* cfg, sdr_fm** out);
SDR_API int sdr_fm_destroy(sdr_fm* h);
SDR_API int sdr_fm_set_stream(sdr_fm* h, void* hip_stream);
SDR_API int sdr_fm_reset(sdr_fm* h);
/* Throughput mode of a streaming FM object (no counterpart in the reference, whose flowgraph runs its blocks as
 * concurrent threads -- this is that pipelining on one GPU): with on != 0 the back half of every sdr_fm_process
 * call (wfm_rcv's audio FIR, fm_deemph, the 48/50 resampler: ~2 M floats in a few latency-bound launches) is
 * queued on a second, library-owned stream of the same priority, so that it runs beside the channel filter of the
 * NEXT call; the next call's discriminator stage waits for it (it overwrites the floats the audio FIR reads).
 * Same kernels, same results; sdr_fm_fetch / sdr_fm_reset / sdr_fm_set_stream / sdr_fm_destroy order the object's
 * stream behind the second one first.  Work of a call may still be in flight on the second stream when
 * sdr_fm_process returns and when the object's stream is idle: synchronise through sdr_fm_fetch (n-only call). */
SDR_API int sdr_fm_set_overlap(sdr_fm* h, int on);
/* iq: n complex64 samples (host or device); n_audio receives the audio sample count. */
SDR_API int sdr_fm_process(sdr_fm* h, const void* iq, size_t n, int is_device_ptr, size_t* n_audio);
/* stage outputs of the last call (this call's new items only): 0 resampled complex (pairs), 1 discriminator,
 * 2 audio FIR, 3 de-emphasis, 4 final audio.  out may be NULL to query *n (floats). */
SDR_API int sdr_fm_fetch(sdr_fm* h, int stage, float* out, size_t cap, size_t* n);
SDR_API const char* sdr_fm_last_error(sdr_fm* h);

/* ---- somfy wake-up (preamble) finder block ----------------------------------
 * ref: somfy/epy_block_1.py:14-67 `blk(gr.sync_block)`, in_sig=[np.byte],
 * out_sig=[np.byte]; embedded in somfy/Telis1RTS.grc:892-945.  One handle = one
 * block instance (state, preamble, position persist between work() calls).
 * sdr_preamble_work == blk.work(input_items, output_items): n bytes in, n bytes
 * out (both host, or both device when is_device_ptr).  `info` returns the block
 * attributes after the call and which of the reference's print lines the call
 * produced: events bit0 "position of preamble" (value preamble_pos), bit1 the
 * state-1 "(len, sum)" line (sum8), bit2 "length of preamble" (value preamble),
 * bit3 np.gradient on fewer than 2 items (the reference raises ValueError after
 * updating its state; nothing is written to out). */
typedef struct sdr_preamble_info {
    int32_t state;
    int32_t preamble;
    int64_t position;
    int32_t events;
    int32_t sum8;         /* the chunk sum as the reference computes it (int8 arithmetic) */
    int64_t preamble_pos;
} sdr_preamble_info;
typedef struct sdr_preamble sdr_preamble;
SDR_API int sdr_preamble_create(sdr_preamble** out);
SDR_API int sdr_preamble_destroy(sdr_preamble* h);
SDR_API int sdr_preamble_work(sdr_preamble* h, const int8_t* in, size_t n, int is_device_ptr, int8_t* out,
                              sdr_preamble_info* info);
SDR_API const char* sdr_preamble_last_error(sdr_preamble* h);

SDR_API int sdr_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SDRHIP_H */
