# -*- coding: utf-8 -*-
"""Drop-in mirrors of the reference's decoder entry points, backed by the HIP
path (libsdrhip through sdr_amd.chain.Chain).

  TS33CStreamDecoder / ts33c_blk  <- TS33C/stream_decoder.py:13-220,
                                     TS33C/live_decoder.grc:630-845 (epy block)
  TFAStreamDecoder  / tfa_blk     <- TFA/stream_decoder.py:13-218,
                                     TFA/analyzer2.grc:765-983 (epy block)

Same constructor arguments (all defaulted, as GRC requires), same ``work``
signatures and return values, same attributes for results (``temperature``,
``humidity``, ``identifier``, ``battery_ok``, ``state``, ``chunk_id``,
``pulse_short_limit``, ``gap_short_limit``), same printed lines and debug
strings (INFO and TRACE lines included), same exception types where the reference
raises.  ``rawBuffer`` / ``currentSymbols`` live in GPU memory and are read back
when the attribute is accessed; ``frames`` (list of decoded records) is additive.

The classes need the built library and a GPU; there is no CPU fallback.
"""
from __future__ import print_function

import numpy as np

from . import _lib
from .chain import Chain

ERROR = 1
INFO = 2
TRACE = 3

try:  # subclass the real gr.sync_block when GNU Radio is importable
    from gnuradio import gr as _gr
    _SyncBase = _gr.sync_block
except Exception:  # pragma: no cover - GNU Radio is not installed in this image
    class _SyncBase(object):
        """Minimal stand-in for gr.sync_block: records name and signatures."""

        def __init__(self, name=None, in_sig=None, out_sig=None):
            self._name = name
            self.in_sig = in_sig
            self.out_sig = out_sig

        def name(self):
            return self._name


def _np1_array(pieces):
    """np.array(np.split(...)) as the numpy of the reference's time returned it (numpy >= 1.24 raises on ragged
    input): equal-length pieces -> a 2-D array (size = all elements), ragged -> an object array of arrays
    (size = number of pieces).  Display only: this is what ("Symbols:", symbols.size, symbols) shows."""
    if len(set(p.size for p in pieces)) == 1:
        a = np.array(pieces)
        return a, int(a.size)
    a = np.empty(len(pieces), dtype=object)
    for k, p in enumerate(pieces):
        a[k] = p
    return a, len(pieces)


class _StreamDecoderBase(object):
    """Front end shared by the two stream decoders (TS33C/stream_decoder.py:39-110,204-220 ==
    TFA/stream_decoder.py:59-130,201-218).

    RESULTS (frames, temperature, humidity, state, chunk_id, limits) always come from the GPU.  With
    ``debug_level >= INFO`` the object also writes what the reference writes at that level: for that it reads
    the call's run table (``edge_positions``) and the pending symbols back from the handle and replays the
    reference's piece loop ON THE HOST FOR DISPLAY ONLY -- the intermediate arrays the reference prints
    (symbols, pieces, currentSymbols, histograms, rawBits) exist nowhere else in that form.  The replay ends
    with a check against the device's state and frame records and raises if the two ever disagree.  Below INFO
    nothing of it runs.  Lines are rendered with numpy's ``legacy='1.13'`` print options, the format of the
    numpy the Python-2 reference ran under (Python 2 ``print`` of a tuple == ``str`` of it)."""
    _DECODER = None
    _MAX_CHUNK = 1 << 22

    def __init__(self, sample_rate=20000, debug_level=0):
        self.debug_level = debug_level
        self.sample_rate = sample_rate
        self.state = "idle"
        self.chunk_id = 0
        self.currentSymbolType = 0
        self.previousSymbolType = 0
        self.frames = []
        self._chain = Chain(float(sample_rate), decoder=self._DECODER, decoder_rate=float(sample_rate),
                            max_samples=self._MAX_CHUNK, max_frames=64)
        # host-side shadow of what is needed to know WHEN the device decodes (pure functions of the byte counts)
        self._raw_n = 0                       # bytes in the device's rawBuffer
        self._raw_first = 0
        self._raw_last = 0
        self._raw_dtype = np.dtype(np.uint8)  # dtype the reference's rawBuffer would have (np.append promotion)
        self._n_cur = 0                       # len(currentSymbols) after the last call
        self._warned_split = False

    # -- the reference's instance arrays (TS33C/stream_decoder.py:27-28), read back from the handle on access
    @property
    def currentSymbols(self):
        cur = self._chain.fetch_pending_symbols()
        return cur if cur.size else np.empty(0, dtype=np.uint8)

    @property
    def rawBuffer(self):
        raw = self._chain.fetch_raw_buffer()
        return raw.astype(self._raw_dtype) if raw.size else np.empty(0, dtype=np.uint8)

    def debug(self, message, level=0, int_hex=False):
        """Debug output depending on debug level.  int_hex: the line sits between the reference's
        np.set_printoptions(formatter={'int': hex}) and its reset (TS33C/stream_decoder.py:168-175)."""
        if self.debug_level >= level:
            opts = {"legacy": "1.13"}
            if int_hex:
                opts["formatter"] = {"int": hex}
            with np.printoptions(**opts):
                print(message)

    def close(self):
        self._chain.close()

    def work(self, input_items):
        """The actual Decoder: one chunk of 0/1 bytes; state persists."""
        chunk = np.ascontiguousarray(input_items[0])
        if chunk.size > self._MAX_CHUNK:
            # the device buffer holds _MAX_CHUNK bytes per call: a longer chunk is fed as several work() calls,
            # which the reference would treat as several chunks (rawBuffer is decoded and cleared per call)
            if not self._warned_split:
                import warnings
                warnings.warn("chunk of %d samples exceeds %d: processed as %d consecutive work() calls (set %s._MAX_CHUNK "
                              "before creating the decoder for one-call semantics)"
                              % (chunk.size, self._MAX_CHUNK, -(-chunk.size // self._MAX_CHUNK), type(self).__name__), RuntimeWarning)
                self._warned_split = True
            for a in range(0, chunk.size, self._MAX_CHUNK):
                self._work_one(chunk[a:a + self._MAX_CHUNK])
            return
        self._work_one(chunk)

    def _work_one(self, chunk):
        verbose = self.debug_level >= INFO
        if chunk.size:
            if self._raw_n == 0:
                self._raw_first = int(chunk[0])
            self._raw_last = int(chunk[-1])
        self._raw_dtype = np.result_type(self._raw_dtype, chunk.dtype) if chunk.size else self._raw_dtype
        total = self._raw_n + chunk.size
        decodes = total * 1e6 / self.sample_rate > self.reset_limit          # :47
        cur_before = None
        if verbose and decodes:
            cur_before = self.currentSymbols if self._n_cur else np.empty(0, dtype=np.uint8)
        state_before, prev_before = self.state, self.previousSymbolType
        self._chain.process_bits(chunk.astype(np.int8, copy=False))
        frames = self._chain.fetch_frames()
        st = self._chain.decoder_state()
        self.state = "frame" if st[0] else "idle"
        self.previousSymbolType = st[2]
        self.chunk_id = st[3]
        if not decodes:
            self._raw_n = total
            self.debug("filling rawBuffer", TRACE)                            # :220
            return
        self.currentSymbolType = self._raw_first
        if verbose:
            self._display_walk(total, cur_before, state_before, prev_before, frames, st)
        else:
            for f in frames:
                self._report(f)
        self._raw_n = 0
        self._raw_dtype = np.dtype(np.uint8)
        self._n_cur = st[1]

    def _display_walk(self, total, cur, state, prev_type, frames, st):
        """TS33C/stream_decoder.py:50-212 replayed for display (class docstring); frames are the device's."""
        pos = self._chain.fetch_runs()
        first, last = self._raw_first, self._raw_last
        any_one = bool(first) or pos.size > 2
        if not (total > 0 and (any_one or cur.size > 0)):                    # :50 / :213-215
            return
        fs, lim, dt = self.sample_rate, self.reset_limit, self._raw_dtype
        it = iter(frames)
        self.debug(("first raw sample", dt.type(first)), TRACE)
        self.debug(("last raw sample", dt.type(last)), TRACE)
        self.debug(("Chunk", self.chunk_id), TRACE)
        edge_positions = pos.astype(np.int64) + 1                            # indices into the buffer with the forced edge in front
        self.debug(("edge_positions:", edge_positions), TRACE)
        sign = 1 if first else -1
        self.debug(("edges", (sign * (1 - 2 * (np.arange(edge_positions.size) % 2))).astype(dt)), TRACE)
        if edge_positions.size > 0:
            symbols = np.diff(edge_positions) * 1e6 / fs
            pieces = np.split(symbols, np.ravel(np.where(symbols > lim)) + 1)
            shown, size = _np1_array(pieces)
            self.debug(("Symbols:", size, shown), TRACE)
            self.debug(("symbols.shape", len(pieces)), TRACE)
            self.debug(("state", state), TRACE)
            if len(pieces) > 1:
                state = "frame"
            for symbol_chunk in pieces:
                self.debug(("state", state), TRACE)
                if (size == 1) or state == "idle":
                    symbol_chunk = symbol_chunk[1::]
                if symbol_chunk.size > 0:
                    self.debug(("symbol_chunk:", symbol_chunk.size, symbol_chunk), TRACE)
                    state = "frame"
                    if (cur.size > 0) and (prev_type == first):
                        cur = cur.copy()
                        cur[-1] += symbol_chunk[0]
                        symbol_chunk = symbol_chunk[1::]
                    cur = np.append(cur, symbol_chunk)
                    if cur[0] > lim:
                        cur = cur[1::]
                    self.debug(("currentSymbols:", cur.size, cur), TRACE)
                    if (cur.size > 1) and (cur[-1] > lim):
                        f = next(it, None)
                        if f is None:
                            raise RuntimeError("display replay found a packet the device did not report")
                        self._report(f, cur)
                        cur = np.empty(0, dtype=np.uint8)
                    else:
                        self.debug("Waiting for more data", TRACE)
                else:
                    self.debug("skipping empty chunk", INFO)
                    state = "idle"
        else:
            self.debug("no edges in chunk", INFO)
        if next(it, None) is not None or (state == "frame") != bool(st[0]) or cur.size != st[1]:
            raise RuntimeError("display replay disagrees with the device (state %s/%d, pending %d/%d)"
                               % (state, st[0], cur.size, st[1]))

    def _report(self, f, symbols=None):
        raise NotImplementedError


class TS33CStreamDecoder(_StreamDecoderBase):
    """TS33C Decoder Class (TS33C/stream_decoder.py:13)."""
    _DECODER = _lib.DEC_TS33C

    def __init__(self, sample_rate=20000, debug_level=0):
        _StreamDecoderBase.__init__(self, sample_rate, debug_level)
        self.pulse_short_limit = 775
        self.gap_short_limit = 775
        self.reset_limit = 2 * self.gap_short_limit
        self.temperature = 0
        self.humidity = 0

    def _report(self, f, symbols=None):
        """One decoded packet.  `symbols` (display replay only): the packet's widths, from which the TRACE lines
        of TS33C/stream_decoder.py:112-153,174 are shown; everything reported comes from the frame record."""
        self.pulse_short_limit = f.lim[0]
        self.gap_short_limit = f.lim[1]
        packetBits = None
        if symbols is not None and self.debug_level >= TRACE:
            pulses = symbols[0::2]
            gaps = symbols[1::2]
            gaps = np.where(gaps > self.reset_limit, 0, gaps)
            pulse_histogram = np.histogram(pulses, "auto")
            gap_histogram = np.histogram(gaps[np.where(gaps > 0)], "auto")
            if (pulse_histogram[1][pulse_histogram[1].size // 2], gap_histogram[1][gap_histogram[1].size // 2]) != (f.lim[0], f.lim[1]):
                raise RuntimeError("display replay: histogram limits differ from the device's")
            self.debug(("Pulses:", pulses.size, pulses), TRACE)
            self.debug(("Pulse Histogram", pulse_histogram), TRACE)
            self.debug(("pulse_short_limit", np.float64(f.lim[0])), TRACE)
            self.debug(("Gaps:", gaps.size, gaps), TRACE)
            self.debug(("Gaps Histogram", gap_histogram), TRACE)
            self.debug(("gap_short_limit", np.float64(f.lim[1])), TRACE)
            combined = np.empty(pulses.size + gaps.size, dtype=pulses.dtype)
            combined[0::2] = pulses
            combined[1::2] = gaps
            self.debug(("combined:", combined.size, combined), TRACE)
            rawBits = np.empty(combined.size, dtype=np.uint8)
            self.debug(("rawBits.size", rawBits.size), TRACE)
            rawBits[0::2] = np.where(combined[0::2] < f.lim[0], 1, 0)
            rawBits[1::2] = np.where(combined[1::2] < f.lim[1], 255, 0)
            rawBits = rawBits[np.where(rawBits <= 1)]
            self.debug(("rawBits:", rawBits.size, rawBits), TRACE)
            if rawBits.size != f.nbits:
                raise RuntimeError("display replay: bit count differs from the device's")
            packetBits = np.extract(np.invert(np.mod(np.arange(rawBits.size) + 1, 9) == 0), rawBits) == 1
        rec = {"status": _lib.STATUS_TEXT[f.status], "packet": None, "nbits": f.nbits}
        if f.status == _lib.FRAME_BAD_LENGTH:
            self.debug("Invalid packet length", ERROR)
        else:
            packet = f.packet()
            rec["packet"] = packet
            parr = np.frombuffer(packet, np.uint8)
            self.debug(("packet:", parr.size, parr), INFO, int_hex=True)
            if packetBits is not None:
                packetParity = (np.sum(np.split(np.invert(packetBits[::-1]), np.arange(8, packetBits.size, 8)), axis=1)[::-1] % 2) == 0
                self.debug(("packetParity:", packetParity.size, packetParity), TRACE, int_hex=True)
            if f.status == _lib.FRAME_PARITY:
                self.debug("Parity check failed", ERROR)
            elif f.status == _lib.FRAME_UNKNOWN_HEADER:
                self.debug("Unknown Sensor or Header", ERROR)
            else:
                p = bytearray(packet)
                channel = (p[1] >> 5) & 0x0F
                if channel >= 5:
                    channel -= 1
                rollingCode = p[1] & 0x0F
                temp = (p[5] & 0x0F) * 100 + ((p[4] & 0xF0) >> 4) * 10 + (p[4] & 0x0F)
                sign = (((p[5] >> 7) & 0x01) * 2) - 1
                self.temperature = temp * sign / 10.
                battery_ok = (p[5] >> 6) & 0x01 == 1
                self.humidity = ((p[6] & 0xF0) >> 4) * 10 + (p[6] & 0x0F)
                assert (channel, rollingCode, int(battery_ok), self.humidity, temp * sign) == (
                    f.v[0], f.v[1], f.v[2], f.v[3], f.temp_tenths)
                self.debug(("Channel:", channel), INFO)
                self.debug(("Rolling Code:", rollingCode), INFO)
                self.debug(("Battery ok:", battery_ok), INFO)
                print("Temperature: %02.1f°C" % self.temperature)
                print("Humidity: %02i%%" % self.humidity)
                rec.update(channel=channel, rolling_code=rollingCode, battery_ok=battery_ok,
                           temperature=self.temperature, humidity=self.humidity)
        self.frames.append(rec)


class TFAStreamDecoder(_StreamDecoderBase):
    """30.3180.IT Decoder Class (TFA/stream_decoder.py:13)."""
    _DECODER = _lib.DEC_TFA

    def __init__(self, sample_rate=20000, debug_level=0):
        _StreamDecoderBase.__init__(self, sample_rate, debug_level)
        self.pulse_short_limit = 30
        self.gap_short_limit = 30
        self.reset_limit = 500
        self.identifier = 0
        self.temperature = 0
        self.humidity = 0
        self.battery_ok = False

    @classmethod
    def crc8(cls, data):
        """Simple CRC-8 calculator (poly 0x31, init 0) -- host helper kept for API
        parity; the packet CRC itself is checked on the GPU."""
        crc = 0
        for currByte in data:
            crc ^= int(currByte)
            for _ in range(8):
                crc = ((crc << 1) ^ 0x31) & 0xFF if crc & 0x80 else (crc << 1) & 0xFF
        return crc

    def _report(self, f, symbols=None):
        """One decoded packet; `symbols` (display replay only) feeds the TRACE lines of
        TFA/stream_decoder.py:133-174, everything reported comes from the frame record."""
        if f.status == _lib.FRAME_EXC_RESHAPE:
            raise ValueError("cannot reshape array of odd size into shape (2)")  # TFA/stream_decoder.py:137
        if f.status == _lib.FRAME_EXC_NAN:
            raise ValueError("cannot convert float NaN to integer")  # :138
        rec = {"status": _lib.STATUS_TEXT.get(f.status, "exception"), "packet": None, "nbits": f.nbits, "symbol_length": f.lim[0]}
        if f.status == _lib.FRAME_SYMBOL_LENGTH:
            self.debug("Symbol length error", ERROR)
            self.frames.append(rec)
            return
        symbol_length = np.float64(f.lim[0])
        self.debug(("symbol_length:", symbol_length), TRACE)
        self.debug(("baudrate:", int(1e6 / symbol_length)), TRACE)
        if f.status == _lib.FRAME_EXC_INDEX:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")  # :143
        if symbols is not None and self.debug_level >= TRACE:
            cur = symbols[6:]
            data_start = np.where(cur > symbol_length * 1.5)[0][0]
            self.debug(("data_start:", data_start), TRACE)
            pulses = cur[data_start:-1:2]
            pulses = np.where(pulses < symbol_length, 0, pulses)
            gaps = cur[data_start + 1:-1:2]
            self.debug(("Pulses:", pulses.size, pulses), TRACE)
            self.debug(("Gaps:", gaps.size, gaps), TRACE)
            combined = np.empty(pulses.size + gaps.size, dtype=pulses.dtype)
            combined[0::2] = np.floor(pulses / symbol_length)
            combined[1::2] = np.ceil(gaps / symbol_length)
            self.debug(("combined:", combined.size, combined), TRACE)
            reps = combined.astype(np.int64)
            rawBits = np.repeat((np.arange(reps.size) % 2 == 0).astype(np.uint8), reps)
            self.debug(("rawBits:", rawBits.size, rawBits), TRACE)
            if rawBits.size != f.nbits:
                raise RuntimeError("display replay: bit count differs from the device's")
        if f.status == _lib.FRAME_BAD_LENGTH:
            self.debug("Invalid packet length", ERROR)
        else:
            packet = f.packet()
            rec["packet"] = packet
            if f.status == _lib.FRAME_CRC:
                self.debug("CRC-Check failed", ERROR)
            else:
                p = bytearray(packet)
                parr = np.frombuffer(packet, np.uint8)
                self.debug(("packet:", parr.size, parr), INFO, int_hex=True)
                self.identifier = "{:02X}".format((p[2] << 8) + p[3])
                self.temperature = (p[4] & 0x0f) * 10 + ((p[5] >> 4) & 0x0f) * 1 + (p[5] & 0x0f) * 0.1 - 40
                self.humidity = (p[6] & 0x7f)
                self.battery_ok = (p[7] >> 7) & 0x01 == 0
                assert ((p[2] << 8) + p[3], self.humidity, int(self.battery_ok)) == (f.v[0], f.v[1], f.v[2])
                self.debug(("Identifier:", self.identifier), INFO)
                self.debug(("Battery ok:", self.battery_ok), INFO)
                print("Temperature: %02.1f°C" % self.temperature)
                print("Humidity: %02i%%" % self.humidity)
                rec.update(identifier=self.identifier, temperature=self.temperature, humidity=self.humidity,
                           battery_ok=self.battery_ok)
        self.frames.append(rec)


class _TFAGrcDecoder(TFAStreamDecoder):
    _DECODER = _lib.DEC_TFA_GRC


def _make_blk(decoder_cls, doc):
    class blk(_SyncBase):  # other base classes are basic_block, decim_block, interp_block
        def __init__(self, sample_rate=20000, debug_level=0):  # only default arguments here
            _SyncBase.__init__(self, name='Decoder Block', in_sig=[np.byte],
                               out_sig=[np.float32, np.float32])
            # attributes named like parameters become GRC callbacks
            self.debug_level = debug_level
            self.sample_rate = sample_rate
            self._dec = decoder_cls(sample_rate=sample_rate, debug_level=debug_level)

        def __getattr__(self, name):  # temperature, humidity, state, ... live on the decoder
            if name.startswith("_"):
                raise AttributeError(name)
            return getattr(self._dec, name)

        def work(self, input_items, output_items):
            """The actual Decoder: fills both outputs with the last decoded
            temperature / humidity (TS33C/live_decoder.grc:843-845)."""
            self._dec.debug_level = self.debug_level
            self._dec.work(input_items)
            output_items[0][:] = self._dec.temperature
            output_items[1][:] = self._dec.humidity
            return len(output_items[0])

    blk.__doc__ = doc
    return blk


ts33c_blk = _make_blk(TS33CStreamDecoder, "Embedded decoder block of TS33C/live_decoder.grc:630-845 on the GPU path")
tfa_blk = _make_blk(_TFAGrcDecoder, "Embedded decoder block of TFA/analyzer2.grc:765-983 on the GPU path")


def somfy_decode(manchester, clock_us=640, rate=1000000):
    """somfy/decoder.py:27-58 on the GPU path: 112 Manchester half-symbols ->
    payload, de-obfuscated frame, checksum, fields.  The half-symbols are laid
    out as the on-air envelope (hw-sync x2, sw-sync, half-symbols at `clock_us`,
    silence) and decoded by the somfy rule kernel (somfy/sniff.py:28-117)."""
    from . import synth
    m = [int(v) for v in np.asarray(manchester).ravel()]
    if len(m) != 112:
        raise ValueError("expected 112 half-symbols")
    sym = [(1, 2500.0), (0, 2500.0)] * 2 + [(1, 4850.0), (0, float(clock_us))]
    sym += [(lvl, float(clock_us)) for lvl in m]
    sym += [(0, 32000.0)]
    env = synth.envelope(synth.merge_symbols(sym), rate, lead_in_us=3000.0).astype(np.int8)
    with Chain(float(rate), decoder=_lib.DEC_SOMFY, decoder_rate=float(rate), max_samples=env.size + 64) as c:
        c.process_bits(env)
        frames = c.fetch_frames()
    if len(frames) != 1:
        raise RuntimeError("somfy frame not found (%d frames)" % len(frames))
    f = frames[0]
    return {"payload": bytes(bytearray(f.raw[:7])), "frame": f.packet(), "checksum_ok": f.status == _lib.FRAME_OK,
            "key": f.bytes[0], "control": f.v[0], "rolling_code": f.v[1], "address": f.v[2]}
