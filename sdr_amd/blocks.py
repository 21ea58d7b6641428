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
strings, same exception types where the reference raises.  ``rawBuffer`` /
``currentSymbols`` live in GPU memory and are not exposed as arrays;
``frames`` (list of decoded records) is additive.

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


class _StreamDecoderBase(object):
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

    def debug(self, message, level=0):
        """Debug output depending on debug level."""
        if self.debug_level >= level:
            print(message)

    def close(self):
        self._chain.close()

    def work(self, input_items):
        """The actual Decoder: one chunk of 0/1 bytes; state persists."""
        chunk = np.ascontiguousarray(input_items[0]).astype(np.int8, copy=False)
        if chunk.size:
            self.currentSymbolType = int(chunk[0])
        for a in range(0, max(chunk.size, 1), self._MAX_CHUNK):
            self._chain.process_bits(chunk[a:a + self._MAX_CHUNK])
            frames = self._chain.fetch_frames()
            st = self._chain.decoder_state()
            self.state = "frame" if st[0] else "idle"
            self.previousSymbolType = st[2]
            self.chunk_id = st[3]
            for f in frames:
                self._report(f)

    def _report(self, f):
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

    def _report(self, f):
        self.pulse_short_limit = f.lim[0]
        self.gap_short_limit = f.lim[1]
        self.debug(("pulse_short_limit", self.pulse_short_limit), TRACE)
        self.debug(("gap_short_limit", self.gap_short_limit), TRACE)
        rec = {"status": _lib.STATUS_TEXT[f.status], "packet": None, "nbits": f.nbits}
        if f.status == _lib.FRAME_BAD_LENGTH:
            self.debug("Invalid packet length", ERROR)
        else:
            packet = f.packet()
            rec["packet"] = packet
            self.debug(("packet:", len(packet), [hex(b) for b in bytearray(packet)]), INFO)
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

    def _report(self, f):
        if f.status == _lib.FRAME_EXC_RESHAPE:
            raise ValueError("cannot reshape array of odd size into shape (2)")  # TFA/stream_decoder.py:137
        if f.status == _lib.FRAME_EXC_NAN:
            raise ValueError("cannot convert float NaN to integer")  # :138
        if f.status == _lib.FRAME_EXC_INDEX:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")  # :143
        rec = {"status": _lib.STATUS_TEXT[f.status], "packet": None, "nbits": f.nbits, "symbol_length": f.lim[0]}
        self.debug(("symbol_length:", f.lim[0]), TRACE)
        if f.status == _lib.FRAME_SYMBOL_LENGTH:
            self.debug("Symbol length error", ERROR)
        elif f.status == _lib.FRAME_BAD_LENGTH:
            self.debug("Invalid packet length", ERROR)
        else:
            packet = f.packet()
            rec["packet"] = packet
            if f.status == _lib.FRAME_CRC:
                self.debug("CRC-Check failed", ERROR)
            else:
                p = bytearray(packet)
                self.debug(("packet:", len(packet), [hex(b) for b in p]), INFO)
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
