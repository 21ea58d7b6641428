# -*- coding: utf-8 -*-
"""numpy restatement of the reference's own Python decoders -- TEST INFRASTRUCTURE.

The reference decoders are Python 2 (``print message``, integer ``/``) and
cannot be imported under this image's Python 3.10; this file follows them
statement by statement with the three changes SURVEY.md section 8c lists:
``//`` for the histogram index, explicit ragged-``np.split`` handling with the
2017-numpy ``.size``/``.shape[0]`` semantics, Python-int field maths (numpy-1
scalar promotion).  ``np.histogram(..., "auto")`` is numpy 2.2.6's.  One place
where that choice matters: widths whose inter-quartile range is 0 (a packet of
mostly one symbol).  numpy >= 1.15 then bins by the Sturges width alone; the
numpy the Python-2 reference ran (< 1.15) made ONE bin, so ``edges[size // 2]``
was the longest width.  The reference holds no vector with IQR == 0, so that
case is PARITY UNPINNED; the modern behaviour is what is restated
(tests/test_host_decode.py::test_hist_auto_with_zero_iqr_is_numpy_2).

Pinned by the reference's known-answer vectors (tests/golden/kat.json,
tests/test_oracle_kat.py): TS33C README pulses/gaps -> packet/fields, the two
``docs/*.png`` bit tables, TFA tfrec line, somfy frames, mumbi/berner codes.

Every class records what the reference would ``print`` in ``.prints``, its
debug() calls (every one the reference makes, TRACE lines included) in ``.log``
and the text of both, line by line as the reference's stdout had them, in
``.stdout`` -- instead of writing to stdout; decoded packets are also appended
to ``.frames`` (additive, for comparisons).
"""
import math
from struct import unpack

import numpy as np

ERROR = 1
INFO = 2
TRACE = 3


class _StreamFrontEnd(object):
    """Shared front end: TS33C/stream_decoder.py:39-110,204-220 ==
    TFA/stream_decoder.py:59-130,201-218 (identical text, other constants)."""

    def __init__(self, sample_rate, debug_level):
        self.debug_level = debug_level
        self.sample_rate = sample_rate
        self.state = "idle"
        self.chunk_id = 0
        self.currentSymbols = np.empty(0, dtype=np.uint8)
        self.rawBuffer = np.empty(0, dtype=np.uint8)
        self.currentSymbolType = 0
        self.previousSymbolType = 0
        self.prints = []
        self.log = []
        self.stdout = []      # every line the reference would have written, debug output and prints, in order
        self.frames = []

    def debug(self, message, level=0, int_hex=False):
        """records (level, message) -- and the text Python 2 `print message` wrote for it: str of a string, str of
        a tuple (= repr of its items) under the array print format of the numpy of the reference's time
        (legacy='1.13'); int_hex: between np.set_printoptions(formatter={'int': hex}) and its reset
        (TS33C/stream_decoder.py:168-175)."""
        if self.debug_level >= level:
            self.log.append((level, message))
            opts = {"legacy": "1.13"}
            if int_hex:
                opts["formatter"] = {"int": hex}
            with np.printoptions(**opts):
                self.stdout.append(str(message))

    def _print(self, text):
        self.prints.append(text)
        self.stdout.append(text)

    @staticmethod
    def _array_like_numpy1(pieces):
        """np.array(list of arrays) as numpy < 1.24 built it: 2-D when the pieces are equally long, else an
        object array of arrays (what ("Symbols:", symbols.size, symbols) printed)."""
        if len(set(p.size for p in pieces)) == 1:
            return np.array(pieces)
        a = np.empty(len(pieces), dtype=object)
        for k, p in enumerate(pieces):
            a[k] = p
        return a

    @staticmethod
    def _split_like_numpy1(symbols, reset_limit):
        """np.array(np.split(symbols, where(symbols > limit)+1)) as numpy<1.24
        returned it: (pieces, size, shape0).  One piece -> 2-D (1, n): size n.
        Several pieces of equal length L -> 2-D: size = count*L.  Ragged ->
        object array: size = count."""
        pieces = np.split(symbols, np.ravel(np.where(symbols > reset_limit)) + 1)
        lens = set(p.size for p in pieces)
        if len(lens) == 1:
            size = len(pieces) * pieces[0].size
        else:
            size = len(pieces)
        return pieces, size, len(pieces)

    def work(self, input_items):
        rawChunk = input_items[0]
        self.rawBuffer = np.append(self.rawBuffer, rawChunk)

        if self.rawBuffer.size * 1e6 / self.sample_rate > self.reset_limit:
            if (self.rawBuffer.size > 0) and ((np.sum(self.rawBuffer) > 0) or (self.currentSymbols.size > 0)):
                self.currentSymbolType = self.rawBuffer[0]
                self.debug(("first raw sample", self.rawBuffer[0]), TRACE)
                self.debug(("last raw sample", self.rawBuffer[-1]), TRACE)
                self.rawBuffer = np.insert(self.rawBuffer, 0, [1 - self.rawBuffer[0]])
                self.rawBuffer = np.append(self.rawBuffer, [1 - self.rawBuffer[-1]])
                self.chunk_id += 1
                self.debug(("Chunk", self.chunk_id), TRACE)

                edges = np.insert(np.diff(self.rawBuffer), 0, [0])
                edge_positions = np.ravel(np.nonzero(edges))
                self.debug(("edge_positions:", edge_positions), TRACE)
                self.debug(("edges", edges[np.nonzero(edges)]), TRACE)

                if edge_positions.size > 0:
                    symbols = np.diff(edge_positions) * 1e6 / self.sample_rate
                    pieces, symbols_size, symbols_shape0 = self._split_like_numpy1(symbols, self.reset_limit)
                    self.debug(("Symbols:", symbols_size, self._array_like_numpy1(pieces)), TRACE)
                    self.debug(("symbols.shape", symbols_shape0), TRACE)
                    self.debug(("state", self.state), TRACE)

                    if symbols_shape0 > 1:
                        self.state = "frame"

                    piece_end = 0   # additive: where each piece ends, for frame["end_pos"]
                    for symbol_chunk in pieces:
                        piece_end += symbol_chunk.size
                        self.debug(("state", self.state), TRACE)
                        if (symbols_size == 1) or self.state == "idle":
                            symbol_chunk = symbol_chunk[1::]
                        if symbol_chunk.size > 0:
                            self.debug(("symbol_chunk:", symbol_chunk.size, symbol_chunk), TRACE)
                            self.state = "frame"
                            if (self.currentSymbols.size > 0) and (self.previousSymbolType == self.currentSymbolType):
                                self.currentSymbols[-1] += symbol_chunk[0]
                                symbol_chunk = symbol_chunk[1::]

                            self.currentSymbols = np.append(self.currentSymbols, symbol_chunk)

                            if self.currentSymbols[0] > self.reset_limit:
                                self.currentSymbols = self.currentSymbols[1::]
                            self.debug(("currentSymbols:", self.currentSymbols.size, self.currentSymbols), TRACE)

                            if (self.currentSymbols.size > 1) and (self.currentSymbols[-1] > self.reset_limit):
                                n_before = len(self.frames)
                                self._decode_packet()
                                for f in self.frames[n_before:]:
                                    # index in this call's rawBuffer (before the two forced samples) behind the
                                    # over-limit width that closed the packet
                                    f["end_pos"] = int(edge_positions[piece_end]) - 1
                                self.currentSymbols = np.empty(0, dtype=np.uint8)
                            else:
                                self.debug("Waiting for more data", TRACE)
                        else:
                            self.debug("skipping empty chunk", INFO)
                            self.state = "idle"
                else:
                    self.debug("no edges in chunk", INFO)
                self.previousSymbolType = self.rawBuffer[-2]
            else:
                self.state = "idle"
            self.rawBuffer = np.empty(0, dtype=np.uint8)
        else:
            self.debug("filling rawBuffer", TRACE)


class TS33CStreamDecoder(_StreamFrontEnd):
    """TS33C/stream_decoder.py:13-220 (GRC twin TS33C/live_decoder.grc:630-845)."""

    def __init__(self, sample_rate=20000, debug_level=0):
        _StreamFrontEnd.__init__(self, sample_rate, debug_level)
        self.pulse_short_limit = 775
        self.gap_short_limit = 775
        self.reset_limit = 2 * self.gap_short_limit
        self.temperature = 0
        self.humidity = 0

    def _decode_packet(self):
        # TS33C/stream_decoder.py:112-202
        pulses = self.currentSymbols[0::2]
        gaps = self.currentSymbols[1::2]
        gaps = np.where(gaps > self.reset_limit, 0, gaps)
        pulse_histogram = np.histogram(pulses, "auto")
        gap_histogram = np.histogram(gaps[np.where(gaps > 0)], "auto")
        self.pulse_short_limit = pulse_histogram[1][pulse_histogram[1].size // 2]
        self.gap_short_limit = gap_histogram[1][gap_histogram[1].size // 2]
        self.debug(("Pulses:", pulses.size, pulses), TRACE)
        self.debug(("Pulse Histogram", pulse_histogram), TRACE)
        self.debug(("pulse_short_limit", self.pulse_short_limit), TRACE)
        self.debug(("Gaps:", gaps.size, gaps), TRACE)
        self.debug(("Gaps Histogram", gap_histogram), TRACE)
        self.debug(("gap_short_limit", self.gap_short_limit), TRACE)

        combined = np.empty(pulses.size + gaps.size, dtype=pulses.dtype)
        combined[0::2] = pulses
        combined[1::2] = gaps
        self.debug(("combined:", combined.size, combined), TRACE)

        rawBits = np.empty(combined.size, dtype=np.uint8)
        self.debug(("rawBits.size", rawBits.size), TRACE)
        rawBits[0::2] = np.where(combined[0::2] < self.pulse_short_limit, 1, 0)
        rawBits[1::2] = np.where(combined[1::2] < self.gap_short_limit, 255, 0)
        rawBits = rawBits[np.where(rawBits <= 1)]
        self.debug(("rawBits:", rawBits.size, rawBits), TRACE)

        parityMask = np.mod(np.arange(rawBits.size) + 1, 9) == 0
        parityBits = np.extract(parityMask, rawBits) == 1
        packetBits = np.extract(np.invert(parityMask), rawBits) == 1

        frame = {"status": None, "packet": None, "pulse_short_limit": float(self.pulse_short_limit),
                 "gap_short_limit": float(self.gap_short_limit), "nbits": int(rawBits.size)}
        if (packetBits.size % 8) == 0:
            packet = np.packbits(np.invert(packetBits[::-1]))[::-1]
            frame["packet"] = bytes(bytearray(packet.tolist()))
            self.debug(("packet:", packet.size, packet), INFO, int_hex=True)
            packetParity = (np.sum(np.split(np.invert(packetBits[::-1]), np.arange(8, packetBits.size, 8)), axis=1)[::-1] % 2) == 0
            self.debug(("packetParity:", packetParity.size, packetParity), TRACE, int_hex=True)
            if np.array_equal(packetParity, parityBits):
                if (packet.size == 10) and (packet[0] == 0x9f):
                    p = [int(b) for b in packet]
                    channel = (p[1] >> 5) & 0x0F
                    if channel >= 5:
                        channel -= 1
                    rollingCode = p[1] & 0x0F
                    temp = (p[5] & 0x0F) * 100 + ((p[4] & 0xF0) >> 4) * 10 + (p[4] & 0x0F)
                    sign = (((p[5] >> 7) & 0x01) * 2) - 1
                    self.temperature = temp * sign / 10.
                    battery_ok = (p[5] >> 6) & 0x01 == 1
                    self.humidity = ((p[6] & 0xF0) >> 4) * 10 + (p[6] & 0x0F)
                    self.debug(("Channel:", channel), INFO)
                    self.debug(("Rolling Code:", rollingCode), INFO)
                    self.debug(("Battery ok:", battery_ok), INFO)
                    self._print("Temperature: %02.1f°C" % self.temperature)
                    self._print("Humidity: %02i%%" % self.humidity)
                    frame.update(status="ok", channel=channel, rolling_code=rollingCode,
                                 battery_ok=battery_ok, temperature=self.temperature,
                                 humidity=self.humidity)
                else:
                    self.debug("Unknown Sensor or Header", ERROR)
                    frame["status"] = "Unknown Sensor or Header"
            else:
                self.debug("Parity check failed", ERROR)
                frame["status"] = "Parity check failed"
        else:
            self.debug("Invalid packet length", ERROR)
            frame["status"] = "Invalid packet length"
        self.frames.append(frame)


def crc8(data):
    """TFA/stream_decoder.py:44-57: CRC-8 poly 0x31, init 0, MSB first."""
    generator = 0x31
    crc = 0
    for currByte in data:
        crc ^= int(currByte)
        for _ in range(8):
            if (crc & 0x80) != 0:
                crc = ((crc << 1) ^ generator) & 0xFF
            else:
                crc <<= 1
    return crc


class TFAStreamDecoder(_StreamFrontEnd):
    """TFA/stream_decoder.py:13-218; ``nan_guard=True`` gives the GRC twin
    TFA/analyzer2.grc:765-983 ("Symbol length error" on an empty sync)."""

    def __init__(self, sample_rate=20000, debug_level=0, nan_guard=False):
        _StreamFrontEnd.__init__(self, sample_rate, debug_level)
        self.pulse_short_limit = 30
        self.gap_short_limit = 30
        self.reset_limit = 500
        self.nan_guard = nan_guard
        self.identifier = 0
        self.temperature = 0
        self.humidity = 0
        self.battery_ok = False

    crc8 = staticmethod(crc8)

    def _decode_packet(self):
        # TFA/stream_decoder.py:132-199
        self.currentSymbols = self.currentSymbols[6:]
        sync = self.currentSymbols[:2 * 75]
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                symbol_length = np.mean(np.sum(np.reshape(sync, (-1, 2)), axis=1))
        frame = {"status": None, "packet": None, "symbol_length": float(symbol_length)}
        if self.nan_guard and math.isnan(symbol_length):
            self.debug("Symbol length error", ERROR)
            frame["status"] = "Symbol length error"
            self.frames.append(frame)
            return
        baudrate = int(1e6 / symbol_length)  # raises on NaN like the reference
        self.debug(("symbol_length:", symbol_length), TRACE)
        self.debug(("baudrate:", baudrate), TRACE)
        data_start = np.where(self.currentSymbols > symbol_length * 1.5)[0][0]
        self.debug(("data_start:", data_start), TRACE)
        pulses = self.currentSymbols[data_start:-1:2]
        pulses = np.where(pulses < symbol_length, 0, pulses)
        gaps = self.currentSymbols[data_start + 1:-1:2]
        self.debug(("Pulses:", pulses.size, pulses), TRACE)
        self.debug(("Gaps:", gaps.size, gaps), TRACE)
        combined = np.empty(pulses.size + gaps.size, dtype=pulses.dtype)
        combined[0::2] = np.floor(pulses / symbol_length)
        combined[1::2] = np.ceil(gaps / symbol_length)
        self.debug(("combined:", combined.size, combined), TRACE)

        rawBits = np.empty(0, dtype=np.uint8)
        for index, symbol in np.ndenumerate(combined):
            if symbol:
                if not index[0] % 2:
                    rawBits = np.append(rawBits, np.ones(int(symbol), dtype=np.uint8))
                else:
                    rawBits = np.append(rawBits, np.zeros(int(symbol), dtype=np.uint8))
        self.debug(("rawBits:", rawBits.size, rawBits), TRACE)
        frame["nbits"] = int(rawBits.size)
        if rawBits.size >= 11 * 8:
            packet = np.packbits(rawBits[:11 * 8][::-1])[::-1]
            frame["packet"] = bytes(bytearray(packet.tolist()))
            p = [int(b) for b in packet]
            if crc8(p[2:10]) == p[10]:
                self.debug(("packet:", packet.size, packet), INFO, int_hex=True)
                self.identifier = "{:02X}".format((p[2] << 8) + p[3])
                self.temperature = (p[4] & 0x0f) * 10 + ((p[5] >> 4) & 0x0f) * 1 + (p[5] & 0x0f) * 0.1 - 40
                self.humidity = (p[6] & 0x7f)
                self.battery_ok = (p[7] >> 7) & 0x01 == 0
                self.debug(("Identifier:", self.identifier), INFO)
                self.debug(("Battery ok:", self.battery_ok), INFO)
                self._print("Temperature: %02.1f°C" % self.temperature)
                self._print("Humidity: %02i%%" % self.humidity)
                frame.update(status="ok", identifier=self.identifier, temperature=self.temperature,
                             humidity=self.humidity, battery_ok=self.battery_ok)
            else:
                self.debug("CRC-Check failed", ERROR)
                frame["status"] = "CRC-Check failed"
        else:
            self.debug("Invalid packet length", ERROR)
            frame["status"] = "Invalid packet length"
        self.frames.append(frame)


# --------------------------------------------------------------------- somfy
SOMFY_COMMANDS = {0x00: 'null', 0x01: 'stop', 0x02: 'up', 0x04: 'down', 0x08: 'prog'}


def somfy_decode(manchester):
    """somfy/decoder.py:27-65 on a 112-element half-symbol array."""
    manchester = np.asarray(manchester)
    bits = np.ravel(np.where(np.reshape(manchester, (-1, 2)) == [0, 1], 1, 0))[::2]
    payload = np.packbits(bits)
    frame = np.zeros(7, dtype=np.uint8)
    frame[0] = payload[0]
    for i in range(1, 7):
        frame[i] = payload[i] ^ payload[i - 1]
    cksum = 0
    for i in range(0, 7):
        cksum = cksum ^ int(frame[i]) ^ (int(frame[i]) >> 4)
    cksum = cksum & 0xf
    rolling_code = unpack(">H", frame[2:4].tobytes())[0]
    address = unpack("<I", frame[4:7].tobytes() + b"\x00")[0]
    control = (int(frame[1]) >> 4) & 0x0f
    return {"payload": bytes(bytearray(payload.tolist())), "frame": bytes(bytearray(frame.tolist())),
            "checksum_ok": cksum == 0, "key": int(frame[0]), "control": control,
            "rolling_code": rolling_code, "address": address}


class SomfySniffer(object):
    """somfy/sniff.py:19-119 callback rule on (level, tick) edge events.
    level 1 = end of gap, 0 = end of pulse, 2 = watchdog (20 ms, :167)."""

    def __init__(self):
        self.start_tick = 0
        self.state = 0
        self.bits = np.empty(0, dtype=np.uint8)
        self.hw_sync = np.empty(0)
        self.tolerance = 100
        self.clock = 640
        self.frames = []

    def cbf(self, level, tick):
        tol = self.tolerance
        if level == 1:
            delta = (tick - self.start_tick) & 0xFFFFFFFF
            self.start_tick = tick
            if (0 <= self.state <= 2) and (delta in range(2500 - 2 * tol, 2500 + 2 * tol)):
                if self.state < 2:
                    self.hw_sync = np.empty(0)
                self.hw_sync = np.append(self.hw_sync, [delta])
                self.state = 2
            elif (self.state == 3) and (delta in range(2 * self.clock - tol, 2 * self.clock + tol)):
                self.bits = np.append(self.bits, [0, 0])
            elif (self.state == 3) and (delta in range(self.clock - tol, self.clock + tol)):
                self.bits = np.append(self.bits, [0])
        elif level == 0:
            delta = (tick - self.start_tick) & 0xFFFFFFFF
            self.start_tick = tick
            if (self.state == 0) and (delta in range(10050 - tol, 10050 + tol)):
                self.state = 1
            elif (0 <= self.state <= 2) and (delta in range(2500 - 2 * tol, 2500 + 2 * tol)):
                if self.state < 2:
                    self.hw_sync = np.empty(0)
                self.hw_sync = np.append(self.hw_sync, [delta])
                self.state = 2
            elif (self.state == 2) and (delta in range(4850 - 2 * tol, 4850 + 2 * tol)):
                self.clock = int(np.average(self.hw_sync) / 4)
                self.bits = np.empty(0, dtype=np.uint8)
                self.state = 3
            elif (self.state == 3) and (delta in range(2 * self.clock - tol, 2 * self.clock + tol)):
                self.bits = np.append(self.bits, [1, 1])
            elif (self.state == 3) and (delta in range(self.clock - tol, self.clock + tol)):
                self.bits = np.append(self.bits, [1])
        elif (level == 2) and (self.state > 0):
            bits = self.bits[1::]
            if bits.size < 112:
                bits = np.append(bits, [0])
            if bits.size == 112:
                decoded = np.ravel(np.where(np.reshape(bits, (-1, 2)) == [0, 1], 1, 0))[::2]
                frame = np.packbits(decoded.astype(np.uint8))
                raw = bytes(bytearray(frame.tolist()))
                for i in range(frame.size - 1, 0, -1):
                    frame[i] = frame[i] ^ frame[i - 1]
                cksum = int(frame[0]) ^ (int(frame[0]) >> 4)
                for i in range(1, 7):
                    cksum = cksum ^ int(frame[i]) ^ (int(frame[i]) >> 4)
                cksum = cksum & 0x0f
                self.frames.append({"raw": raw, "frame": bytes(bytearray(frame.tolist())),
                                    "checksum_ok": cksum == 0, "clock": self.clock,
                                    "control": (int(frame[1]) >> 4) & 0x0f,
                                    "rolling_code": (int(frame[2]) << 8) | int(frame[3]),
                                    "address": int(frame[4]) | (int(frame[5]) << 8) | (int(frame[6]) << 16)})
            self.bits = np.empty(0, dtype=np.uint8)
            self.state = 0


def _wrap8(v):
    """two's-complement int8 of a Python int"""
    return ((int(v) + 128) & 0xFF) - 128


class SomfyPreambleBlock(object):
    """somfy/epy_block_1.py:14-67, the arithmetic of work() spelled out.

    The reference adds up the chunk with the builtin sum() over np.int8 items:
    numpy scalar arithmetic, so the total wraps modulo 256 into int8 (numpy 1.x and
    2.x alike), and `preamble` (an np.int8 after its first assignment) wraps the
    same way.  `position - preamble` widens under the value-based casting of the
    numpy 1.x the reference was written for; that is what is restated (numpy 2
    would raise OverflowError for positions beyond 127).  np.gradient(x, 0.5) of a
    byte chunk is, cast back to bytes: 2*(x[1]-x[0]), x[i+1]-x[i-1], 2*(x[-1]-x[-2]),
    and raises ValueError for chunks shorter than 2."""

    def __init__(self, bit_length=1e-6):
        self.bit_length = bit_length
        self.state = 0
        self.preamble = 0
        self.position = 0
        self.prints = [("length of one sample", bit_length)]

    def work(self, input_items, output_items):
        x = np.asarray(input_items[0]).astype(np.int64)
        out = output_items[0]
        n = x.size
        s8 = _wrap8(x.sum())
        self.position += n
        if self.state == 0:
            if s8 > 0:
                self.preamble = s8
                self.prints.append(("position of preamble", self.position - self.preamble))
                self.state = 1
                out[:] = 0
                out[int(np.flatnonzero(x)[0])] = 1
            else:
                out[:] = 0
        elif self.state == 1:
            self.prints.append((n, s8))
            self.preamble = _wrap8(self.preamble + s8)
            if s8 < n:
                self.state = 2
                self.prints.append(("length of preamble", self.preamble))
                if n < 2:
                    raise ValueError("Shape of array too small to calculate a numerical gradient, "
                                     "at least (edge_order + 1) elements are required.")
                g = np.empty(n, np.int64)
                g[1:-1] = x[2:] - x[:-2]
                g[0] = 2 * (x[1] - x[0])
                g[-1] = 2 * (x[-1] - x[-2])
                out[:] = g.astype(np.int8)
            else:
                out[:] = 1
        else:
            out[:] = 1
        return len(out)


# -------------------------------------------------------------------- berner
class BernerSniffer(object):
    """berner/sniff.py:17-44 callback rule."""

    def __init__(self):
        self.start_tick = 0
        self.state = 0
        self.bits = None
        self.frames = []

    def cbf(self, level, tick):
        if level == 1:
            self.start_tick = tick
        elif level == 0:
            delta = (tick - self.start_tick) & 0xFFFFFFFF
            if (self.state == 0) and (delta > 5300) and (delta < 5700):
                self.bits = np.empty(0, dtype=np.uint8)
                self.state = 1
            elif (self.state == 1) and (delta < 500):
                self.bits = np.append(self.bits, [0])
            elif (self.state == 1) and (delta > 500) and (delta < 1000):
                self.bits = np.append(self.bits, [1])
            elif self.state == 1:
                if self.bits.size >= 32:
                    self.frames.append(bytes(bytearray(np.packbits(self.bits.astype(np.uint8))[:4].tolist())))
                self.state = 0


# --------------------------------------------------------------------- mumbi
class MumbiSniffer(object):
    """mumbi/sniff.py:19-65 callback rule (watchdog 8 ms, :112)."""

    def __init__(self):
        self.start_tick = 0
        self.state = 0
        self.bits = np.empty(0, dtype=np.uint8)
        self.tolerance = 50
        self.frames = []

    def cbf(self, level, tick):
        tol = self.tolerance
        if level == 1:
            self.start_tick = tick
        elif level == 0:
            delta = (tick - self.start_tick) & 0xFFFFFFFF
            if delta in range(820 - 2 * tol, 820 + 2 * tol):
                self.bits = np.append(self.bits, [1])
                self.state = 1
            elif delta in range(300 - tol, 300 + tol):
                self.bits = np.append(self.bits, [0])
        elif (level == 2) and (self.state > 0):
            if self.bits.size == 34:
                self.frames.append(bytes(bytearray(np.packbits(self.bits.astype(np.uint8)).tolist())))
            self.bits = np.empty(0, dtype=np.uint8)
            self.state = 0


# ------------------------------------------------------------------ kwmobile
class KwmobileDecoder(object):
    """kwmobile/decoder.py:58-135 callback rule + decode (watchdog 3 ms, :145).
    Integer field maths as numpy 1.x promoted it (Python ints)."""

    def __init__(self):
        self.start_tick = 0
        self.symbol_tolerance = 50
        self.state = 0
        self.symbols = np.empty(0, dtype=np.uint8)
        self.frames = []

    def cbf(self, level, tick):
        tol = self.symbol_tolerance
        if level == 1:
            delta = (tick - self.start_tick) & 0xFFFFFFFF
            self.start_tick = tick
            if self.state == 0 and delta in range(4500 - 4 * tol, 4500 + 4 * tol):
                self.state = 1
            elif (self.state == 1) and delta in range(2500 - 2 * tol, 2500 + 2 * tol):
                self.symbols = np.append(self.symbols, [1])
            elif (self.state == 1) and delta in range(1500 - 2 * tol, 1500 + 2 * tol):
                self.symbols = np.append(self.symbols, [0])
        elif (level == 2) and (self.state > 0):
            if self.symbols.size == 36:
                self.frames.append(kwmobile_decode(self.symbols))
            self.symbols = np.empty(0, dtype=np.uint8)
            self.state = 0


def kwmobile_decode(symbols):
    """kwmobile/decoder.py:126-134 on 36 symbols."""
    frame = [int(v) for v in np.packbits(np.asarray(symbols, np.uint8))]
    return dict(frame=bytes(bytearray(frame)),
                temperature=float(((frame[1] & 0x0f) << 8 | frame[2]) / 10.),
                humidity=int(((frame[3] & 0x0f) << 4) + (frame[4] >> 4)),
                channel=int((frame[1] & 0x30) >> 4),
                battery_ok=int(frame[1] & 0x80) == 0x80,
                sensor_id=int(frame[0]))


# --------------------------------------------------------------------- baldr
class BaldrDecoder(object):
    """baldr/decoder.py:53-142 callback rule + decode (watchdog 18 ms, :152)."""

    def __init__(self):
        self.start_tick = 0
        self.pulse_medium_gap = 2650
        self.pulse_long_gap = 4700
        self.pulse_verylong_gap = 9600
        self.symbol_tolerance = 200
        self.state = 0
        self.symbols = np.empty(0, dtype=np.uint8)
        self.frames = []

    def cbf(self, level, tick):
        tol = self.symbol_tolerance
        if level == 1:
            delta = (tick - self.start_tick) & 0xFFFFFFFF
            self.start_tick = tick
            if self.state == 0 and delta in range(self.pulse_verylong_gap - tol, self.pulse_verylong_gap + tol):
                self.state = 1
            elif (self.state == 1) and delta in range(self.pulse_long_gap - tol, self.pulse_long_gap + tol):
                self.symbols = np.append(self.symbols, [1])
            elif (self.state == 1) and delta in range(self.pulse_medium_gap - tol, self.pulse_medium_gap + tol):
                self.symbols = np.append(self.symbols, [0])
            elif (self.state == 1) and delta in range(self.pulse_verylong_gap - tol, self.pulse_verylong_gap + tol):
                if self.symbols.size == 37:
                    self.state = 2
                else:
                    self.symbols = np.empty(0, dtype=np.uint8)
                    self.state = 0
        elif (level == 2) and (self.state > 0):
            if self.symbols.size == 37:
                self.frames.append(baldr_decode(self.symbols))
            self.symbols = np.empty(0, dtype=np.uint8)
            self.state = 0


def baldr_decode(symbols):
    """baldr/decoder.py:126-141 on 37 symbols."""
    frame = [int(v) for v in np.packbits(np.asarray(symbols, np.uint8))]
    temp_raw = (frame[2] << 4) | (frame[3] >> 4)
    if temp_raw > 0x7ff:
        temp_raw -= 2 ** 12
    return dict(frame=bytes(bytearray(frame)),
                temperature=float(temp_raw) / 10.,
                humidity=int(((frame[3] & 0x07) << 4) | (frame[4] >> 4)),
                channel=int(frame[1] & 0x03),
                txMode=int(frame[1] & 0x04),
                sensor_id=int(((frame[0] & 0x0f) << 4) | (frame[1] >> 4)))


# --------------------------------------------------------------------- velux
VELUX_SOF = [0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1]  # "0 11111111 1 0 11001100 1" (velux/decoder.py:31)


def crc16_kermit(data):
    """Crc16.KERMIT of the `crc` package (velux/decoder.py:64): poly 0x1021
    reflected (0x8408), init 0, no final xor."""
    crc = 0
    for b in bytearray(data):
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x8408 if crc & 1 else crc >> 1
    return crc


def velux_decode(bits):
    """velux/decoder.py:30-101 on a list of 0/1 symbols, with plain lists in
    place of bitstring (same slice-deletion semantics).  Returns None when no
    start of frame is found (the script exits), else a dict; `error` is set
    where bitstring's unpack would raise (too few bits)."""
    bits = [int(b) for b in bits]
    n, m = len(bits), len(VELUX_SOF)
    start = -1
    for i in range(0, n - m + 1):
        if bits[i:i + m] == VELUX_SOF:
            start = i
            break
    if start < 0:
        return None
    frame = bits[start:]
    del frame[0:len(frame):10]      # start bits (:43)
    del frame[8:len(frame):9]       # stop bits (:45)
    for i in range(0, len(frame) - 8, 8):   # LSB first -> reverse every byte but the last group (:48-49)
        frame[i:i + 8] = frame[i:i + 8][::-1]
    out = dict(frame_start=start, error=None)
    if len(frame) < 24:
        out["error"] = "short"
        return out
    control1 = int("".join(str(b) for b in frame[16:24]), 2)
    ml = control1 & 0x1F
    packet = frame[:(3 + ml + 2) * 8]
    pb = packet + [0] * (-len(packet) % 8)
    data = bytes(bytearray(int("".join(str(b) for b in pb[i:i + 8]), 2) for i in range(0, len(pb), 8)))
    out.update(packet=data, control1=control1 & ~0x1F & 0xFF, message_length=ml)
    if len(packet) < (3 + ml + 2) * 8:
        out["error"] = "short"
        return out
    crc16 = data[3 + ml] | (data[3 + ml + 1] << 8)
    crc_calc = crc16_kermit(data[2:3 + ml])
    out.update(crc16=crc16, crc_ok=crc_calc == crc16)
    msg = data[3:3 + ml]
    if ml < 16:
        out["error"] = "short message"   # unpack("... bytes:{ml-16} ...") raises
        return out
    out.update(control2=msg[0], destination=int.from_bytes(msg[1:4], "big"), source=int.from_bytes(msg[4:7], "big"),
               command=msg[7], payload=msg[8:ml - 8], counter=int.from_bytes(msg[ml - 8:ml - 6], "big"),
               mac=int.from_bytes(msg[ml - 6:ml], "big"))
    return out


def edges_to_events(bits, sample_rate, watchdog_us=None, start_level=0):
    """Turn a sliced 0/1 stream into the (level, tick) events a pigpio
    EITHER_EDGE callback with an optional watchdog would deliver: tick = sample
    index of the first sample at the new level, in integer microseconds
    (floor); one watchdog event (level 2) per run longer than ``watchdog_us``,
    at run start + watchdog_us.  This is the harness that feeds the *Sniffer
    rules from IQ-derived bits (the reference fed them from a GPIO pin)."""
    b = np.asarray(bits).astype(np.int8)
    ev = []
    if b.size == 0:
        return ev
    pos = np.flatnonzero(np.diff(b)) + 1
    level = int(start_level)
    prev_tick = 0
    starts = np.concatenate(([0], pos))
    levels = b[starts]
    if levels[0] != level:
        ev.append((int(levels[0]), 0))
    level = int(levels[0])
    for k in range(1, starts.size):
        tick = int(starts[k]) * 1000000 // int(sample_rate)
        if watchdog_us is not None and tick - prev_tick > watchdog_us:
            ev.append((2, prev_tick + watchdog_us))
        ev.append((int(levels[k]), tick))
        prev_tick = tick
    end_tick = int(b.size) * 1000000 // int(sample_rate)
    if watchdog_us is not None and end_tick - prev_tick > watchdog_us:
        ev.append((2, prev_tick + watchdog_us))
    return ev
