# -*- coding: utf-8 -*-
"""Deterministic synthetic captures for the receive chain.

The reference ships no IQ data (its .gitignore excludes every capture format),
so tests and bench.py transmit known payloads with the timings of the
reference's own transmitters / READMEs and check what comes back:

  TS33C  pulse/gap widths of TS33C/README.md:110-111 (x4 us), 90 bits/frame
  TFA    38400 Bd envelope implied by TFA/stream_decoder.py:130-190
  somfy  somfy/transmitter.py:100-173
  berner berner/transmitter.py:74-97
  mumbi  mumbi/transmitter.py:79-105, on-air widths mumbi/README.md:54-57

Everything is numpy on the host and seeded (numpy.random.default_rng); this
module is data generation, not part of the timed path.
"""
import numpy as np

# ------------------------------------------------------------------ TS33C


def ts33c_packet(channel=1, rolling_code=1, temp_tenths=216, humidity=39, battery_ok=True,
                 filler=(0x0e, 0x5e, 0xfb, 0x67, 0x11)):
    """10-byte packet laid out as TS33C/stream_decoder.py:180-190 reads it."""
    sign = 1 if temp_tenths >= 0 else 0
    t = abs(int(temp_tenths))
    p = [0] * 10
    p[0] = 0x9f
    p[1] = ((channel & 0x7) << 5) | (rolling_code & 0x0f)
    p[2], p[3] = filler[0], filler[1]
    p[4] = (((t // 10) % 10) << 4) | (t % 10)
    p[5] = (sign << 7) | ((1 if battery_ok else 0) << 6) | ((t // 100) & 0x0f)
    p[6] = (((humidity // 10) % 10) << 4) | (humidity % 10)
    p[7], p[8], p[9] = filler[2], filler[3], filler[4]
    return bytes(bytearray(p))


def ts33c_bits(packet):
    """On-air bit sequence: per byte 8 data bits (LSB first, inverted) + parity
    bit = 1 iff the byte has an even number of ones (inverse of
    TS33C/stream_decoder.py:156-178)."""
    bits = []
    for b in bytearray(packet):
        for j in range(8):
            bits.append(1 - ((b >> j) & 1))
        bits.append(1 if bin(b).count("1") % 2 == 0 else 0)
    return bits


def _ts33c_rule(sym):
    """The decode rule of TS33C/README.md:28-33 on ideal widths."""
    bits = []
    for lvl, dur in sym:
        if lvl:
            bits.append(1 if dur < 700 else 0)
        elif 700 < dur < 1550:
            bits.append(0)
    return bits


def _ts33c_runs(levels, widths, reset_gap_us):
    ps, pl, gs, gl = widths
    sym = []
    i = 0
    while i < len(levels):
        k = i
        while k + 1 < len(levels) and levels[k + 1] == levels[i]:
            k += 1
        run = k - i + 1
        if run > 2:
            return None
        sym.append((1, ps if run == 1 else pl) if levels[i] else (0, gs if run == 1 else gl))
        i = k + 1
    if sym[-1][0] == 0:
        sym[-1] = (0, reset_gap_us)
    else:
        sym.append((0, reset_gap_us))
    return sym


def ts33c_symbols(bits, rng=None, jitter=0.03, reset_gap_us=10004.0,
                  widths=(416.0, 912.0, 556.0, 1040.0)):
    """bits -> [(level, duration_us)].  The sensor sends differential
    Manchester (bit 0 = toggle the half-bit pair HL<->LH, bit 1 = repeat it),
    led by one high half-bit; on the README's 90 bits this reproduces the 67
    pulses / 67 gaps and their long/short pattern (TS33C/README.md:309-375).
    A run of one half-bit is "short", of two "long"; high and low half-bits
    have different lengths on air (416/912 us pulses, 556/1040 us gaps =
    README x4 us).  The frame tail is chosen so that the reference's rule
    yields exactly `bits` (a trailing short pulse reads as a final 1).  Ends
    with the reset gap."""
    bits = list(bits)
    levels = [1]
    prev = 'b'
    for b in bits:
        cur = prev if b == 1 else ('a' if prev == 'b' else 'b')
        levels += [1, 0] if cur == 'a' else [0, 1]
        prev = cur
    sym = None
    for cut in (2, 0, 1, 3):
        cand = _ts33c_runs(levels[:len(levels) - cut], widths, reset_gap_us)
        if cand is not None and _ts33c_rule(cand) == bits:
            sym = cand
            break
    if sym is None:
        raise ValueError("bit sequence not representable")
    if rng is not None:
        sym = [(l, d * (1.0 + jitter * (2.0 * rng.random() - 1.0))) for l, d in sym[:-1]] + [sym[-1]]
    return sym


# -------------------------------------------------------------------- TFA


def crc8_31(data):
    crc = 0
    for b in bytearray(data):
        crc ^= b
        for _ in range(8):
            crc = ((crc << 1) ^ 0x31) & 0xff if crc & 0x80 else (crc << 1) & 0xff
    return crc


def tfa_packet(ident=0x71c9, temp_c=21.4, humidity=54, battery_ok=True, b8=0x00, b9=0x56):
    """11 bytes `2d d4 id id T T H F .. .. crc` (TFA/README.md:31,
    TFA/stream_decoder.py:182-190)."""
    t = int(round((temp_c + 40.0) * 10))
    p = [0x2d, 0xd4, (ident >> 8) & 0xff, ident & 0xff,
         0x80 | ((t // 100) & 0x0f), (((t // 10) % 10) << 4) | (t % 10),
         humidity & 0x7f, 0x60 | (0 if battery_ok else 0x80), b8, b9]
    p.append(crc8_31(p[2:10]))
    return bytes(bytearray(p))


def tfa_bits(packet):
    """On-air order: each byte LSB first (TFA/stream_decoder.py:179)."""
    return [(b >> j) & 1 for b in bytearray(packet) for j in range(8)]


def tfa_symbols(bits, rng=None, jitter=0.02, t_bit=1e6 / 38400.0, delta=0.6, n_sync=90,
                reset_gap_us=700.0):
    """Envelope the TFA decoder expects after mag^2+LPF+threshold: six junk
    widths, n_sync zeros, data, reset gap.  A 0 bit is `delta*T` high then
    `(1-delta)*T` low; a 1 bit is high for T (TFA/docs/Demodulation_Description.png)."""

    def j(x):
        return x if rng is None else x * (1.0 + jitter * (2.0 * rng.random() - 1.0))

    sym = [(1, 40.0), (0, 9.0), (1, 31.0), (0, 12.0), (1, 22.0), (0, 14.0)]
    allbits = [0] * n_sync + list(bits) + [0]
    run = 0.0
    for b in allbits:
        if b:
            run += j(t_bit)
        else:
            d = j(delta * t_bit)
            sym.append((1, run + d))
            sym.append((0, j(t_bit) - d))
            run = 0.0
    sym[-1] = (0, reset_gap_us)
    return sym


# ------------------------------------------------------------------ somfy


def somfy_frame(key=0xA0, ctrl=0x01, rolling_code=5, address=0x365240):
    """7 obfuscated on-air bytes (somfy/transmitter.py:102-118)."""
    f = [(key | (rolling_code & 0x0f)) & 0xff, (ctrl << 4) & 0xff, (rolling_code >> 8) & 0xff,
         rolling_code & 0xff, address & 0xff, (address >> 8) & 0xff, (address >> 16) & 0xff]
    ck = 0
    for b in f:
        ck ^= b ^ (b >> 4)
    f[1] |= ck & 0xf
    for i in range(1, 7):
        f[i] ^= f[i - 1]
    return bytes(bytearray(f))


def somfy_symbols(frame, repeats=0, clock=640.0, wakeup=True, tail_gap_us=32000.0):
    """somfy/transmitter.py:125-173: wake-up, 2 hw-sync, sw-sync, 56 Manchester
    bits ("0" = high,low; "1" = low,high), eof, inter-frame gap; repeats carry
    7 hw-syncs."""
    bits = [(b >> (7 - j)) & 1 for b in bytearray(frame) for j in range(8)]
    sym = []
    if wakeup:
        sym += [(1, 10000.0), (0, 95000.0)]

    def one(nsync):
        s = [(1, 2500.0), (0, 2500.0)] * nsync + [(1, 4850.0), (0, clock)]
        for b in bits:
            s += [(0, clock), (1, clock)] if b else [(1, clock), (0, clock)]
        s += [(0, clock), (0, tail_gap_us)]
        return s

    sym += one(2)
    for _ in range(repeats):
        sym += one(7)
    return merge_symbols(sym)


# ----------------------------------------------------------- berner / mumbi


def berner_symbols(code32, repeats=5, tail_gap_us=12000.0):
    """berner/transmitter.py:82-97: preamble 5430/334, "0" 334/666, "1" 666/334."""
    bits = [(code32 >> (31 - j)) & 1 for j in range(32)]
    sym = []
    for _ in range(repeats):
        sym += [(1, 5430.0), (0, 334.0)]
        for b in bits:
            sym += [(1, 666.0), (0, 334.0)] if b else [(1, 334.0), (0, 666.0)]
    # trailing long pulse ends the last frame (berner/sniff.py:36-40 needs a
    # pulse outside 0..1000 us to print), then silence
    sym += [(1, 5430.0), (0, tail_gap_us)]
    return merge_symbols(sym)


def mumbi_symbols(code_bytes, repeats=8, on_air=True, pause_us=10000.0):
    """mumbi/transmitter.py:86-105; on-air widths 300/820 us (mumbi/README.md:54-57)."""
    bits = [(b >> (7 - j)) & 1 for b in bytearray(code_bytes) for j in range(8)][:34]
    short, long_ = (300.0, 820.0) if on_air else (333.0, 667.0)
    period = 1120.0 if on_air else 1000.0
    sym = []
    for _ in range(repeats):
        for b in bits:
            w = long_ if b else short
            sym += [(1, w), (0, period - w)]
        sym += [(0, pause_us)]
    return merge_symbols(sym)


MUMBI_CODES = {  # mumbi/transmitter.py:18-29 (34 bits = first 34 of 5 bytes)
    'a_on': (0x19, 0x10, 0x0F, 0x09, 0x00), 'a_off': (0x19, 0x10, 0x0E, 0x08, 0x00),
    'b_on': (0x19, 0x10, 0x0D, 0x0A, 0x00), 'b_off': (0x19, 0x10, 0x0C, 0x0B, 0x00),
    'c_on': (0x19, 0x10, 0x0B, 0x0E, 0x00), 'c_off': (0x19, 0x10, 0x0A, 0x0F, 0x00),
    'd_on': (0x19, 0x10, 0x07, 0x01, 0x00), 'd_off': (0x19, 0x10, 0x06, 0x00, 0x00),
    'all_on': (0x19, 0x10, 0x04, 0x03, 0x00), 'all_off': (0x19, 0x10, 0x08, 0x0D, 0x00),
}


# --------------------------------------------------------- kwmobile / baldr


def kwmobile_bits(sensor_id, channel, battery_ok, temp_tenths, humidity):
    """36 bits laid out as kwmobile/decoder.py:128-133 reads them (README frame
    `80 81 26 F5 E0`: the high nibble of byte 3 is constant 0xF)."""
    f = [sensor_id & 0xff,
         ((1 if battery_ok else 0) << 7) | ((channel & 3) << 4) | ((temp_tenths >> 8) & 0x0f),
         temp_tenths & 0xff,
         0xF0 | ((humidity >> 4) & 0x0f),
         (humidity & 0x0f) << 4]
    return [(b >> (7 - j)) & 1 for b in f for j in range(8)][:36]


def kwmobile_symbols(bits, repeats=12, tail_gap_us=6000.0):
    """kwmobile/README.md:32-44: 500 us pulse + 1000 us gap = 0, + 2000 us gap = 1,
    + 4000 us gap = end of frame; the frame is sent `repeats` times."""
    sym = []
    for _ in range(repeats):
        for b in bits:
            sym += [(1, 500.0), (0, 2000.0 if b else 1000.0)]
        sym += [(1, 500.0), (0, 4000.0)]
    sym += [(0, tail_gap_us)]
    return merge_symbols(sym)


def baldr_bits(sensor_id, channel, tx_mode, temp_tenths, humidity, lead_nibble=0x5, last_bit=0):
    """37 bits laid out as baldr/decoder.py:128-140 reads them (12-bit two's
    complement temperature)."""
    t = temp_tenths & 0xfff
    f = [((lead_nibble & 0xf) << 4) | ((sensor_id >> 4) & 0x0f),
         ((sensor_id & 0x0f) << 4) | (4 if tx_mode else 0) | (channel & 3),
         (t >> 4) & 0xff,
         ((t & 0x0f) << 4) | ((humidity >> 4) & 0x07),
         ((humidity & 0x0f) << 4) | ((last_bit & 1) << 3)]
    return [(b >> (7 - j)) & 1 for b in f for j in range(8)][:37]


def baldr_symbols(bits, repeats=7, tail_gap_us=25000.0):
    """baldr/README.md:27-46: 50 us start pulse + 9000 us gap, then per frame 585 us
    pulse + 2000 us gap = 0, + 4000 us gap = 1, + 9000 us gap between frames;
    a last 585 us pulse ends the transmission."""
    sym = [(1, 50.0), (0, 9000.0)]
    for _ in range(repeats):
        for b in bits:
            sym += [(1, 585.0), (0, 4000.0 if b else 2000.0)]
        sym += [(1, 585.0), (0, 9000.0)]
    sym += [(1, 585.0), (0, tail_gap_us)]
    return merge_symbols(sym)


# --------------------------------------------------------------- envelopes


def merge_symbols(sym):
    out = []
    for lvl, dur in sym:
        if out and out[-1][0] == lvl:
            out[-1] = (lvl, out[-1][1] + dur)
        else:
            out.append((lvl, dur))
    return out


def envelope(symbols, fs, lead_in_us=5000.0, total=None):
    """[(level, us)] -> uint8 0/1 samples at fs (edges at rounded sample times)."""
    sym = merge_symbols([(0, lead_in_us)] + list(symbols))
    t = np.cumsum([d for _, d in sym])
    edges = np.rint(t * fs / 1e6).astype(np.int64)
    n = int(edges[-1]) if total is None else int(total)
    env = np.zeros(n, np.uint8)
    start = 0
    for (lvl, _), e in zip(sym, edges):
        e = min(int(e), n)
        if lvl and e > start:
            env[start:e] = 1
        start = e
        if start >= n:
            break
    return env


def concat_envelopes(envs, total):
    out = np.zeros(total, np.uint8)
    pos = 0
    for e in envs:
        k = min(e.size, total - pos)
        if k <= 0:
            break
        out[pos:pos + k] = e[:k]
        pos += k
    return out


def ook_iq(env, fs, f_off=0.0, amp=0.7, sigma=0.02, rng=None, phase0=0.0):
    """On-off keyed carrier at f_off plus complex AWGN (sigma per component)."""
    n = env.size
    if f_off != 0.0:
        ph = 2.0 * np.pi * (f_off / fs) * np.arange(n, dtype=np.float64) + phase0
        c = np.exp(1j * ph)
    else:
        c = np.full(n, np.exp(1j * phase0))
    x = amp * env.astype(np.float64) * c
    if rng is not None and sigma > 0:
        x = x + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)


def wideband_capture(n, fs, emitters, seed, sigma=0.02, amp=0.5):
    """BASELINE.json configs[4] input: several OOK emitters at different carrier
    offsets in one wideband capture.  emitters: [(f_off_hz, symbols, lead_in_us)].
    Returns complex64[n]."""
    rng = np.random.default_rng(seed)
    x = sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    t = np.arange(n, dtype=np.float64)
    for f_off, sym, lead in emitters:
        env = envelope(sym, fs, lead_in_us=lead, total=n)
        ph = 2.0 * np.pi * (f_off / fs) * t + rng.random() * 2 * np.pi
        x += amp * env * np.exp(1j * ph)
    return x.astype(np.complex64)


def ts33c_capture(n, fs, seed, f_off=0.0, n_frames=3, **fields):
    """One TS33C capture: `n_frames` repeats of one packet from 5 ms in."""
    rng = np.random.default_rng(seed)
    pkt = ts33c_packet(**fields)
    sym = []
    for _ in range(n_frames):
        sym += ts33c_symbols(ts33c_bits(pkt), rng)
    env = envelope(sym, fs, total=n)
    return ook_iq(env, fs, f_off=f_off, rng=rng, phase0=rng.random() * 2 * np.pi), pkt, env


def random_ts33c_fields(rng):
    """Random sensor fields.  The last filler byte gets even bit-parity: its
    parity bit is then 1 and rides on the frame's final short pulse, the only
    tail `ts33c_symbols` can represent (README frame ends the same way)."""
    f = dict(_random_ts33c_fields(rng))
    fl = list(f["filler"])
    if bin(fl[4]).count("1") % 2:
        fl[4] ^= 1
    f["filler"] = tuple(fl)
    return f


def _random_ts33c_fields(rng):
    return dict(channel=int(rng.integers(1, 4)), rolling_code=int(rng.integers(0, 16)),
                temp_tenths=int(rng.integers(-399, 600)), humidity=int(rng.integers(0, 100)),
                battery_ok=bool(rng.integers(0, 2)),
                filler=tuple(int(v) for v in rng.integers(0, 256, 5)))


def fm_iq(n, fs, f_off=200e3, tones=((1e3, 0.5), (19e3, 0.1)), dev=75e3, sigma=0.01, seed=98):
    """WBFM test signal: sum of tones frequency-modulated +-dev at f_off.  FM-Radio.grc translates with
    center_freq = -shift (FM-Radio/FM-Radio.grc:621-666), i.e. the station sits at f_off = -shift of the tuner."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / fs
    m = np.zeros(n)
    for f, a in tones:
        m += a * np.sin(2 * np.pi * f * t)
    ph = 2 * np.pi * f_off * t + 2 * np.pi * dev * np.cumsum(m) / fs
    x = 0.7 * np.exp(1j * ph) + sigma * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    return x.astype(np.complex64)
