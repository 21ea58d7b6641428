#!/usr/bin/env python3
"""Writes tests/golden/somfy_fixtures.json by RUNNING the reference's own somfy/decoder.py (Python 3, numpy +
struct only, a script with no functions) in the build container with its `manchester` array literal
(somfy/decoder.py:27) substituted, and recording what it prints.  The reference source is read from
/root/reference at run time and executed in memory; nothing of it is stored here -- the fixture file holds
inputs (112 half-symbols) and the printed results only.  Never runs on the GPU box (no /root/reference there).

Cases: random valid frames (all five commands, random key / rolling code / address), frames with an
unknown control nibble (the script raises KeyError at its COMMANDS lookup, after the checksum line),
frames with one flipped on-air bit (checksum error when it sits in the last byte, undetected otherwise), and half-symbol arrays with invalid Manchester pairs
(0,0)/(1,1), which the script maps to bit 0.
"""
import contextlib
import io
import json
import os
import re
import sys

import numpy as np

REF = "/root/reference/somfy/decoder.py"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def run_reference(src, manchester):
    lit = "manchester = np.array([%s])" % ", ".join(str(int(v)) for v in manchester)
    lines = src.split("\n")
    idx = [i for i, l in enumerate(lines) if l.startswith("manchester = np.array([")]
    assert len(idx) == 1, "expected exactly one manchester array literal"
    lines[idx[0]] = lit
    out = io.StringIO()
    exc = None
    with contextlib.redirect_stdout(out):
        try:
            exec(compile("\n".join(lines), REF, "exec"), {"__name__": "__somfy_fixture__"})
        except Exception as e:   # the script's own latent exceptions are part of its behaviour
            exc = type(e).__name__
    return out.getvalue().split("\n"), exc


def parse(lines, exc):
    r = {"exception": exc}
    for l in lines:
        l = l.strip()
        m = re.match(r"Payload: (.*)", l)
        if m:
            r["payload_hex"] = "".join(t[2:] for t in m.group(1).split()).lower()
        m = re.match(r"Frame: (.*)", l)
        if m:
            r["frame_hex"] = "".join(t[2:] for t in m.group(1).split()).lower()
        m = re.match(r"Checksum: (\w+)", l)
        if m:
            r["checksum_ok"] = m.group(1) == "ok"
        m = re.match(r"Key: 0x([0-9A-F]+)", l)
        if m:
            r["key"] = int(m.group(1), 16)
        m = re.match(r"Control: (\w+) \(0x([0-9A-F]+)\)", l)
        if m:
            r["control_name"], r["control"] = m.group(1), int(m.group(2), 16)
        m = re.match(r"Rolling Code: (\d+) ", l)
        if m:
            r["rolling_code"] = int(m.group(1))
        m = re.match(r"Address: (\d+) ", l)
        if m:
            r["address"] = int(m.group(1))
    return r


def manchester_of(payload):
    out = []
    for b in payload:
        for k in range(7, -1, -1):
            out += [0, 1] if (b >> k) & 1 else [1, 0]      # rising edge = 1 (somfy/decoder.py:28)
    return out


def main():
    from sdr_amd import synth
    src = open(REF).read()
    rng = np.random.default_rng(20261111)
    cases = []

    def add(kind, m):
        lines, exc = run_reference(src, m)
        rec = parse(lines, exc)
        rec.update({"kind": kind, "manchester": [int(v) for v in m]})
        cases.append(rec)

    for i in range(24):
        ctrl = [0, 1, 2, 4, 8][i % 5]
        fr = synth.somfy_frame(0xA0 | int(rng.integers(0, 16)), ctrl, int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 24)))
        add("valid", manchester_of(fr))
    for i in range(8):
        ctrl = int(rng.choice([3, 5, 6, 7, 9, 10, 12, 15]))
        fr = synth.somfy_frame(0xA0 | int(rng.integers(0, 16)), ctrl, int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 24)))
        add("unknown_control", manchester_of(fr))
    for i in range(12):
        fr = bytearray(synth.somfy_frame(0xA7, [1, 2, 4, 8][i % 4], int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 24))))
        # one flipped on-air bit: in the last byte it changes frame[6] alone (checksum error); anywhere else it
        # flips the same bit of two de-obfuscated bytes and the XOR checksum does not see it
        bit = int(rng.integers(48, 56)) if i < 6 else int(rng.integers(16, 48))
        fr[bit // 8] ^= 0x80 >> (bit % 8)
        add("bit_flip", manchester_of(bytes(fr)))
    for i in range(12):
        m = manchester_of(synth.somfy_frame(0xA3, 2, int(rng.integers(0, 65536)), int(rng.integers(0, 1 << 24))))
        for p in rng.choice(np.arange(16, 56), size=int(rng.integers(1, 4)), replace=False):
            v = int(rng.integers(0, 2))
            m[2 * int(p)] = m[2 * int(p) + 1] = v   # (0,0) or (1,1): not a Manchester symbol
        add("invalid_pair", m)
    json.dump({"source": "printed by /root/reference/somfy/decoder.py with its manchester literal substituted "
                         "(tests/golden/make_somfy_fixtures.py)", "cases": cases},
              open(os.path.join(HERE, "somfy_fixtures.json"), "w"), indent=0)
    print("wrote %d cases (%s)" % (len(cases), ", ".join("%s: %d" % (k, sum(c["kind"] == k for c in cases))
                                                         for k in ("valid", "unknown_control", "bit_flip", "invalid_pair"))))


if __name__ == "__main__":
    main()
