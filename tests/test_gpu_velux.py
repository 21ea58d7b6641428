"""velux io-homecontrol decoder on the GPU through the C ABI
(sdr_process_bits, SDR_DEC_VELUX) against the oracle restatement of
velux/decoder.py, pinned by the script's own captured symbol arrays."""
import numpy as np
import pytest

from oracle import decoders as odec
from sdr_amd import _lib, chain, velux

pytestmark = pytest.mark.gpu


def test_velux_reference_capture(kat):
    k = kat["velux_active"]
    d = velux.decode(k["bits"])
    o = odec.velux_decode(k["bits"])
    assert d["packet"].hex() == k["packet_hex"] and d["crc_ok"] and d["status"] == _lib.FRAME_OK
    for key in ("frame_start", "message_length", "control1", "crc16", "control2", "destination", "source", "command",
                "payload", "counter", "mac"):
        assert d[key] == o[key], key
    assert d["counter"] == k["counter"] and "%012x" % d["mac"] == k["mac_hex"]
    d2 = velux.decode(kat["velux_commented"]["bits"])
    assert d2["status"] == _lib.FRAME_CRC and d2["counter"] == kat["velux_commented"]["counter"]
    assert velux.decode([0, 1] * 200) is None


def test_velux_batch_rows_and_edge_cases(kat):
    """Several symbol streams as one batch (rows of sliced bytes), truncated and corrupted."""
    import torch
    rng = np.random.default_rng(3)
    base = kat["velux_active"]["bits"]
    rows = [list(base), list(base[:380]), list(base[:150]), list(base[50:]), [0] * 500]
    flip = list(base)
    flip[200] ^= 1
    rows.append(flip)
    n = 512
    mat = np.zeros((len(rows), n), np.int8)
    for i, r in enumerate(rows):
        mat[i, :len(r)] = r
    # the batch entry point takes IQ; the rows are fed through the slicer as a real signal: level 1 -> 1.0
    iq = (mat.astype(np.float32) + 0j).astype(np.complex64)
    with chain.Chain(1.0, decim=1, demod=chain.DEMOD_MAG2, thr=(0.3, 0.7), decoder=chain.DEC_VELUX, decoder_rate=1.0,
                     max_batch=len(rows), max_samples=n, max_frames=2) as c:
        c.batch_process(torch.from_numpy(iq.view(np.float32).reshape(len(rows), n, 2)).to("cuda:0"))
        bits = c.fetch_bits().reshape(len(rows), n)
        frames = c.fetch_frames()
    assert np.array_equal(bits, mat)
    got = {f.capture: f for f in frames}
    for i in range(len(rows)):
        o = odec.velux_decode(mat[i])
        if o is None:
            assert i not in got
            continue
        f = got[i]
        assert int(f.lim[0]) == o["frame_start"]
        if o["error"] == "short":
            assert f.status == _lib.FRAME_EXC_SHORT
        else:
            assert f.packet() == o["packet"] and (f.status == _lib.FRAME_OK) == bool(o["crc_ok"])
    assert got[0].status == _lib.FRAME_OK and got[5].status == _lib.FRAME_CRC and 4 not in got
    del rng
