"""velux io-homecontrol packet decoder (velux/decoder.py:30-101) on the GPU:
``decode(bits)`` takes the captured symbols (one 0/1 value per symbol, as the
reference's script holds them) and returns what the script prints, as a dict.
The search for the start of frame, UART de-framing, byte order and the
CRC-16/KERMIT check run in libsdrhip (SDR_DEC_VELUX); the field split below is
presentation of the packet bytes, as in the script's last lines (:89-101)."""
import numpy as np

from . import _lib, chain


def decode(bits):
    b = np.ascontiguousarray(bits, np.int8)
    with chain.Chain(1.0, decoder=chain.DEC_VELUX, decoder_rate=1.0, max_samples=max(64, b.size), max_frames=2) as c:
        c.process_bits(b)
        frames = c.fetch_frames()
    if not frames:
        return None   # "Received {} Bits but no Frame detected!"
    f = frames[0]
    data = f.packet()
    ml = int(f.lim[1])
    out = dict(frame_start=int(f.lim[0]), packet=data, status=f.status, message_length=ml,
               control1=f.pad_ & ~0x1F & 0xFF, crc_ok=f.status == _lib.FRAME_OK)
    if f.status in (_lib.FRAME_OK, _lib.FRAME_CRC) and ml >= 16:
        msg = data[3:3 + ml]
        out.update(crc16=data[3 + ml] | (data[3 + ml + 1] << 8), control2=msg[0],
                   destination=int.from_bytes(msg[1:4], "big"), source=int.from_bytes(msg[4:7], "big"),
                   command=msg[7], payload=msg[8:ml - 8], counter=int.from_bytes(msg[ml - 8:ml - 6], "big"),
                   mac=int.from_bytes(msg[ml - 6:ml], "big"))
    return out
