"""Drop-in names for TS33C/stream_decoder.py (StreamDecoder) and the epy block
of TS33C/live_decoder.grc (blk), on the GPU path."""
from .blocks import ERROR, INFO, TRACE, TS33CStreamDecoder as StreamDecoder, ts33c_blk as blk  # noqa: F401
