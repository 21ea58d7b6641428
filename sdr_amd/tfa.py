"""Drop-in names for TFA/stream_decoder.py (StreamDecoder) and the epy block of
TFA/analyzer2.grc (blk), on the GPU path."""
from .blocks import ERROR, INFO, TRACE, TFAStreamDecoder as StreamDecoder, tfa_blk as blk  # noqa: F401
