# -*- coding: utf-8 -*-
"""Drop-in names of the reference's somfy pieces on the GPU path.

  blk     <- somfy/epy_block_1.py:14-67 (embedded in somfy/Telis1RTS.grc:892-945):
             the wake-up pulse finder, `blk(gr.sync_block)` with in_sig=[np.byte],
             out_sig=[np.byte] and the single parameter `bit_length`.
  decode  <- somfy/decoder.py:27-58 (sdr_amd.blocks.somfy_decode).

`blk.work(input_items, output_items)` runs the chunk through `k_preamble`
(libsdrhip `sdr_preamble_work`); the attributes `state`, `preamble`, `position`
and the printed lines are the reference's.  No CPU fallback."""
from __future__ import print_function

import ctypes as C

import numpy as np

from . import _lib
from .blocks import _SyncBase, somfy_decode as decode  # noqa: F401


class blk(_SyncBase):  # other base classes are basic_block, decim_block, interp_block
    """Embedded Python Block of somfy/Telis1RTS.grc on the GPU path."""

    def __init__(self, bit_length=1e-6):  # only default arguments here
        _SyncBase.__init__(self, name='Embedded Python Block', in_sig=[np.byte], out_sig=[np.byte])
        # attributes named like parameters become GRC callbacks
        self.bit_length = bit_length
        self.state = 0
        self.preamble = 0
        self.position = 0
        self._L = _lib.load()
        self._h = C.c_void_p()
        rc = self._L.sdr_preamble_create(C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            text = self._L.sdr_preamble_last_error(None)
            raise _lib.SdrError(rc, text.decode("utf-8", "replace") if text else "")
        print("length of one sample", bit_length)

    def close(self):
        if self._h:
            self._L.sdr_preamble_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def work(self, input_items, output_items):
        x = np.ascontiguousarray(input_items[0], np.int8)
        out = output_items[0]
        tmp = out if (out.dtype == np.int8 and out.flags.c_contiguous) else np.empty(x.size, np.int8)
        info = _lib.PreambleInfo()
        rc = self._L.sdr_preamble_work(self._h, C.c_void_p(x.ctypes.data), x.size, 0, C.c_void_p(tmp.ctypes.data),
                                       C.byref(info))
        if rc != 0:
            text = self._L.sdr_preamble_last_error(self._h)
            raise _lib.SdrError(rc, text.decode("utf-8", "replace") if text else "")
        self.state, self.preamble, self.position = info.state, info.preamble, info.position
        if info.events & 1:
            print("position of preamble", info.preamble_pos)
        if info.events & 2:
            print(x.size, info.sum8)
        if info.events & 4:
            print("length of preamble", info.preamble)
        if info.events & 8:
            raise ValueError("Shape of array too small to calculate a numerical gradient, "
                             "at least (edge_order + 1) elements are required.")
        if tmp is not out:
            out[:] = tmp
        return len(output_items[0])
