"""The C-ABI library loads and exports every symbol include/sdrhip.h declares
(no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

from sdr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sdrhip.h")).read()
    return sorted(set(re.findall(r"SDR_API\s+[\w\s\*]+?\b(sdr_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.EXPORTS) == names


def test_struct_mirrors_match_header_sizes():
    assert C.sizeof(_lib.Frame) == 96
    assert C.sizeof(_lib.ChainCfg) == 80
    assert C.sizeof(_lib.Out) == 32
    assert C.sizeof(_lib.PreambleInfo) == 32
    assert C.sizeof(_lib.SegmentPieces) == 56 and C.sizeof(_lib.SegmentInfo) == 40
    assert _lib.load().sdr_version() >= 100


def test_no_gpu_blocks_fail_loudly():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from sdr_amd import somfy
    with pytest.raises(_lib.SdrError):
        somfy.blk()


def test_firdes_is_host_only_and_matches_oracle():
    """Tap design runs on the host (no GPU needed) and equals the oracle's."""
    import numpy as np
    from oracle import dsp as odsp
    from sdr_amd import chain
    for args in [(1, 2e6, 20e3, 40e3, 0), (1, 2e6, 400e3, 200e3, 2), (1, 5e5, 23437.5, 1562.5, 0)]:
        assert np.array_equal(chain.low_pass(*args), odsp.low_pass(*args))
    assert chain.low_pass(1, 2e6, 20e3, 40e3).size == 121
    assert chain.low_pass(1, 5e5, 23437.5, 1562.5).size == 771
    assert np.array_equal(chain.resampler_taps(48, 50), odsp.resampler_taps(48, 50))


def test_no_gpu_fails_loudly():
    """Without a device sdr_create must fail (no CPU fallback)."""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from sdr_amd import chain
    with pytest.raises(_lib.SdrError) as e:
        chain.Chain(1e6)
    assert e.value.code in (_lib.SDR_E_NODEVICE, _lib.SDR_E_HIP)
