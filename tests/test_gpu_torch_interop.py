"""libsdrhip and PyTorch-ROCm in one process, library used first: both must end up
on one HIP runtime (sdr_amd._lib._share_torch_hip_runtime), device tensors go in
by pointer and the chain runs on torch's stream."""
import numpy as np
import pytest

from oracle import dsp as odsp

pytestmark = pytest.mark.gpu


def test_library_first_then_torch_device_tensor():
    from sdr_amd import chain, synth
    n = 1 << 16
    iq, _, _ = synth.ts33c_capture(n, 1e6, seed=5, f_off=100e3)
    taps = chain.low_pass(1.0, 1e6, 20e3, 40e3)
    want = odsp.ook_chain(iq, taps, 50, 1.0, 0.05, 0.15, 1, fc=100e3, fs=1e6)
    with chain.Chain(1e6, decim=50, center_freq=100e3, taps=taps, max_samples=n) as c:
        c.batch_process(iq[None, :])          # library touches the GPU before torch does
        np.testing.assert_array_equal(c.fetch_bits(), want)
        import torch
        x = torch.from_numpy(iq.view(np.float32).reshape(1, n, 2).copy()).cuda()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            x2 = x * 1.0                       # produced on a side stream
            c.set_stream(s.cuda_stream)
            c.batch_process(x2)
        np.testing.assert_array_equal(c.fetch_bits(), want)
        c.set_stream(0)
