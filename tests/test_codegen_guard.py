"""Build-time guard for a code-generation hazard measured on the GPU (DESIGN.md section 4,
r01_i): if the compiler places `s_waitcnt vmcnt(0)` inside the staging loop, every 16-B
direct-to-LDS DMA is waited for before the next one is issued and the front end runs 2.7x
slower (0.46 -> 1.24 ms) with identical results -- no parity test can see it.  This test
compiles the device code to assembly (no GPU needed) and checks that the innermost loop
around every `global_load_lds_dwordx4` of the streaming kernels holds no vmcnt wait."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {
    "_Z10k_frontendILb1ELb0ELb0EEv11FrontParams10FrontKTaps": "k_frontend<true,false,false>",
    "_Z10k_frontendILb1ELb0ELb1EEv11FrontParams10FrontKTaps": "k_frontend<true,false,true>",
    "_Z15k_frontend_mfma11FrontParams": "k_frontend_mfma",
    "_Z15k_frontend_raw811FrontParams10FrontKTaps": "k_frontend_raw8",
    "_Z5k_pfbILi8EEv9PfbParams": "k_pfb<8>",
    "_Z14k_frontend_winILb0ELi0EEv11FrontParams": "k_frontend_win<false, 0>",
    "_Z14k_frontend_winILb1ELi0EEv11FrontParams": "k_frontend_win<true, 0>",
    "_Z14k_frontend_winILb0ELi1EEv11FrontParams": "k_frontend_win<false, MAG2>",
    "_Z14k_frontend_winILb1ELi1EEv11FrontParams": "k_frontend_win<true, MAG2>",
    "_Z14k_frontend_winILb0ELi18EEv11FrontParams": "k_frontend_win<false, QUAD | SLICE_ONLY>",
}


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "sdrhip.s"
    src = os.path.join(ROOT, "sdr_amd", "csrc", "sdrhip.hip")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                           "--cuda-device-only", "-o", str(out), src], cwd=os.path.dirname(src))
    return open(str(out)).read()


def _kernel_body(asm, sym):
    i = asm.index(sym + ":")
    return asm[i:asm.index(".end_amdhsa_kernel", i)].split("\n")


def _innermost_loop(lines, at):
    labels = {m.group(1): k for k, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    best = None
    for b in range(at + 1, len(lines)):
        m = re.match(r"^\s+s_(?:c?branch\w*)\s+(\.LBB\d+_\d+)", lines[b])
        if m and labels.get(m.group(1), 1 << 30) <= at:
            best = (labels[m.group(1)], b)
            break
    return best


@pytest.mark.parametrize("sym", sorted(KERNELS))
def test_no_vmcnt_wait_inside_the_dma_loop(device_asm, sym):
    lines = _kernel_body(device_asm, sym)
    dma = [k for k, l in enumerate(lines) if "global_load_lds_dwordx4" in l]
    assert dma, "%s no longer stages by LDS DMA" % KERNELS[sym]
    for at in dma:
        assert " nt" in lines[at], "staging DMA lost its non-temporal policy"
        loop = _innermost_loop(lines, at)
        assert loop is not None, "DMA is not inside a loop?"
        body = lines[loop[0]:loop[1] + 1]
        waits = [l.strip() for l in body if re.search(r"s_waitcnt.*vmcnt\(", l)]
        assert not waits, "%s: the staging loop waits on its own DMA: %s" % (KERNELS[sym], waits)
        others = [l.strip() for l in body if re.search(r"\b(global_load_dword|global_load_u|buffer_load|flat_load)", l)]
        assert not others, "%s: other loads share the DMA loop: %s" % (KERNELS[sym], others)


def test_hot_kernels_use_no_scratch_memory(device_asm):
    """Round 2: a dynamically indexed array inside the by-value BackParams argument sent the whole struct to
    private (scratch) memory -- every k_backend launch 2.4x slower, identical results.  The kernels on the
    measured paths must keep their state in registers: private_segment_fixed_size == 0."""
    hot = ["_Z9k_backend10BackParams", "_Z14k_frontend_winILb0ELi0EEv11FrontParams", "_Z14k_frontend_winILb1ELi0EEv11FrontParams",
           "_Z14k_frontend_winILb0ELi1EEv11FrontParams", "_Z14k_frontend_winILb1ELi1EEv11FrontParams",
           "_Z14k_frontend_winILb0ELi18EEv11FrontParams",
           "_Z15k_frontend_mfma11FrontParams", "_Z14k_fir_fff_mfmaPKfS0_ixxxS0_iiPf", "_Z16k_frontend_mfma211FrontParamsi",
           "_Z10k_frontendILb1ELb0ELb0EEv11FrontParams10FrontKTaps", "_Z10k_frontendILb1ELb0ELb1EEv11FrontParams10FrontKTaps", "_Z15k_frontend_raw811FrontParams10FrontKTaps",
           "_Z5k_pfbILi8EEv9PfbParams", "_Z10k_runs_seg10BackParams", "_Z15k_decode_pieces10BackParams"]
    # Round 3: the mag^2-only instantiations of k_frontend_win are held to 80 VGPRs (6 waves per SIMD instead of 5) at
    # the price of a few spilled dwords in cold paths -- measured 3-8 % FASTER; anything beyond that few
    # would be the round-2 kind of accident again
    # (and the slice-only discriminator instantiation: 92 B at 80 VGPRs, measured 5 % faster than the generic kernel's 95)
    spill_ok = {"_Z14k_frontend_winILb0ELi1EEv11FrontParams": 64, "_Z14k_frontend_winILb1ELi1EEv11FrontParams": 96,
                "_Z14k_frontend_winILb0ELi18EEv11FrontParams": 128}
    for sym in hot:
        i = device_asm.index(".amdhsa_kernel " + sym)
        blk = device_asm[i:device_asm.index(".end_amdhsa_kernel", i)]
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", blk)
        assert m and int(m.group(1)) <= spill_ok.get(sym, 0), "%s uses %s bytes of scratch per lane" % (sym, m.group(1) if m else "?")


def test_lds_only_barriers_do_not_wait_for_global_memory(device_asm):
    """k_frontend_mfma2's phase loop keeps the next tile's 16-B loads in flight across its barriers
    (lds_only_barrier: s_waitcnt lgkmcnt(0) + s_barrier between compiler-only memory clobbers).  A vmcnt wait in
    front of one of them would put an HBM round trip into every phase with identical results."""
    lines = _kernel_body(device_asm, "_Z16k_frontend_mfma211FrontParamsi")
    bars = [k for k, l in enumerate(lines) if re.search(r"\bs_barrier\b", l)]
    assert len(bars) >= 2
    for k in bars:
        near = [l.strip() for l in lines[max(0, k - 6):k] if "s_waitcnt" in l]
        assert near, "no s_waitcnt in front of an s_barrier: LDS traffic may still be in flight"
        assert not [l for l in near if "vmcnt" in l], "a barrier of the phase loop waits for global memory: %s" % near
