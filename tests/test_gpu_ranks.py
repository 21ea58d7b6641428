"""The N > 1 run of bench.py with real kernels on a one-GPU box: N ranks share GPU 0, the process group is gloo and
its collectives are staged through the host (SDR_BENCH_REHEARSAL=gloo-gpu).  Every rank runs what it runs on an
8-GPU node -- its shard of the headline batch with the frame gather on a side stream, its segment of the wideband
capture (channelizer rows gathered and dealt out), its segment of the long capture (frames-only exchange) -- and
bench.py's own checks compare them with single passes.  Uneven segment lengths at 3 ranks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


@pytest.mark.gpu
@pytest.mark.timeout(420)
@pytest.mark.parametrize("world", [2, 3])
def test_bench_ranks_share_one_gpu(world):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SDR_BENCH_REHEARSAL"] = "gloo-gpu"
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(world), "--steps", "3", "--warmup", "1", "--captures", "48",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and "rehearsal" in line
    assert line["frames_decoded"] == 2 * 48 * world
    assert len(line["roofline"]["kernel_ms_per_rank"]) == world
    ow = line["other_workloads"]
    seg, lc = ow["wideband_pfb64_segments"], ow["long_capture_frames_only"]
    assert seg["n_gpus"] == world and "equal ONE sdr_batch_process pass" in seg["verified"]
    assert lc["n_gpus"] == world and lc["frames"] >= 2 * 256 - 2
    assert lc["frames_decoded_locally"] + lc["frames_from_the_joined_table"] >= lc["frames"]
    assert lc["collective_bytes_per_rank_and_step"] < 64 * 1024 + 96 * lc["frames"]
