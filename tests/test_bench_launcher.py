"""`python3 bench.py --gpus N` must start N ranks itself (the driver's command form) and fail loudly when
the machine has fewer GPUs.  The launcher is driven here with 2 gloo ranks on CPU in bench.py's rehearsal
mode (rendezvous, frame gather, barriers, max over ranks, rank 0's line -- no chain work, nothing measured)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launcher_starts_two_gloo_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(SDR_BENCH_REHEARSAL="gloo", SDR_BENCH_CPU_BUDGET="0.4"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["value"] is None and "rehearsal" in line   # nothing measured, and the line says so
    # the N > 1 line carries what the N = 1 line does: the CPU leg (rank 0, after the process group is gone, one
    # core and all cores) and every rank's kernel time
    cb = line["cpu_baseline"]
    assert cb is not None and cb["value"] > 0 and cb["cores"] == 1 and cb["all_cores"]["cores"] >= 1
    assert len(line["roofline"]["kernel_ms_per_rank"]) == 2


def test_cpu_workers_do_not_inherit_a_profiler(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    monkeypatch.setenv("LD_PRELOAD", "/x/libharness.so:/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("ROCPROF_COUNTER_COLLECTION", "1")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    env = bench.worker_env()
    assert env["LD_PRELOAD"] == "/x/libharness.so"
    assert not [k for k in env if k.startswith("ROCP")] and env["OMP_NUM_THREADS"] == "1"


def test_launcher_refuses_more_gpus_than_present():
    import torch
    have = torch.cuda.device_count()
    n = have + 1 if have >= 1 else 2
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs %d GPUs" % n in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_world_size_mismatch_is_an_error():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4"],
                       env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_a_failing_rank_fails_the_launcher():
    # rank 1 of a rehearsal dies at start-up (bad steps value only it sees): the launcher stops rank 0 and returns non-zero
    code = ("import os, sys, runpy\n"
            "if os.environ.get('RANK') == '1': sys.exit(7)\n"
            "sys.argv = [%r] + sys.argv[1:]\n"
            "runpy.run_path(%r, run_name='__main__')\n" % (BENCH, BENCH))
    # drive launch_ranks directly with a child that fails on rank 1
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    real = bench.__file__
    tmp = os.path.join(ROOT, "tests", "_bench_child_tmp.py")
    with open(tmp, "w") as f:
        f.write(code)
    try:
        bench.__file__ = tmp
        os.environ["SDR_BENCH_REHEARSAL"] = "gloo"
        rc = bench.launch_ranks(2, ["--gpus", "2", "--steps", "1", "--warmup", "0"], True)
    finally:
        bench.__file__ = real
        os.environ.pop("SDR_BENCH_REHEARSAL", None)
        os.remove(tmp)
    assert rc != 0
