"""bench.py's contract on a small workload: ONE JSON line with the keys the driver reads, through the frame pipeline
(default), one frame at a time, and through the multi-GPU code path with one rank (RCCL reduce behind the C ABI)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--levels", "5", "--width", "320", "--spp", "8", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"]


def _bench(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + extra, capture_output=True, text=True,
                       timeout=240, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # the contract: one line on stdout, everything else on stderr
    return json.loads(lines[0])


def _check_contract(d, steps=4):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "Mrays/s" and d["steps"] == steps and d["n_gpus"] == 1 and d["value"] > 0 and d["ms_per_step"] > 0
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["kernel_ms"] > 0


def test_default_run_goes_through_the_frame_pipeline():
    d = _bench([])
    _check_contract(d)
    assert d["config"]["frames_in_flight"] == 4  # 8 spp per rank: four frames in flight
    assert d["config"]["ms_frame_alone"] > 0 and d["detail"]["separate_hw_queues"] >= 2
    assert d["roofline"]["traffic"] is None  # no counter profile of THIS workload is committed: the line must say so
    assert "NO counter profile" in d["roofline"]["note"]


def test_one_frame_at_a_time_and_trace_workload():
    d = _bench(["--frames-in-flight", "1"])
    _check_contract(d)
    assert "frames_in_flight" not in d["config"]
    t = _bench(["--workload", "trace"])
    _check_contract(t)
    assert t["scaling"] == "weak" and t["roofline"]["kernel"] == "k_trace_closest"


def test_multi_gpu_code_path_with_one_rank():
    env = {"RT_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577", "RANK": "0", "WORLD_SIZE": "1",
           "LOCAL_RANK": "0"}
    d = _bench([], env)
    _check_contract(d)
    assert d["detail"]["reduce"] == "rt_reduce_accum (RCCL, C ABI)" and d["config"]["frames_in_flight"] == 4
    d = _bench(["--frames-in-flight", "1"], env)
    assert d["detail"]["reduce"] == "rt_reduce_accum (RCCL, C ABI)"
    d = _bench(["--torch-reduce"], env)
    assert d["detail"]["reduce"] == "torch.distributed" and "frames_in_flight" not in d["config"]
