"""bench.py --gpus N as the driver starts it: a plain `python bench.py --gpus N`, no launcher around it.  The
script must start its own worker processes (one per GPU, torch.distributed underneath), relay rank 0's single
JSON line and the workers' exit status.  On a one-GPU box the ranks share the device and exchange their rows over
gloo (a rehearsal of the control flow, not a benchmark): the frame gathered from both ranks' row tiles must equal
a single-GPU render byte for byte."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_launches_its_own_ranks():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--spp", "8", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=ROOT)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["config"]["gathered_image_equals_single_gpu_render"] is True
    assert out["scaling"] == "strong" and out["value"] > 0 and out["config"]["spp"] == 8
    # 1200 x 800 x 8 paths per step, each at least one segment
    assert out["config"]["segments_per_step"] >= 1200 * 800 * 8


def test_bench_single_gpu_line_is_unchanged_by_the_launcher():
    """N = 1 runs in the calling process as before: one JSON line with the roofline block."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--spp", "8",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["roofline"]["kernel"] == "rtow_trace_kernel" and out["cpu_baseline"] is None
    assert 0.0 < out["roofline"]["frac"] < 2.0


def test_bench_collective_path_with_one_rank_over_rccl():
    """The code the multi-GPU benchmark runs per step - process group on the RCCL backend, the context's stream as a
    torch stream, all_gather_into_tensor of the padded rows enqueued behind the frame on that stream, rank 0's row
    scatter - with one rank on the box's one GPU (--rehearse-collective): three frames in flight must come back
    assembled into the image of a whole-frame render, and the process must exit cleanly."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-collective", "--steps", "5", "--warmup", "1",
                        "--spp", "8", "--pipeline", "3", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines   # nothing but the result on stdout: RCCL's version banner must not land there
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and "RCCL all_gather" in out["config"]["parallelism"] and "REHEARSAL" in out["config"]["parallelism"]
    assert out["config"]["gathered_image_equals_single_gpu_render"] is True
    assert out["config"]["segments_per_step"] >= 1200 * 800 * 8
