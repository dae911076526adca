"""Two ranks of bench.py on ONE GPU: the multi-rank control flow of the measured step -- process-group set-up, gradient buckets leaving
from inside backward, the fused all-gather of the InfoNCE projections, global padded lengths, max-over-ranks timing, one JSON line from
rank 0 -- with the collectives through gloo on device tensors (RCCL refuses two ranks on one device; the RCCL calls themselves are
rehearsed by the 1-rank group of tests/test_trainer_gpu.py).  What the driver's 2 / 4 / 8-GPU runs execute, minus the interconnect."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_of_the_bench_step_on_one_gpu():
    env = dict(os.environ, MMDTI_DIST_BACKEND="gloo", MMDTI_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-rooflines", "--no-ragged-workload"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, p.stdout[-2000:]                         # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak" and r["config"]["parallelism"] == "dp2"
    assert r["value"] > 0 and abs(r["value"] - 2 * 256 / (r["ms_per_step"] * 1e-3)) < 1e-2 * r["value"]      # whole-job throughput
    done, total = (int(x) for x in r["config"]["grad_buckets_reduced_during_backward"].split("/"))
    assert total > 0 and done == total                               # every bucket of the gradient arena left from inside backward
