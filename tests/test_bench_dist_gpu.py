"""bench.py's multi-GPU plumbing driven on ONE GPU (TAN_FORCE_DIST=1: the NCCL process group exists and every collective of the step runs
at world size 1) -- so that the first 8-GPU driver run cannot die on plumbing: the JSON line carries `comm`, the default gradient
reduction is the north star's single all-reduce ('flat'), and `extra` holds the other mode, global negatives and both stage-2 batches."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_bench_line_under_forced_dist_has_comm_and_every_variant():
    env = dict(os.environ, TAN_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("TAN_DDP_MODE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--extra-steps", "2", "--settle-s", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    comm = d["comm"]
    assert comm["world_size_seen_by_backend"] == 1 and comm["backend"] == "nccl"
    assert comm["ddp_mode"] == "flat" and comm["collectives_per_step"] == 1           # ONE all-reduce of the flat gradient per step
    assert comm["gradient_bytes_per_step"] > 150e6
    names = [e["name"] for e in d["extra"]]
    assert any("'buckets'" in n for n in names) and any("global negatives" in n for n in names)
    assert any("B_local=128" in n for n in names) and any("B_local=16" in n for n in names)
    for e in d["extra"]:
        assert e["value"] > 0 and e["comm"]["world_size_seen_by_backend"] == 1
    bk = next(e for e in d["extra"] if "'buckets'" in e["name"])
    assert bk["comm"]["ddp_mode"] == "buckets" and bk["comm"]["collectives_per_step"] > 1
