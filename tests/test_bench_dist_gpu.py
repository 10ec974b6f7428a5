"""bench.py's multi-GPU plumbing driven on ONE GPU (TAN_FORCE_DIST=1: the NCCL process group exists and every collective of the step runs
at world size 1) -- so that the first 8-GPU driver run cannot die on plumbing: the JSON line carries `comm`, the default gradient
reduction is the north star's one logical all-reduce of the whole gradient ('flat': in the two-chain step issued in contiguous pieces as
they become final), and `extra` holds the other modes ('buckets', 'single'), the bf16 wire dtype, global negatives, the stage-1 B_local = 16
point and both stage-2 batches."""
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
    # ONE logical all-reduce of the flat gradient per step: every element once, in <= 4 contiguous pieces (video stack, joint stack,
    # and what surrounds them in the flat buffer), on the two-chain step that the single-GPU line times
    assert comm["ddp_mode"] == "flat" and comm["two_chain_step"] and 1 <= comm["collectives_per_step"] <= 4
    assert comm["collectives_last_step"] == comm["collectives_per_step"]
    assert 150e6 < comm["gradient_bytes_per_step"] < 170e6 and comm["gradient_wire_dtype"] == "f32"
    names = [e["name"] for e in d["extra"]]
    assert any("'buckets'" in n for n in names) and any("'single'" in n for n in names) and any("global negatives" in n for n in names)
    assert any("as bf16" in n for n in names)
    assert any("B_local=128" in n for n in names) and sum("B_local=16" in n for n in names) == 2
    for e in d["extra"]:
        assert e["value"] > 0 and e["comm"]["world_size_seen_by_backend"] == 1
    bk = next(e for e in d["extra"] if "'buckets'" in e["name"])
    assert bk["comm"]["ddp_mode"] == "buckets" and bk["comm"]["collectives_per_step"] > 4 and bk["comm"]["two_chain_step"]
    sg = next(e for e in d["extra"] if "'single'" in e["name"])
    assert sg["comm"]["collectives_per_step"] == 1 and sg["comm"]["collectives_last_step"] == 1
    b16 = next(e for e in d["extra"] if "as bf16" in e["name"])
    assert b16["comm"]["gradient_wire_dtype"] == "bf16"


def test_bench_two_ranks_through_its_own_launcher():
    """`python bench.py --gpus 2` exactly as a user (or the driver's launch line) runs it -- `os.execvp` into `torch.distributed.run`,
    `dist.init_from_env`, per-rank seeds, the first step's parameter broadcast, barrier + max-over-ranks timing, rank 0's one JSON
    line -- at world size TWO before the first multi-GPU driver run (the reference idiom is
    end2end/main_nce.py:142-158).  This box has one GPU: both ranks share it (TAN_DIST_SHARE_GPU=1) and the collectives go over gloo
    (RCCL refuses two ranks on one device), so the numbers mean nothing; the path is what is tested."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", HSA_ENABLE_IPC_MODE_LEGACY="0",
               TAN_DIST_BACKEND="gloo", TAN_DIST_SHARE_GPU="1")
    for k in ("TAN_DDP_MODE", "TAN_FORCE_DIST", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extra",
                        "--settle-s", "0", "--batch", "32"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                         # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 64 and d["config"]["per_gpu_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    assert abs(d["value"] - 64 * 1e3 / d["ms_per_step"]) <= 0.01 * d["value"]          # whole-job rate = all ranks' videos / max-over-ranks time
    comm = d["comm"]
    assert comm["world_size_seen_by_backend"] == 2 and comm["backend"] == "gloo"
    assert comm["ddp_mode"] == "flat" and comm["two_chain_step"] and comm["collectives_last_step"] == comm["collectives_per_step"]
    assert "cpu_baseline" not in d                        # (rank 0 at N = 1 only)
