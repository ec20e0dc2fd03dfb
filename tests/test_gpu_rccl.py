"""RCCL really runs ("single-process world_size=1 fallback so the collective path is
exercised").  The test box has one GPU, so the process group has ONE rank — but its backend is "nccl" (RCCL on ROCm) and
every collective of the path goes through it on device tensors: the score all-gather of `get_ood_scores_clip` and of
`bench.py`'s timed region, the histogram all-reduce, the Mahalanobis broadcast.  N > 1 on real devices is the driver's
scaling run; N = 2 logic (two ranks sharing the one GPU, gloo) is tests/test_gpu_configs.py::test_cli_two_ranks_equal_one_rank
and the CPU suite's world_size-2 tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script_and_args, timeout=600):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_and_args,
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_every_collective_of_the_path_on_rccl_world_size_1():
    r = _torchrun([os.path.join(ROOT, "tests", "probes", "rccl_ws1.py")])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["backend"] == "nccl" and d["all_ok"], d


def test_bench_timed_region_with_the_rccl_all_gather():
    """bench.py's N-GPU logic at N = 1 under torchrun: barrier, all_gather_into_tensor of the score shards inside the
    timed region, MAX all-reduce of the time — on RCCL, device tensors, no host bounce."""
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "3", "--warmup", "1",
                   "--quick", "--ckpt", "ViT-B/32", "--batch", "128"])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["collective"].startswith("nccl (RCCL)"), d.get("collective")
    assert d["value"] > 0


def test_cli_under_torchrun_one_rank_nccl(tmp_path):
    """eval_ood_detection.py under torchrun --nproc-per-node 1: WORLD_SIZE = 1, so no process group is needed — and none
    is created; the CSV equals the plain run's."""
    import pandas as pd

    common = [os.path.join(ROOT, "eval_ood_detection.py"), "--in_dataset", "ImageNet10", "--CLIP_ckpt", "ViT-B/32", "-b", "64",
              "--synthetic", "--synthetic-n", "200"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    one = subprocess.run([sys.executable] + common + ["--name", "plain"], cwd=tmp_path, env=env, capture_output=True, text=True,
                         timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                          "127.0.0.1", "--master-port", str(port)] + common + ["--name", "tr1"], cwd=tmp_path, env=env,
                         capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    base = tmp_path / "results" / "ImageNet10" / "MCM"
    a = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_plain" / "plain.csv", index_col=0)
    b = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_tr1" / "tr1.csv", index_col=0)
    assert a.equals(b), (a, b)
