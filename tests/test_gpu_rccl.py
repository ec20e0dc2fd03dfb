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


def test_bench_gpus_8_logic_on_one_device():
    """The round-end 8-GPU run cannot be rehearsed on a 1-GPU box; its LOGIC can: `python bench.py --gpus 8`
    respawns itself as 8 ranks, which here share the one device (gloo, host bounce — the line says so).  Checked: the line is
    whole-job (8 x the per-rank images), every rank reports its own record — own images/s over its own steps, its all-gather time,
    its own clock / power sample — so that a sub-linear curve can be attributed to clocks, feed or the collective from the
    line alone."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--ckpt", "tiny", "--batch", "48", "--prompts", "20",
                        "--steps", "4", "--warmup", "1", "--sustain-seconds", "1.5", "--cpu-seconds", "0", "--ingest", "none",
                        "--no-arms", "--no-configs", "--no-refined", "--no-drift", "--no-live-traffic",
                        "--detail", "/tmp/bench_detail_ws8.json"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["collective"].startswith("gloo: 8 ranks share"), d
    assert d["config"]["batch_per_gpu"] == 48 and d["config"]["parallelism"] == "image-sharded x8"
    pr = d["per_rank"]
    assert len(pr) == 8 and all(r_["img_s"] > 0 and r_["ag_ms"] >= 0 for r_ in pr), pr
    assert d["value"] <= sum(r_["img_s"] for r_ in pr) * 1.001          # whole-job rate = images of all ranks / the slowest's time
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 8 * 48) < 1e-3 * 8 * 48
    assert len(d["allgather_ms"]) == 2 and d["allgather_ms"][0] <= d["allgather_ms"][1]
    full = json.load(open("/tmp/bench_detail_ws8.json"))
    assert [r_["rank"] for r_ in full["per_rank"]] == list(range(8))
    assert all(r_["kernel_faults"] == 0 and "smi" in r_ for r_ in full["per_rank"]), full["per_rank"]
    # eight ranks sampled the one device: whoever got busy samples agrees on the clock within the part's range
    clocks = [r_["sclk_mhz"] for r_ in full["per_rank"] if r_.get("sclk_mhz")]
    print("per-rank records:", json.dumps(full["per_rank"]))
    assert all(100 <= c <= 2600 for c in clocks), clocks


def test_cli_eight_ranks_ragged_shards_equal_one_rank(tmp_path):
    """eval_ood_detection.py as 8 ranks on the one device (gloo): 101 ID and 37 OOD images at batch 16 — shards of 13 / 5 images,
    the last ones short — and the refinement window falling into a few of the ranks only: CSV == the 1-rank CSV, re-score counts
    add up."""
    import pandas as pd

    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = [os.path.join(ROOT, "eval_ood_detection.py"), "--in_dataset", "ImageNet10", "--CLIP_ckpt", "ViT-B/32",
              "-b", "16", "--synthetic", "--synthetic-n", "101"]
    one = subprocess.run([sys.executable] + common + ["--name", "r1"], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    many = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                           "--master-port", str(port)] + common + ["--name", "r8"], cwd=tmp_path, env=env, capture_output=True, text=True,
                          timeout=1200)
    assert many.returncode == 0, many.stderr[-3000:]
    base = tmp_path / "results" / "ImageNet10" / "MCM"
    a = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_r1" / "r1.csv", index_col=0)
    b = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_r8" / "r8.csv", index_col=0)
    assert a.equals(b), (a, b)
    r1 = json.load(open(base / "CLIP_ViT-B/32_T_1_ID_r1" / "refine_rank0.json"))
    r8 = [json.load(open(base / "CLIP_ViT-B/32_T_1_ID_r8" / f"refine_rank{r}.json")) for r in range(8)]
    assert all(r["rescored"] == r1["rescored"] and r["threshold"] == r1["threshold"] for r in r8)
    assert sum(r["rescored_by_this_rank"] for r in r8) == r1["rescored_total"]
