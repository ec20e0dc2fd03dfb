"""BASELINE.json configs 1, 2 and 5 run through the command line on the HIP path (1 and 2 were never run):

  C1  `--in_dataset ImageNet10 --CLIP_ckpt ViT-B/16 -b 64`  (ImageNet-10 ID vs ImageNet-20 OOD, the full 500 + 1 000
      images, K = 10; reference eval_ood_detection.py:63-68, utils/common.py:36-60) in fp32 and in fp16: the first 16
      ID scores against the C oracle on the same pixels and prompts, device metrics == host (sklearn) metrics;
  C2  `--in_dataset ImageNet100 --dtype bf16`  (5 000 ID + iNaturalist / SUN / Places / Textures at 10 000 / 10 000 /
      10 000 / 5 640, K = 100, batch 512): the size-independent invariants of test_full_size_properties at K = 100, and
      bf16's (and fp16's) AUROC / FPR95 difference to the exact-fp32 arm per OOD set — bf16 is the dtype BASELINE names
      for this config and it does NOT meet 1e-4; the test records by how much and bounds it.
  C5  `--in_dataset {bird200,food101,pet37,car196} --templates FILE` (the fine-grained ID suites at their test-split sizes,
      80 templates x K prompts) against the four OOD sets;
  plus the same CLI under `torchrun --nproc-per-node 2` (two ranks sharing the one GPU, gloo): the CSV equals the 1-rank CSV.

No dataset exists offline: every set is the seeded synthetic set of the reference's size (mcm_amd.synth), generated in
HBM; weights are the seeded stand-ins."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def test_cli_full_round_batch_same_scores(tmp_path, monkeypatch):
    """`--full-round-batch` (512 -> 665 at ViT-B/16: every GEMM fills its tile rounds, EXPERIMENTS.md R5.11) changes the batch
    the images are scored in and nothing else: the same float32 scores, bit for bit."""
    import eval_ood_detection as cli

    monkeypatch.chdir(tmp_path)
    common = ["--in_dataset", "ImageNet10", "--CLIP_ckpt", "ViT-B/16", "--synthetic", "--synthetic-n", "700",
              "--refine-threshold", "off"]
    a = cli.main(common + ["-b", "512", "--name", "b512"])
    b = cli.main(common + ["-b", "512", "--full-round-batch", "--name", "b665"])
    assert (a["batch_size"], b["batch_size"]) == (512, 665)
    np.testing.assert_array_equal(_np(a["in_score"]), _np(b["in_score"]))
    np.testing.assert_array_equal(_np(a["out_scores"]["ImageNet20"]), _np(b["out_scores"]["ImageNet20"]))


@pytest.mark.parametrize("dtype", ["fp32", "fp16", "fp16x2"])
def test_config1_imagenet10_vs_imagenet20_b16_batch64(tmp_path, monkeypatch, dtype):
    import pandas as pd

    import eval_ood_detection as cli
    from mcm_amd import detection
    from mcm_amd.config import geometry
    from mcm_amd.metrics import get_measures
    from mcm_amd.synth import DevicePatternLoader
    from mcm_amd.weights import synth_state_dict
    from oracle import oracle as orc
    from utils.common import get_test_labels

    monkeypatch.chdir(tmp_path)
    r = cli.main(["--in_dataset", "ImageNet10", "--CLIP_ckpt", "ViT-B/16", "-b", "64", "--dtype", dtype,
                  "--name", "c1", "--synthetic"])
    s_in, s_out = _np(r["in_score"]), _np(r["out_scores"]["ImageNet20"])
    assert s_in.shape == (500,) and s_out.shape == (1000,)          # the full sets of config 1
    assert s_in.dtype == np.float32 and np.isfinite(s_in).all() and np.isfinite(s_out).all()
    assert (s_in <= -1.0 / 10 + 1e-6).all() and (s_in >= -1.0).all()  # MCM = -max softmax over K = 10
    # first 16 ID images against the C oracle: same pixels (regenerated: image i = f(seed, i)), same prompts
    geo = geometry("ViT-B/16")
    args = types.SimpleNamespace(in_dataset="ImageNet10", ckpt="openai/clip-vit-base-patch16", weights=None)
    labels = get_test_labels(args)
    tok = detection._tokenizer(args, None)
    ids = tok([detection.PROMPT.format(c=c) for c in labels], padding=True, return_tensors="pt")["input_ids"].numpy()
    px = next(iter(DevicePatternLoader(500, 224, 10, 64, torch.device("cuda", 0), ood=False, seed=cli.SEEDS["id"])))[0]
    o = orc.OracleCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"))  # the CLI's default --synthetic-weights
    want = orc.score_features(o.encode_image(px[:16].cpu().numpy()), o.encode_text(ids), 1.0, 0)
    # (fp16x2: the whole run through the split-activation arm — the fp32 arm's tolerance, from fp16 MFMAs)
    tol = dict(rtol=0, atol=2e-7) if dtype in ("fp32", "fp16x2") else dict(rtol=0, atol=2e-5)
    assert ("refine" in r) == (dtype == "fp16")   # only a raw 16-bit run has a threshold neighbourhood to re-score
    np.testing.assert_allclose(s_in[:16], want, **tol)
    # device metrics (the CLI's default route) == host metrics (sklearn, the reference's route) on the same scores
    a, p, f = r["measures"]["ImageNet20"]
    ha, hp, hf = get_measures(-s_in, -s_out)
    assert abs(a - ha) <= 1e-12 and abs(p - hp) <= 1e-12 and f == hf
    df = pd.read_csv(tmp_path / "results/ImageNet10/MCM/CLIP_ViT-B/16_T_1_ID_c1/c1.csv", index_col=0)
    assert list(df.index) == ["ImageNet20", "AVG"]
    np.testing.assert_allclose(df.loc["ImageNet20"].values, np.round([100 * f, 100 * a, 100 * p], 2), atol=1e-9)


def test_config2_imagenet100_bf16_full_size(tmp_path, monkeypatch):
    import eval_ood_detection as cli
    from mcm_amd.metrics import get_measures

    monkeypatch.chdir(tmp_path)
    common = ["--in_dataset", "ImageNet100", "--CLIP_ckpt", "ViT-B/16", "--synthetic"]
    sizes = {"iNaturalist": 10000, "SUN": 10000, "places365": 10000, "dtd": 5640}
    runs = {}
    with pytest.warns(RuntimeWarning):  # ImageNet-100's class-name files are not on this box: placeholder names
        for dt in ("bf16", "fp32", "fp16"):   # (fp16: --refine-threshold exact, the form that promises the fp32 arm's count)
            runs[dt] = cli.main(common + ["--dtype", dt, "--name", f"c2_{dt}"] + (["--refine-threshold", "exact"] if dt == "fp16" else []))
        runs["fp16_default"] = cli.main(common + ["--dtype", "fp16", "--name", "c2_fp16_default"])
    r = runs["bf16"]
    s_in = _np(r["in_score"])
    assert s_in.shape == (5000,)
    K = 100
    assert (s_in <= -1.0 / K + 1e-7).all() and (s_in >= -1.0).all() and np.isfinite(s_in).all()
    for name, n in sizes.items():
        s = _np(r["out_scores"][name])
        assert s.shape == (n,) and np.isfinite(s).all() and (s <= -1.0 / K + 1e-7).all()
        a, p, f = r["measures"][name]
        ha, hp, hf = get_measures(-s_in, -s)                       # device metrics == host metrics
        assert abs(a - ha) <= 1e-12 and abs(p - hp) <= 1e-12 and f == hf
        assert 0.0 < a < 1.0
    # determinism at size: a second bf16 run writes bit-identical scores
    again = cli.main(common + ["--dtype", "bf16", "--name", "c2_bf16_again"])
    assert torch.equal(again["in_score"], r["in_score"])
    assert all(torch.equal(again["out_scores"][k], r["out_scores"][k]) for k in sizes)
    # drift of the 16-bit modes against the exact-fp32 arm, per OOD set (quantum of FPR95: 1e-4, dtd 1.8e-4)
    report = {}
    for dt in ("bf16", "fp16"):
        report[dt] = {k: tuple(abs(x - y) for x, y in zip(runs[dt]["measures"][k], runs["fp32"]["measures"][k]))
                      for k in sizes}
    print("config 2 drift vs the fp32 arm (dAUROC, dAUPR, dFPR95):", report)
    # The CLI's seeded stand-in weights are fp16-exact by default (--synthetic-weights, like the reference's checkpoints).
    # Round 3 ran this config on fp32-VALUED weights rounded to one operand and recorded fp16 dAUROC 1.9 - 2.0e-4 — the
    # one measured miss of the bar; `--synthetic-weights fp32` now runs the split-weight GEMMs instead and is held to the
    # same bar below.  bf16: 8 significand bits in every activation, the documented coarser arm.
    runs["fp16_fp32w"] = cli.main(common + ["--dtype", "fp16", "--synthetic-weights", "fp32", "--refine-threshold", "exact",
                                            "--name", "c2_fp16_fp32w"])
    runs["fp32_fp32w"] = cli.main(common + ["--dtype", "fp32", "--synthetic-weights", "fp32", "--name", "c2_fp32_fp32w"])
    report["fp16_fp32w"] = {k: tuple(abs(x - y) for x, y in zip(runs["fp16_fp32w"]["measures"][k],
                                                                  runs["fp32_fp32w"]["measures"][k])) for k in sizes}
    print("config 2, fp32-valued weights, split-weight fp16 arm vs the fp32 arm:", report["fp16_fp32w"])
    # FPR95: the CLI's threshold refinement (mcm_amd/refine.py) re-scores the images within a few noise widths of the
    # threshold — by default with the split-activation arm of the same handle (an exact-grade arm: within the quantum of the
    # fp32 run, like HF itself), with `--refine-threshold exact` the inner window also with the exact-fp32 arm, so that the
    # 16-bit run reports the fp32 run's FPR95 — equal, not close
    for k, n in sizes.items():
        dflt = abs(runs["fp16_default"]["measures"][k][2] - runs["fp32"]["measures"][k][2])
        assert dflt * n <= 1.5, (k, dflt)
    assert runs["fp16_default"]["refine"]["rescorer"] == "x2" and "rescored_exact" not in runs["fp16_default"]["refine"]
    assert runs["fp16"]["refine"]["rescorer"] == "x2" and runs["fp16"]["refine"]["rescored_exact_total"] <= 64 + 0.1 * runs["fp16"]["refine"]["rescored_total"]
    assert runs["bf16"]["refine"]["rescorer"] == "fp32"
    for k, n in sizes.items():
        for arm in ("fp16", "fp16_fp32w"):
            da, dp, df = report[arm][k]
            assert da <= 1e-4 and dp <= 1e-4 and df == 0.0, (arm, k, report[arm][k])
        da, dp, df = report["bf16"][k]
        assert da <= 5e-3 and df * n <= 1.5, (k, report["bf16"][k])
    for arm in ("fp16", "fp16_fp32w", "bf16"):
        st = runs[arm]["refine"]
        print(f"config 2 threshold refinement [{arm}]:", st)
        # measured: fp16 744 of 40 640 images (1.8 %), bf16 — 11 x the score noise — 3 182 (7.8 %)
        assert 0 < st["rescored_total"] <= (0.05 if arm != "bf16" else 0.15) * (5000 + sum(sizes.values())), st
    # without it the fp16 arm is within a couple of images, not equal (recorded, bounded)
    raw = cli.main(common + ["--dtype", "fp16", "--refine-threshold", "off", "--name", "c2_fp16_raw"])
    assert "refine" not in raw
    for k, n in sizes.items():
        assert abs(raw["measures"][k][2] - runs["fp32"]["measures"][k][2]) * n <= 2.5, k


def test_cli_two_ranks_equal_one_rank(tmp_path):
    """eval_ood_detection.py under torchrun, world size 2 (both ranks on the one GPU of the test box, gloo): rank-0
    reporting, per-rank shards of the device-generated sets, device metrics after the gather — CSV == the 1-rank CSV."""
    import pandas as pd

    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = [os.path.join(ROOT, "eval_ood_detection.py"), "--in_dataset", "ImageNet10", "--CLIP_ckpt", "ViT-B/32",
              "-b", "64", "--synthetic", "--synthetic-n", "333"]
    one = subprocess.run([sys.executable] + common + ["--name", "ws1"], cwd=tmp_path, env=env, capture_output=True,
                         text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533"] + common + ["--name", "ws2"],
                         cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-3000:]
    base = tmp_path / "results" / "ImageNet10" / "MCM"
    a = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_ws1" / "ws1.csv", index_col=0)
    b = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_ws2" / "ws2.csv", index_col=0)
    assert list(a.index) == ["ImageNet20", "AVG"] and a.equals(b), (a, b)
    # threshold refinement was on in both runs (the default) and is SHARDED under torchrun: each rank re-scored only the window
    # images of its own shard — the per-rank counts add up to the 1-rank run's, nobody did all of it
    import json

    r1 = json.load(open(base / "CLIP_ViT-B/32_T_1_ID_ws1" / "refine_rank0.json"))
    r2 = [json.load(open(base / "CLIP_ViT-B/32_T_1_ID_ws2" / f"refine_rank{r}.json")) for r in (0, 1)]
    assert r1["rescorer"] == "x2" and r1["rescored_total"] > 0 and r1["rescored_by_this_rank"] == r1["rescored_total"]
    assert all(r["rescored"] == r1["rescored"] and r["threshold"] == r1["threshold"] for r in r2)
    assert sum(r["rescored_by_this_rank"] for r in r2) == r1["rescored_total"]
    assert max(r["rescored_by_this_rank"] for r in r2) < r1["rescored_total"]


@pytest.mark.parametrize("suite,K,n_id", [("bird200", 200, 5794), ("food101", 101, 25250), ("pet37", 37, 3669),
                                          ("car196", 196, 8041)])
def test_config5_finegrained_suite_with_80_template_bank(tmp_path, monkeypatch, suite, K, n_id):
    """BASELINE config 5 through the command line at the suites' own sizes: CUB-200 / Food-101 / Pets / Cars as the ID
    set (K = 200 / 101 / 37 / 196 concepts; test-split sizes of SURVEY.md §8d), the bank built from 80 templates x K
    prompts (`--templates FILE`), against the four OOD sets.  Size-independent checks: shapes, range of -max softmax
    over K, device metrics == host metrics per set, and that the 80-template bank differs from the single-prompt one."""
    import pandas as pd

    import eval_ood_detection as cli
    from mcm_amd.metrics import get_measures

    monkeypatch.chdir(tmp_path)
    t = tmp_path / "templates80.txt"
    t.write_text("\n".join(f"a photo of a {{c}}, rendition {i}." if i % 2 else f"style {i}: the {{c}}" for i in range(80)) + "\n")
    common = ["--in_dataset", suite, "--CLIP_ckpt", "ViT-B/16", "--synthetic"]
    with pytest.warns(RuntimeWarning):   # the suites' class names come from their archives, which are not on this box
        r = cli.main(common + ["--name", "c5", "--templates", str(t)])
        single = cli.main(common + ["--name", "c5_single", "--synthetic-n", "512"])
    s_in = _np(r["in_score"])
    assert s_in.shape == (n_id,) and np.isfinite(s_in).all()
    assert (s_in <= -1.0 / K + 1e-7).all() and (s_in >= -1.0).all()
    for name, n in {"iNaturalist": 10000, "SUN": 10000, "places365": 10000, "dtd": 5640}.items():
        s = _np(r["out_scores"][name])
        assert s.shape == (n,) and np.isfinite(s).all()
        a, p, f = r["measures"][name]
        ha, hp, hf = get_measures(-s_in, -s)
        assert abs(a - ha) <= 1e-12 and abs(p - hp) <= 1e-12 and f == hf
    assert not np.array_equal(s_in[:512], _np(single["in_score"])[:512])   # the ensemble bank is another bank
    df = pd.read_csv(tmp_path / f"results/{suite}/MCM/CLIP_ViT-B/16_T_1_ID_c5/c5.csv", index_col=0)
    assert list(df.index) == ["iNaturalist", "SUN", "places365", "dtd", "AVG"] and np.isfinite(df.values).all()
