"""--score maha on the device (mcm_maha_*, mcm_encode_image_raw) vs the reference's own outputs
(tests/golden/maha_tiny.npz) and the C oracle."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net():
    from mcm_amd.config import TEST_GEOMETRIES
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = TEST_GEOMETRIES["tiny"]
    n = NativeCLIP(geo, synth_state_dict(geo, seed=0), precision="fp32", max_batch=64, max_prompt_tokens=64 * 16)
    yield n
    n.close()


def _t(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag", ["raw", "norm"])
def test_scoring_kernel_vs_reference_outputs(net, golden_dir, tag):
    from oracle import oracle as orc

    g = np.load(os.path.join(golden_dir, "maha_tiny.npz"))
    state = net.maha_prepare(_t(g[f"mean_{tag}"]), _t(g[f"prec_{tag}"]))
    got = net.maha_scores(_t(g[f"feat_in_{tag}"]), state).cpu().numpy()
    np.testing.assert_allclose(got, g[f"in_{tag}"], rtol=2e-5, atol=1e-5)       # the reference itself
    np.testing.assert_allclose(got, orc.maha_scores(g[f"feat_in_{tag}"], g[f"mean_{tag}"], g[f"prec_{tag}"]),
                               rtol=2e-5, atol=1e-5)


def test_scoring_kernel_many_classes_near_the_means(net):
    """C = 1000 classes, features close to a class mean: d = q - W.f + k cancels to ~1e-4 of its terms;
    the fp64 accumulation must still agree with the direct per-class form."""
    from oracle import oracle as orc

    P = net.geo.proj_dim
    rng = np.random.default_rng(0)
    means = rng.standard_normal((1000, P)).astype(np.float32) * 3
    a = rng.standard_normal((P, P)).astype(np.float64)
    prec = (a @ a.T / P + np.eye(P)).astype(np.float32)
    prec[0, 1] += 0.01                                       # the reference does not symmetrise P
    feats = (means[rng.integers(0, 1000, 300)] + 0.01 * rng.standard_normal((300, P))).astype(np.float32)
    state = net.maha_prepare(_t(means), _t(prec))
    got = net.maha_scores(_t(feats), state).cpu().numpy()
    want = orc.maha_scores(feats, means, prec)
    p64 = prec.astype(np.float64)
    exact = np.array([min(0.5 * (f - m).astype(np.float64) @ p64 @ (f - m).astype(np.float64)
                          for m in means[np.argsort(((means - f) ** 2).sum(1))[:3]]) for f in feats])
    np.testing.assert_allclose(got, exact, rtol=1e-5, atol=1e-9)          # fp64 ground truth
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=1e-7)           # the fp32 restatement is the noisier side


def test_end_to_end_vs_reference(net, golden_dir, tmp_path):
    """get_mean_prec + get_Mahalanobis_score through the native tower (fp32 mode) vs the reference's run
    on HF: same loaders, same quirks (batch-index class sets, dropped OOD tail)."""
    import torch

    from mcm_amd.detection import get_Mahalanobis_score, get_mean_prec
    from mcm_amd.synth import make_pixels

    g = np.load(os.path.join(golden_dir, "maha_tiny.npz"))
    n_cls, bs, geo = int(g["n_cls"]), int(g["batch"]), net.geo

    class DS:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    class Loader:
        def __init__(self, n, ood, seed):
            self.dataset, self.ood, self.seed = DS(n), ood, seed

        def __len__(self):
            return -(-self.dataset.n // bs)

        def __iter__(self):
            for s in range(0, self.dataset.n, bs):
                n = min(bs, self.dataset.n - s)
                px, lab = make_pixels(n, geo.image_size, n_cls, ood=self.ood, seed=self.seed, start=s)
                yield torch.from_numpy(px), torch.from_numpy(lab)

    for normalize, tag in ((False, "raw"), (True, "norm")):
        args = types.SimpleNamespace(n_cls=n_cls, feat_dim=geo.proj_dim, model="CLIP", normalize=normalize,
                                     template_dir=str(tmp_path), in_dataset="ImageNet10", max_count=250,
                                     batch_size=bs)
        mean, prec = get_mean_prec(args, net, Loader(int(g["n_train"]), False, 7))
        np.testing.assert_allclose(mean.numpy(), g[f"mean_{tag}"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(prec.numpy(), g[f"prec_{tag}"], rtol=2e-3, atol=2e-3 * np.abs(g[f"prec_{tag}"]).max())
        assert os.path.exists(tmp_path / f"CLIP_precision_ImageNet10_250_{normalize}.pt")
        s_in = get_Mahalanobis_score(args, net, Loader(int(g["n_id"]), False, 1), mean, prec, in_dist=True)
        s_out = get_Mahalanobis_score(args, net, Loader(int(g["n_ood"]), True, 2), mean, prec, in_dist=False)
        assert s_in.dtype == np.float32 and s_in.shape == g[f"in_{tag}"].shape
        assert s_out.shape == g[f"out_{tag}"].shape          # trailing partial OOD batch dropped
        np.testing.assert_allclose(s_in, g[f"in_{tag}"], rtol=2e-3, atol=1e-4)
        np.testing.assert_allclose(s_out, g[f"out_{tag}"], rtol=2e-3, atol=1e-4)


def test_cli_maha(tmp_path, monkeypatch):
    import pandas as pd

    import eval_ood_detection as cli

    monkeypatch.chdir(tmp_path)
    cli.main(["--in_dataset", "ImageNet10", "--CLIP_ckpt", "ViT-B/32", "-b", "64", "--synthetic-n", "640",
              "--score", "maha", "--name", "m", "--dtype", "fp16"])
    df = pd.read_csv(tmp_path / "results" / "ImageNet10" / "maha" / "CLIP_ViT-B/32_T_1_ID_m" / "m.csv", index_col=0)
    assert list(df.index) == ["ImageNet20", "AVG"] and np.isfinite(df.values).all()
