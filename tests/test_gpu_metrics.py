"""Device metric kernels (mcm_measures) vs the reference's get_measures
outputs (tests/golden/measures.npz) and vs the host restatement pinned to them."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from mcm_amd.metrics import get_measures

pytestmark = pytest.mark.gpu

TOL = 1e-12  # fp64 sums in a different order than sklearn's; counts themselves are exact


@pytest.fixture(scope="module")
def net():
    from mcm_amd.config import TEST_GEOMETRIES
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = TEST_GEOMETRIES["tiny"]
    n = NativeCLIP(geo, synth_state_dict(geo, seed=0), precision="bf16", max_batch=64,
                   max_prompt_tokens=64 * 16)
    yield n
    n.close()


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.mark.parametrize("case", ["kat", "gauss", "ties", "narrow", "equal", "sep"])
def test_measures_match_reference_golden(net, golden_dir, case):
    g = np.load(os.path.join(golden_dir, "measures.npz"))
    pos, neg = g[f"{case}_pos"], g[f"{case}_neg"]
    got = np.array(net.measures(_dev(pos), _dev(neg)))
    p32, n32 = pos.astype(np.float32), neg.astype(np.float32)
    np.testing.assert_allclose(got, np.array(get_measures(p32, n32)), rtol=0, atol=TOL)
    both, both32 = np.concatenate([pos, neg]), np.concatenate([p32, n32])
    if np.unique(both).size == np.unique(both32).size:  # the fp32 cast kept every distinct value
        np.testing.assert_allclose(got, g[f"{case}_measures"], rtol=0, atol=TOL)


@pytest.mark.parametrize("n_pos,n_neg,quant,level", [
    (50000, 10000, 0, 0.95),     # ImageNet-1k vs one OOD set, the size the reference evaluates
    (50000, 10000, 4096, 0.95),  # heavy ties: scores quantised to 4096 levels
    (5000, 5640, 64, 0.95),      # Textures-sized, very heavy ties
    (1237, 4099, 0, 0.90),       # ragged sizes (tile tails), another recall level
    (1, 1, 0, 0.95), (3, 1, 0, 0.95), (1, 7, 2, 0.95),
])
def test_measures_match_host(net, n_pos, n_neg, quant, level):
    rng = np.random.default_rng(n_pos * 7 + n_neg + quant)
    pos = rng.normal(0.6, 1.0, n_pos).astype(np.float32)
    neg = rng.normal(-0.4, 1.2, n_neg).astype(np.float32)
    if quant:
        pos = (np.round(pos * quant / 8) * 8 / quant).astype(np.float32)
        neg = (np.round(neg * quant / 8) * 8 / quant).astype(np.float32)
    got = np.array(net.measures(_dev(pos), _dev(neg), recall_level=level))
    want = np.array(get_measures(pos, neg, recall_level=level))
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    assert got[2] == want[2]  # a ratio of exact counts


def test_measures_negate_matches_reference_call(net):
    """get_and_print_results evaluates -score (reference utils/detection_util.py:255)."""
    rng = np.random.default_rng(5)
    in_score = -rng.beta(5, 2, 4000).astype(np.float32)   # negated confidences, as stored
    out_score = -rng.beta(2, 3, 3000).astype(np.float32)
    got = np.array(net.measures(_dev(in_score), _dev(out_score), negate=True))
    np.testing.assert_allclose(got, np.array(get_measures(-in_score, -out_score)), rtol=0, atol=TOL)


def test_measures_degenerate_and_errors(net):
    import torch

    same = np.full(40, 0.25, np.float32)
    got = net.measures(_dev(same), _dev(same))
    assert got == tuple(float(v) for v in get_measures(same, same))  # 0.5, n_pos/n, 1.0
    with pytest.raises(RuntimeError):
        net.measures(torch.empty(0, device="cuda"), _dev(same))
    with pytest.raises(RuntimeError):
        net.measures(_dev(same), _dev(same), recall_level=1.5)


def test_end_to_end_device_metrics(net, capsys):
    """get_ood_scores_clip(device_out=True) + get_and_print_results(net=) == the host route."""
    from mcm_amd.detection import get_and_print_results, get_ood_scores_clip
    from mcm_amd.synth import SyntheticImageSet, SyntheticLoader

    geo = net.geo
    args = SimpleNamespace(ckpt="tiny", model="CLIP", score="MCM", T=1.0)
    labels = [f"thing{i}" for i in range(12)]
    id_loader = SyntheticLoader(SyntheticImageSet(150, geo.image_size, 12, ood=False, seed=1), 64)
    ood_loader = SyntheticLoader(SyntheticImageSet(90, geo.image_size, 12, ood=True, seed=2), 64)
    d_in = get_ood_scores_clip(args, net, id_loader, labels, device_out=True)
    d_out = get_ood_scores_clip(args, net, ood_loader, labels, device_out=True)
    h_in = get_ood_scores_clip(args, net, id_loader, labels)
    h_out = get_ood_scores_clip(args, net, ood_loader, labels)
    assert d_in.is_cuda and np.array_equal(d_in.cpu().numpy(), h_in)
    dev, host = ([], [], []), ([], [], [])
    get_and_print_results(args, None, d_in, d_out, *dev, net=net)
    get_and_print_results(args, None, h_in, h_out, *host)
    np.testing.assert_allclose(np.array(dev).ravel(), np.array(host).ravel(), rtol=0, atol=TOL)
    with pytest.raises(TypeError):
        get_and_print_results(args, None, d_in, d_out, [], [], [])


def test_cli_device_metrics_equal_host_metrics(tmp_path, monkeypatch):
    """eval_ood_detection.py end to end (reference eval_ood_detection.py:69-98): the device-metric
    default and --host-metrics (sklearn, the reference's route) write the same table."""
    import pandas as pd

    import eval_ood_detection as cli

    monkeypatch.chdir(tmp_path)
    common = ["--in_dataset", "ImageNet10", "--CLIP_ckpt", "ViT-B/32", "-b", "64", "--synthetic-n", "96"]
    cli.main(common + ["--name", "dev"])
    cli.main(common + ["--name", "host", "--host-metrics"])
    base = tmp_path / "results" / "ImageNet10" / "MCM"
    dev = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_dev" / "dev.csv", index_col=0)
    host = pd.read_csv(base / "CLIP_ViT-B/32_T_1_ID_host" / "host.csv", index_col=0)
    assert list(dev.index) == ["ImageNet20", "AVG"]
    assert dev.equals(host)
    assert 0.0 <= dev.loc["AVG", "AUROC"] <= 100.0
