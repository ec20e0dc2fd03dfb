"""gloo tests of the N>1 path on the CPU: shard → score → all-gather == single process, at world sizes 2, 3 and 8 (the
round-end scaling run is the first execution on 8 devices and cannot be rehearsed: its logic can) with set sizes 3 (most
ranks hold an EMPTY shard), 5 640 and 50 000 (BASELINE config 3's Textures / ImageNet-1k sizes: ragged last shards), and the
sharded threshold refinement incl. a window that falls entirely into one rank's shard."""
import os
import socket
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _StubNet:
    """Deterministic per-image score so shard order errors are visible."""

    def get_text_features(self, input_ids, attention_mask):
        return torch.zeros(input_ids.shape[0], 4)

    def score_images(self, images, text, T, score):
        return -(images.reshape(images.shape[0], -1)[:, :7].sum(dim=1)).float()


def _worker(rank, ws, port, n, bs, use_shard, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from mcm_amd.detection import get_ood_scores_clip
    from mcm_amd.dist import all_gather_histograms
    from mcm_amd.synth import SyntheticImageSet, SyntheticLoader, class_names

    dist.init_process_group("gloo", rank=rank, world_size=ws)
    ds = SyntheticImageSet(n, 8, 3, ood=False, seed=1)
    loader = SyntheticLoader(ds, bs)
    if not use_shard:  # generic loader without .shard(): batch-range split
        loader = types.SimpleNamespace(dataset=ds, __iter__=None)
        base = SyntheticLoader(ds, bs)

        class Plain:
            dataset = ds

            def __len__(self):
                return len(base)

            def __iter__(self):
                return iter(base)

        loader = Plain()
    args = types.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1)
    s = get_ood_scores_clip(args, _StubNet(), loader, class_names(3))
    hist = all_gather_histograms(torch.from_numpy(s[rank::ws].copy()), np.linspace(-40, 40, 9))
    q.put((rank, s, hist))
    dist.destroy_process_group()


def _run_ranks(target, ws, args, timeout=240):
    """`target(rank, ws, port, *args, q)` as `ws` spawned processes; their queue items sorted by rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, ws, port) + tuple(args) + (q,)) for r in range(ws)]
    for p in procs:
        p.start()
    got = [q.get(timeout=timeout) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(got, key=lambda t: t[0])


@pytest.mark.parametrize("ws,n,bs,use_shard", [(3, 3, 4, True), (8, 3, 2, True), (8, 3, 2, False), (3, 5640, 512, True),
                                               (8, 5640, 512, True), (8, 5640, 100, False), (3, 50000, 512, True),
                                               (8, 50000, 512, True), (8, 7, 16, False), (3, 40, 12, False)])
def test_many_rank_gather_equals_single(ws, n, bs, use_shard):
    """World sizes 3 and 8: empty shards (n = 3 on 8 ranks: five ranks score nothing and still take part in the collective),
    ragged shards (5 640 = 8 x 705; 50 000 / 3), opaque loaders split by batch ranges (fewer batches than ranks)."""
    import types as _t

    from mcm_amd.detection import get_ood_scores_clip
    from mcm_amd.dist import shard_range
    from mcm_amd.synth import SyntheticImageSet, SyntheticLoader, class_names

    ds = SyntheticImageSet(n, 8, 3, ood=False, seed=1)
    args = _t.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1)
    want = get_ood_scores_clip(args, _StubNet(), SyntheticLoader(ds, bs), class_names(3))
    assert len({shard_range(n, r, ws) for r in range(ws)}) >= min(ws, 2) and shard_range(n, ws - 1, ws)[1] == n
    for rank, s, hist in _run_ranks(_worker, ws, (n, bs, use_shard)):
        assert s.shape == (n,) and np.array_equal(s, want), (rank, s[:8], want[:8])
        assert hist.sum() == n  # the ranks' histograms sum to the whole dataset


@pytest.mark.parametrize("n,bs,use_shard", [(37, 5, True), (37, 5, False), (16, 8, True), (3, 4, True),
                                            (40, 12, False), (1000, 512, False), (7, 16, False)])
def test_two_rank_gather_equals_single(n, bs, use_shard):
    import types as _t

    from mcm_amd.detection import get_ood_scores_clip
    from mcm_amd.synth import SyntheticImageSet, SyntheticLoader, class_names

    ds = SyntheticImageSet(n, 8, 3, ood=False, seed=1)
    args = _t.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1)
    want = get_ood_scores_clip(args, _StubNet(), SyntheticLoader(ds, bs), class_names(3))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, bs, use_shard, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, s, hist in got:
        assert s.shape == (n,) and np.array_equal(s, want), (rank, s, want)
        assert hist.sum() == n  # the two ranks' histograms sum to the whole dataset


class _CountingSet(torch.utils.data.Dataset):
    """Map-style dataset that remembers which samples were actually produced (decoded)."""

    def __init__(self, n):
        self.n, self.touched = n, []

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        self.touched.append(int(i))
        g = torch.Generator().manual_seed(1000 + int(i))
        return torch.randn(3, 8, 8, generator=g), int(i) % 3


class _StubMahaNet(_StubNet):
    def get_image_features(self, pixel_values):
        return pixel_values.reshape(pixel_values.shape[0], -1)[:, :6].float()

    def maha_prepare(self, mu, prec):
        return {"mu": mu.double(), "prec": prec.double()}

    def maha_scores(self, f, st):
        d = f.double()[:, None, :] - st["mu"][None]
        return (0.5 * torch.einsum("bcp,pq,bcq->bc", d, st["prec"], d)).min(dim=1).values.float()


def _worker_torch_loader(rank, ws, port, n, bs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    import types

    import torch.distributed as dist

    from mcm_amd.detection import get_Mahalanobis_score, get_ood_scores_clip
    from mcm_amd.synth import class_names

    dist.init_process_group("gloo", rank=rank, world_size=ws)
    ds = _CountingSet(n)
    loader = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False)
    args = types.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1, normalize=False, batch_size=bs)
    s = get_ood_scores_clip(args, _StubNet(), loader, class_names(3))
    touched = sorted(set(ds.touched))
    mu = torch.arange(18, dtype=torch.float32).reshape(3, 6) / 10
    m_id = get_Mahalanobis_score(args, _StubMahaNet(), loader, mu, torch.eye(6), in_dist=True)
    m_ood = get_Mahalanobis_score(args, _StubMahaNet(), loader, mu, torch.eye(6), in_dist=False)
    q.put((rank, s, touched, m_id, m_ood))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bs", [(37, 5), (64, 16), (9, 4)])
def test_two_rank_torch_dataloader_is_sharded_by_index_before_decode(n, bs):
    """A reference-style torch DataLoader (no `.shard`): each rank re-builds it over its own contiguous index range, so
    a rank only ever produces (decodes) its own samples — round 3 made every rank iterate every batch.  Also the
    Mahalanobis scorer under world_size 2 (round 3 refused it), incl. the reference's rule that an OOD set's trailing
    partial batch is not scored."""
    import types as _t

    from mcm_amd.detection import get_Mahalanobis_score, get_ood_scores_clip
    from mcm_amd.dist import shard_range
    from mcm_amd.synth import class_names

    args = _t.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1, normalize=False, batch_size=bs)
    one = torch.utils.data.DataLoader(_CountingSet(n), batch_size=bs, shuffle=False)
    want = get_ood_scores_clip(args, _StubNet(), one, class_names(3))
    mu = torch.arange(18, dtype=torch.float32).reshape(3, 6) / 10
    want_id = get_Mahalanobis_score(args, _StubMahaNet(), one, mu, torch.eye(6), in_dist=True)
    want_ood = get_Mahalanobis_score(args, _StubMahaNet(), one, mu, torch.eye(6), in_dist=False)
    assert want_id.shape == (n,) and want_ood.shape == ((n // bs) * bs,)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_torch_loader, args=(r, 2, port, n, bs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, s, touched, m_id, m_ood in got:
        assert np.array_equal(s, want), rank
        lo, hi = shard_range(n, rank, 2)
        assert touched == list(range(lo, hi)), (rank, touched)  # (the maha runs re-touch a subset of the same range)
        assert np.array_equal(m_id, want_id) and np.array_equal(m_ood, want_ood), rank


# ---- threshold refinement under world_size 2: the re-scoring is sharded, the result is the 1-rank result


class _NoisyNet(_StubNet):
    """The '16-bit arm': the stub's score plus a deterministic per-image perturbation."""

    def score_images(self, images, text, T, score):
        s = super().score_images(images, text, T, score)
        noise = torch.sin(images.reshape(images.shape[0], -1)[:, 7:13].sum(dim=1) * 1e3) * 0.3
        return (s + noise).float()


class _CountingScorer(_StubNet):
    """The 'better arm' behind mcm_amd.refine.Rescorer: the exact stub score; counts the images it is handed."""

    max_batch = 16

    def __init__(self):
        self.seen = 0

    def score_images(self, images, text, T, score, out=None):
        self.seen += int(images.shape[0])
        return super().score_images(images, text, T, score)


def _refined_scores(n, bs):
    """ID set + one OOD set scored by the noisy arm through get_ood_scores_clip (sharded when a group is up), then threshold
    refinement with the sharded Rescorer.  Returns (id scores, ood scores, refiner stats, images this rank re-scored)."""
    import types as _t

    from mcm_amd.detection import get_ood_scores_clip
    from mcm_amd.refine import Rescorer, ThresholdRefiner
    from mcm_amd.synth import SyntheticImageSet, SyntheticLoader, class_names

    args = _t.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1)
    loaders = {"id": SyntheticLoader(SyntheticImageSet(n, 8, 3, ood=False, seed=1), bs),
               "ood": SyntheticLoader(SyntheticImageSet(n // 2, 8, 3, ood=True, seed=2), bs)}
    scores = {k: torch.from_numpy(get_ood_scores_clip(args, _NoisyNet(), v, class_names(3))) for k, v in loaders.items()}
    scorer = _CountingScorer()
    r = Rescorer(scorer, torch.zeros(3, 4), loaders, 1.0, "MCM")
    ref = ThresholdRefiner(r, calib=8)
    ref.fit_id(scores["id"])
    ref.apply("ood", scores["ood"])
    return scores["id"].numpy(), scores["ood"].numpy(), ref.stats, r.scored_here, scorer.seen


def _worker_refine(rank, ws, port, n, bs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=ws)
    sid, sood, st, here, seen = _refined_scores(n, bs)
    q.put((rank, sid, sood, st["rescored"], st["threshold"], here, seen))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bs", [(600, 64), (257, 50)])
def test_two_rank_refinement_equals_single_and_is_sharded(n, bs):
    from mcm_amd.dist import shard_range

    want_id, want_ood, st, here, seen = _refined_scores(n, bs)
    assert st["rescored_total"] > 16 and here == st["rescored_total"] == seen   # one rank: it re-scores the whole window
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_refine, args=(r, 2, port, n, bs, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, sid, sood, rescored, thr, here_r, seen_r in got:
        np.testing.assert_array_equal(sid, want_id)       # every rank ends with the 1-rank refined scores, bit for bit
        np.testing.assert_array_equal(sood, want_ood)
        assert rescored == st["rescored"] and thr == st["threshold"]
        assert here_r == seen_r < st["rescored_total"]    # ... having re-scored only part of the window
    assert sum(g[5] for g in got) == st["rescored_total"]  # the parts add up to the window
    # rank 0's share of the calibration images: those of its own index shard
    lo, hi = shard_range(n, 0, 2)
    assert got[0][5] >= min(8, hi - lo) and got[1][5] > 0


class _RampSet(torch.utils.data.Dataset):
    """Map-style set whose stub score RISES with the index (score_i = step * i + a bounded pseudo-noise term in the noisy arm): the
    FPR95 threshold — the score below which 95 % of the set lies — then sits at index 0.95 n, i.e. the refinement window falls
    into ONE contiguous index shard (the last rank's at world size 8)."""

    def __init__(self, n, step=1.0):
        self.n, self.step = n, step

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        i = int(i)
        x = torch.zeros(3, 8, 8)
        x.view(-1)[0] = -self.step * i                      # _StubNet: score = -(sum of the first 7 values) = step * i
        x.view(-1)[7] = float((i * 7919) % 1000) / 1000.0    # _NoisyNet: noise = 0.3 sin(1e3 * sum of values 7..12)
        return x, i % 3


def _ramp_refined(n, bs):
    import types as _t

    from mcm_amd.detection import get_ood_scores_clip
    from mcm_amd.refine import Rescorer, ThresholdRefiner
    from mcm_amd.synth import class_names

    args = _t.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1, batch_size=bs)
    loaders = {"id": torch.utils.data.DataLoader(_RampSet(n), batch_size=bs, shuffle=False),
               "ood": torch.utils.data.DataLoader(_RampSet(n // 2, step=2.0), batch_size=bs, shuffle=False)}
    scores = {k: torch.from_numpy(get_ood_scores_clip(args, _NoisyNet(), v, class_names(3))) for k, v in loaders.items()}
    scorer = _CountingScorer()
    r = Rescorer(scorer, torch.zeros(3, 4), loaders, 1.0, "MCM")
    ref = ThresholdRefiner(r, calib=64)
    ref.fit_id(scores["id"])
    before = r.scored_here
    ref.apply("ood", scores["ood"])
    return scores["id"].numpy(), scores["ood"].numpy(), ref.stats, before, r.scored_here - before


def _worker_ramp(rank, ws, port, n, bs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=ws)
    sid, sood, st, here_id, here_ood = _ramp_refined(n, bs)
    q.put((rank, sid, sood, st["rescored"], st["threshold"], here_id, here_ood))
    dist.destroy_process_group()


@pytest.mark.parametrize("ws", [3, 8])
def test_refinement_window_inside_one_rank(ws):
    """World sizes 3 and 8, a reference-style torch DataLoader: the ID window sits at index ~0.95 n, the OOD window at ~0.475 n
    of its set — each inside ONE rank's shard.  Every rank ends with the 1-rank refined scores bit for bit; only the calibration
    runs are spread over the ranks, the window itself is re-scored by its owner alone; ranks with nothing to re-score in a set
    still take part in the all-reduce."""
    from mcm_amd.dist import shard_range

    n, bs = 4096, 128
    want_id, want_ood, st, here_id, here_ood = _ramp_refined(n, bs)
    n_win_id = st["rescored"]["id"] - st["calibration_images"]      # window images beyond the calibration runs (disjoint here)
    assert st["calibration_images"] == 64 and 0 < n_win_id <= 16 and 0 < st["rescored"]["ood"] <= 16, st
    got = _run_ranks(_worker_ramp, ws, (n, bs))
    owner_id = [r for r in range(ws) if shard_range(n, r, ws)[0] <= int(0.95 * n) < shard_range(n, r, ws)[1]][0]
    for rank, sid, sood, rescored, thr, h_id, h_ood in got:
        np.testing.assert_array_equal(sid, want_id)
        np.testing.assert_array_equal(sood, want_ood)
        assert rescored == st["rescored"] and thr == st["threshold"]
    assert sum(g[5] for g in got) == st["rescored"]["id"] and sum(g[6] for g in got) == st["rescored"]["ood"]
    # the 64 calibration images (one run of 64 here) belong to one shard, the window to the shard that holds index 0.95 n
    assert got[owner_id][5] >= n_win_id and sum(1 for g in got if g[5] > 0) <= 2
    assert sum(1 for g in got if g[6] > 0) == 1            # the OOD window: one owner, everybody else contributes zeros
