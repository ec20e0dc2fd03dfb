"""world_size-2 gloo test of the N>1 path: shard → score → all-gather == single process."""
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
