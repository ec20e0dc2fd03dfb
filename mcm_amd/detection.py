"""The hot path's host side: `get_ood_scores_clip` and its reporting helpers, re-written
from scratch with the reference's signatures (utils/detection_util.py:209-265).

Differences from the reference that do not change results:
  * the prompt bank is tokenised and encoded ONCE per call instead of once per batch
    (reference :228-231 is loop-invariant);
  * the per-batch body runs as one fused native call (`net.score_images`), so only [b]
    scores exist per batch and they stay in HBM until the dataset is done (the reference
    copies the whole [b,K] softmax to the host every batch, :236);
  * under torch.distributed (one process per GPU) each rank scores a contiguous shard of the
    batches and the shards are all-gathered, so every rank returns the full vector in the
    reference's sample order.
"""
from __future__ import annotations

import numpy as np

from . import dist as mdist
from .config import SCORE_KINDS
from .metrics import get_measures
from .tokenizer import load_tokenizer

PROMPT = "a photo of a {c}"  # reference utils/detection_util.py:228 (no trailing period)


def _tokenizer(args, net=None):
    """args.tokenizer_dir (additive) wins over args.ckpt.  The hash stand-in is allowed only with synthetic
    weights: the net records that itself (`net.synthetic_weights`, set by build_model / NativeCLIP), so a
    reference-shaped caller that has only `args.ckpt` and hands over a NativeCLIP built from a real checkpoint
    is refused instead of getting meaningless ids; a net that does not say falls back to `args.weights`."""
    src = getattr(args, "tokenizer_dir", None) or getattr(args, "ckpt", "")
    synthetic = getattr(net, "synthetic_weights", None)
    if synthetic is None:
        synthetic = not getattr(args, "weights", None)
    return load_tokenizer(src, allow_hash=bool(synthetic))


def encode_prompt_bank(args, net, test_labels):
    """`text_features` of the reference (:228-231): K prompts → [K,P] unit-norm fp32."""
    tokenizer = _tokenizer(args, net)
    text_inputs = tokenizer([PROMPT.format(c=c) for c in test_labels], padding=True, return_tensors="pt")
    return _unit_text_features(net, text_inputs)


# A small built-in template set for the prompt-ensemble bank (BASELINE config 5).  The reference
# ships OpenAI's 80 ImageNet templates as data but never uses them; pass your own list (e.g. read
# from that file) through `templates=` to reproduce the 80-template recipe.
DEFAULT_TEMPLATES = ["a photo of a {c}", "a blurry photo of a {c}", "a close-up photo of a {c}",
                     "a photo of the {c}", "a drawing of a {c}", "a bright photo of a {c}",
                     "a cropped photo of a {c}", "a photo of a small {c}", "a photo of a large {c}"]


def encode_prompt_ensemble(args, net, test_labels, templates=None):
    """CLIP zero-shot ensemble bank: per class, encode every template, normalise, average,
    re-normalise → [K,P].  Still a [K,P] bank, so the scoring path is unchanged."""
    templates = list(templates or DEFAULT_TEMPLATES)
    labels = list(test_labels)
    tokenizer = _tokenizer(args, net)
    prompts = [t.format(c=c) if "{c}" in t else t.format(c) for c in labels for t in templates]  # class-major
    tok = tokenizer(prompts, padding=True, return_tensors="pt")
    feats = _unit_text_features(net, tok)
    return net.reduce_bank(feats, len(labels), len(templates))


def _unit_text_features(net, tok):
    """`get_text_features(...).float()` followed by `/= norm` (reference :229-231).  A NativeCLIP fuses
    the normalisation; any other `net` honouring the HF contract is normalised here."""
    import inspect

    try:
        fused = "normalize" in inspect.signature(net.get_text_features).parameters
    except (TypeError, ValueError):
        fused = False
    if fused:
        return net.get_text_features(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"],
                                     normalize=True)
    f = net.get_text_features(input_ids=tok["input_ids"], attention_mask=tok["attention_mask"]).float()
    return f / f.norm(dim=-1, keepdim=True).clamp_min(1e-30)


def prompt_bank(args, net, test_labels):
    """The [K,P] unit-norm bank for (labels, templates, tokenizer source), encoded once per `net`: a CLI run
    scores one ID and four OOD sets against the same bank (the reference re-encodes it every BATCH, :228-231;
    round 2 re-encoded it every dataset — 5 s per call with 80 templates x 1000 classes).  The key holds
    everything the bank depends on; T and the score kind do not enter it."""
    templates = getattr(args, "templates", None)
    key = (tuple(str(c) for c in test_labels), tuple(templates) if templates else None,
           getattr(args, "tokenizer_dir", None) or getattr(args, "ckpt", ""))
    cache = getattr(net, "_bank_cache", None)
    if cache is None:
        try:
            cache = net._bank_cache = {}
        except AttributeError:  # a net that refuses attributes: no caching
            cache = {}
    if key not in cache:
        cache.clear()  # one bank at a time (a bank is K x P floats; keep the handle small)
        cache[key] = (encode_prompt_ensemble(args, net, test_labels, templates) if templates
                      else encode_prompt_bank(args, net, test_labels))
    return cache[key]


def read_templates(path):
    """Prompt templates from a user file (`--templates`).  Two formats:
      * a text file, one template per line, the class name marked `{c}` or `{}`;
      * a Python file in the style of the reference's utils/imagenet_templates.py (a list of
        `lambda c: f'a bad photo of a {c}.'`): the f-string bodies of the file's FIRST list are
        extracted, nothing is executed."""
    import re

    text = open(path, encoding="utf-8").read()
    if path.endswith(".py"):
        m = re.search(r"=\s*\[(.*?)^\]", text, re.S | re.M)  # the first list only (the 80 templates); the
        text = m.group(1) if m else text                      # file's later subsets repeat entries
        out = [m.group(2) for m in re.finditer(r"""f(['"])((?:(?!\1).)*\{c\}(?:(?!\1).)*)\1""", text)]
    else:
        out = [ln.strip() for ln in text.splitlines() if ln.strip() and not ln.lstrip().startswith("#")]
    if not out or not all(("{c}" in t) or ("{}" in t) for t in out):
        raise ValueError(f"{path}: no templates found, or a template without a {{c}} / {{}} placeholder")
    return out


def get_ood_scores_clip(args, net, loader, test_labels, in_dist=False, device_out=False):
    """Scores every sample of `loader` against the concept bank `test_labels`.

    Same contract as reference utils/detection_util.py:209-249: reads `args.ckpt`,
    `args.model`, `args.score`, `args.T`; `loader` yields `(images[b,3,S,S] fp32, labels)`
    in dataset order; returns float32 ndarray `[len(loader.dataset)]` of *negated*
    confidences (lower = more ID) for MCM / max-logit / energy / var and the entropy for
    'entropy'.  `in_dist` and the labels are unused, as in the reference.
    `device_out=True` (an addition) returns the same vector as a device tensor instead, for
    `get_and_print_results(..., net=net)` to evaluate without a host round trip.
    """
    import torch

    if getattr(args, "model", "CLIP") != "CLIP":
        raise ValueError(f"unsupported --model {args.model!r} (the reference only defines CLIP)")
    if args.score not in SCORE_KINDS:
        raise ValueError(f"unsupported --score {args.score!r} for get_ood_scores_clip")
    if not hasattr(net, "score_images"):
        raise TypeError("net must be a mcm_amd NativeCLIP (fused score_images path); "
                        "there is no eager fallback")
    rank, ws = mdist.world()
    n_total = len(loader.dataset)
    with torch.no_grad():
        text_features = prompt_bank(args, net, test_labels)
        by_index = True  # this rank's shard is the contiguous index range shard_range gives
        mine = lambda i: True  # noqa: E731
        batches = loader
        if ws > 1:
            lo, hi = mdist.shard_range(n_total, rank, ws)
            batches = shard_loader(loader, lo, hi)
            if batches is None:  # an opaque iterable: batch-range shards (every rank still pays its decode)
                by_index = False
                nb = len(loader)
                blo, bhi = mdist.shard_range(nb, rank, ws)
                batches = loader
                mine = lambda i: blo <= i < bhi  # noqa: E731
        parts = []
        for batch_idx, (images, _labels) in enumerate(batches):
            if not mine(batch_idx):
                continue
            parts.append(net.score_images(images, text_features, float(args.T), args.score))
        local = torch.cat(parts) if parts else torch.empty(0, dtype=torch.float32,
                                                           device=text_features.device)
        if mdist.group_active():  # (also a 1-rank group: the collective really runs — RCCL on a GPU box)
            if by_index:
                full = mdist.all_gather_scores(local, n_total)
            else:  # batch-range shards of an opaque loader: sizes follow the batch split
                full = _gather_batch_shards(local, n_total, ws)
        else:
            full = local
    if device_out:
        return full.detach()[:n_total]
    return full.detach().cpu().numpy().astype(np.float32, copy=False)[:n_total].copy()


def shard_loader(loader, lo: int, hi: int):
    """A loader over samples [lo, hi) of `loader.dataset`, in order — a rank's shard BEFORE any decode happens.
    The build's own loaders have `.shard`; a torch-style DataLoader (map-style `dataset`, `batch_size`, the reference's
    kind: utils/train_eval_util.py:96-146, shuffle=False) is re-built over `Subset(dataset, range(lo, hi))` with the same
    batch size, workers and collate function; anything else returns None (the caller falls back to skipping batches)."""
    if hasattr(loader, "shard"):
        return loader.shard(lo, hi)
    ds, bs = getattr(loader, "dataset", None), getattr(loader, "batch_size", None)
    if ds is None or not bs or not hasattr(ds, "__getitem__"):
        return None
    try:
        from torch.utils.data import DataLoader, Subset
    except Exception:
        return None
    if not isinstance(loader, DataLoader):
        return None
    kw = dict(batch_size=bs, shuffle=False, num_workers=loader.num_workers, collate_fn=loader.collate_fn,
              pin_memory=loader.pin_memory, drop_last=False)
    if loader.num_workers > 0:
        kw.update(prefetch_factor=loader.prefetch_factor, persistent_workers=False)
    return DataLoader(Subset(ds, range(lo, hi)), **kw)


def get_mean_prec(args, net, train_loader):
    """Mahalanobis fit: (classwise_mean [n_cls, feat_dim], precision [feat_dim, feat_dim]), with the
    results and side effects of reference utils/detection_util.py:146-174.  Two behaviours of the
    reference that a drop-in has to reproduce, stated as the maths they amount to:

      * the reference records, per label, the index of the BATCH a sample came from and then uses those
        numbers as row indices into the matrix of all features (:159-160,164-165).  So class c's "mean"
        is  sum_b n[c,b] * F[b] / sum_b n[c,b],  n[c,b] = samples of class c in batch b, F[b] = the b-th
        feature ROW.  It is computed here as one [n_cls, n_batches] x [n_batches, P] product in float64;
      * one covariance over all features, inverted in float64, cast to float32 (:168-169).

    Features come from `net.get_image_features(pixel_values=...)` — the plain HF contract (raw
    projections; L2-normalised when args.normalize).  Writes the reference's two .pt files."""
    import os

    import torch

    feats, counts = [], []
    with torch.no_grad():
        for images, labels in train_loader:
            f = net.get_image_features(pixel_values=images).float()
            if args.normalize:
                f = f / f.norm(dim=-1, keepdim=True)
            feats.append(f.cpu())
            # labels >= n_cls are ignored, like the reference's per-class loop (:161-166) never visits them
            lab = labels.detach().cpu().numpy() if hasattr(labels, "detach") else np.asarray(labels)
            counts.append(np.bincount(lab.astype(np.int64).reshape(-1), minlength=args.n_cls)[: args.n_cls])
    F = torch.cat(feats)                                    # [n, P] float32, dataset order
    n_cb = torch.from_numpy(np.stack(counts, axis=1).astype(np.float64))  # [n_cls, n_batches]
    rows = F[: n_cb.shape[1]].double()                       # the rows the reference's indices select
    classwise_mean = ((n_cb @ rows) / n_cb.sum(dim=1, keepdim=True)).float()
    if args.normalize:
        classwise_mean = classwise_mean / classwise_mean.norm(dim=-1, keepdim=True)
    precision = torch.linalg.inv(torch.cov(F.T.double())).float()
    print(f"cond number: {torch.linalg.cond(precision)}")
    tdir = getattr(args, "template_dir", None)
    if tdir:
        os.makedirs(tdir, exist_ok=True)
        for what, t in (("classwise_mean", classwise_mean), ("precision", precision)):
            torch.save(t, os.path.join(tdir, maha_file_name(args, what)))
    return classwise_mean, precision


def maha_file_name(args, what):
    """File names of the stored Mahalanobis statistics (reference :171-172, eval_ood_detection.py:77-78)."""
    return f"{args.model}_{what}_{args.in_dataset}_{args.max_count}_{args.normalize}.pt"


def get_Mahalanobis_score(args, net, test_loader, classwise_mean, precision, in_dist=True):
    """`--score maha`: per sample min_c 0.5 (f - mu_c) P (f - mu_c)^T — what reference
    utils/detection_util.py:176-207 returns (the negated max of -0.5 d_c) — with the per-class loop of
    torch.mm pairs replaced by the native kernels (one quadratic form + C dot products per image).
    Keeps the reference's loop rule that for OOD sets (`in_dist=False`) iteration stops at batch
    `len(dataset) // batch_size`, i.e. a trailing partial batch is NOT scored (:185-186)."""
    import torch

    state = net.maha_prepare(classwise_mean, precision)
    total_len = len(test_loader.dataset)
    # the samples the reference scores: everything for the ID set, whole batches only for an OOD set
    n_scored = total_len if in_dist else min(total_len, (total_len // args.batch_size) * args.batch_size)
    rank, ws = mdist.world()
    batches, lo, hi = test_loader, 0, n_scored
    if ws > 1:  # image-sharded like get_ood_scores_clip: a contiguous index range per rank, all-gathered at the end
        lo, hi = mdist.shard_range(n_scored, rank, ws)
        batches = shard_loader(test_loader, lo, hi)
        if batches is None:
            raise TypeError("--score maha under world_size > 1 needs a loader that can be sharded by index")
    out, seen = [], 0
    with torch.no_grad():
        for images, _labels in batches:
            if seen >= hi - lo:
                break
            images = images[: hi - lo - seen]
            features = net.get_image_features(pixel_values=images).float()
            if args.normalize:
                features = features / features.norm(dim=-1, keepdim=True)
            out.append(net.maha_scores(features, state))
            seen += images.shape[0]
    res = torch.cat(out) if out else torch.empty(0, device=state["prec"].device)
    if mdist.group_active():
        res = mdist.all_gather_scores(res, n_scored)
    return res.cpu().numpy().astype(np.float32)


def _gather_batch_shards(local, n_total, ws):
    """All-gather score shards whose sizes only the owning rank knows (batch-range split of a generic
    loader: the last batch may be short, batch sizes need not be uniform).  Counts are exchanged first;
    the payload buffer is sized by the largest shard and sliced by the true counts."""
    import torch
    import torch.distributed as dist

    home = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:  # ranks sharing a device (logic checks): host bounce
        local = local.cpu()
    cnt = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    cnts = torch.empty(ws, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(cnts, cnt)
    cnts = [int(c) for c in cnts.cpu()]
    if sum(cnts) != n_total:
        raise RuntimeError(f"rank shards hold {sum(cnts)} scores, the dataset has {n_total}")
    cap = max(max(cnts), 1)
    buf = torch.zeros(cap, dtype=torch.float32, device=local.device)
    buf[: local.numel()] = local
    out = torch.empty(ws * cap, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * cap: r * cap + cnts[r]] for r in range(ws)]).to(home)


def print_measures(log, auroc, aupr, fpr, method_name="Ours", recall_level=0.95):
    """Same output format as reference utils/detection_util.py:37-45."""
    pct = int(100 * recall_level)
    if log is None:
        print("FPR{:d}:\t\t\t{:.2f}".format(pct, 100 * fpr))
        print("AUROC: \t\t\t{:.2f}".format(100 * auroc))
        print("AUPR:  \t\t\t{:.2f}".format(100 * aupr))
    else:
        log.debug("\t\t\t\t" + method_name)
        log.debug("  FPR{:d} AUROC AUPR".format(pct))
        log.debug("& {:.2f} & {:.2f} & {:.2f}".format(100 * fpr, 100 * auroc, 100 * aupr))


def get_and_print_results(args, log, in_score, out_score, auroc_list, aupr_list, fpr_list, net=None):
    """Reference utils/detection_util.py:253-265: the scores are negated confidences, so the
    metrics are taken on their negation with ID as the positive class.  Device tensors (from
    `get_ood_scores_clip(..., device_out=True)`) are evaluated by the native metric kernels
    of `net`; ndarrays go through the host implementation exactly like the reference."""
    if hasattr(in_score, "is_cuda"):
        if net is None or not hasattr(net, "measures"):
            raise TypeError("device score tensors need net= (a NativeCLIP) for the metric kernels")
        auroc, aupr, fpr = net.measures(in_score, out_score, negate=True)
        in_score, out_score = in_score[:3].cpu().numpy(), out_score[:3].cpu().numpy()
    else:
        auroc, aupr, fpr = get_measures(-in_score, -out_score)
    print(f"in score samples (random sampled): {in_score[:3]}, out score samples: {out_score[:3]}")
    auroc_list.append(auroc)
    aupr_list.append(aupr)
    fpr_list.append(fpr)
    print_measures(log, auroc, aupr, fpr, args.score)
