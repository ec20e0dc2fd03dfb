"""Generate tests/golden/*.npz — the vectors that pin the oracle (and through it the HIP
path) to the reference's arithmetic.

Run ONLY in the build container (`python tests/golden/make_golden.py`): it imports the
reference from /root/reference and Hugging Face transformers (the un-pinned third-party
library the reference delegates every model op to, utils/train_eval_util.py:9,23;
installed version recorded in the fixture).  Nothing here travels to the GPU box except
the .npz outputs: inputs are regenerated from seeds by mcm_amd.synth / mcm_amd.weights.

What is captured
  clip_<geo>.npz     HF CLIPModel on seeded weights/inputs: per-layer hidden states
                     (vision + text), pooled/projected features, for geo in
                     {tiny, B16-2L (sampled rows), ViT-B/16 full depth}.
  scores_tiny.npz    the reference's own get_ood_scores_clip (utils/detection_util.py:209-249)
                     driven end-to-end on the tiny geometry for all five --score kinds at
                     T=1 and T=2, under the four shims of SURVEY.md §8c.
  measures.npz       the reference's get_measures (utils/detection_util.py:108-119) on
                     synthetic score sets incl. ties and the SURVEY KAT.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import transformers  # noqa: E402  (must precede the torchvision stubs)
from transformers import CLIPModel  # noqa: E402

from mcm_amd.config import geometry  # noqa: E402
from mcm_amd.synth import make_pixels, make_token_ids  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def hf_model(geo, sd):
    m = CLIPModel(geo.hf_configs()).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()},
                                            strict=False)
    assert not unexpected, unexpected
    assert all(("position_ids" in k) or k == "logit_scale" for k in missing), missing
    return m


class Net4x:
    """Shim (3) of SURVEY §8c: transformers-4.x tensor-returning contract."""

    def __init__(self, m):
        self.m = m

    def eval(self):
        return self

    def get_image_features(self, pixel_values):
        return self.m.get_image_features(pixel_values=pixel_values).pooler_output

    def get_text_features(self, input_ids, attention_mask):
        return self.m.get_text_features(input_ids=input_ids,
                                        attention_mask=attention_mask).pooler_output


def capture_clip(name, n_img, n_txt, sample_rows=None):
    geo = geometry(name)
    sd = synth_state_dict(geo, seed=0)
    m = hf_model(geo, sd)
    px, _ = make_pixels(n_img, geo.image_size, 10, ood=False, seed=1)
    ids, mask = make_token_ids(n_txt, seed=2)
    out = {"transformers_version": np.array(transformers.__version__),
           "n_img": np.array(n_img), "n_txt": np.array(n_txt)}
    v = m.vision_model(pixel_values=torch.from_numpy(px), output_hidden_states=True)
    # hidden_states[0] is the raw embedding output (before pre_layrnorm) in HF; the towers'
    # layer inputs/outputs follow.  Recompute stage 0 = after pre_layrnorm explicitly.
    emb = m.vision_model.embeddings(torch.from_numpy(px))
    h0 = m.vision_model.pre_layrnorm(emb)
    hs = [h0]
    x = h0
    for layer in m.vision_model.encoder.layers:
        x = layer(x, attention_mask=None)
        hs.append(x)
    rows = slice(None) if sample_rows is None else sample_rows
    for i, h in enumerate(hs):
        out[f"v_hidden_{i}"] = h[:, rows, :].numpy()
    if sample_rows is not None:
        out["v_rows"] = np.asarray(sample_rows)
    feats = m.get_image_features(pixel_values=torch.from_numpy(px)).pooler_output
    out["image_features"] = feats.numpy()
    assert torch.allclose(v.pooler_output, m.vision_model.post_layernorm(hs[-1][:, 0, :]))
    # text tower
    tin = torch.from_numpy(ids)
    t = m.text_model(input_ids=tin, attention_mask=torch.from_numpy(mask),
                     output_hidden_states=True)
    for i, h in enumerate(t.hidden_states):
        out[f"t_hidden_{i}"] = h.numpy() if sample_rows is None else h[:, :4, :].numpy()
    out["text_features"] = m.get_text_features(
        input_ids=tin, attention_mask=torch.from_numpy(mask)).pooler_output.numpy()
    # padding-invariance KAT: no mask → same pooled features
    nomask = m.get_text_features(input_ids=tin).pooler_output.numpy()
    out["text_features_nomask_maxdiff"] = np.array(np.abs(nomask - out["text_features"]).max())
    np.savez_compressed(os.path.join(HERE, f"clip_{name.replace('/', '_')}.npz"), **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.ndim})
    return m, geo


def load_reference_detection_util():
    """Shim (1): stub torchvision, load the file directly (utils/__init__ pulls torchvision
    + dataloaders)."""
    for mod in ("torchvision", "torchvision.datasets", "torchvision.transforms"):
        sys.modules.setdefault(mod, types.ModuleType(mod))
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    spec = importlib.util.spec_from_file_location(
        "ref_detection_util", "/root/reference/utils/detection_util.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def capture_scores(ref, m, geo):
    K, n_id, n_ood, bs = 10, 48, 40, 16
    ids, mask = make_token_ids(K, seed=2)

    class FakeTok:  # shim (4): no vocab files exist; honour the call contract only
        @classmethod
        def from_pretrained(cls, ckpt):
            return cls()

        def __call__(self, texts, padding=True, return_tensors="pt"):
            assert len(texts) == K
            return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}

    ref.CLIPTokenizer = FakeTok
    torch.Tensor.cuda = lambda self, *a, **k: self  # shim (2): no GPU here

    class DS:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    class Loader:
        def __init__(self, n, ood):
            self.dataset, self.ood = DS(n), ood

        def __len__(self):
            return -(-self.dataset.n // bs)

        def __iter__(self):
            for s in range(0, self.dataset.n, bs):
                n = min(bs, self.dataset.n - s)
                px, lab = make_pixels(n, geo.image_size, K, ood=self.ood, seed=1, start=s)
                yield torch.from_numpy(px), torch.from_numpy(lab)

    out = {"K": np.array(K), "n_id": np.array(n_id), "n_ood": np.array(n_ood),
           "batch": np.array(bs)}
    labels = [f"concept{k:04d}" for k in range(K)]
    net = Net4x(m)
    for score in ("MCM", "max-logit", "energy", "entropy", "var"):
        for T in (1, 2):
            args = types.SimpleNamespace(ckpt="unused", model="CLIP", score=score, T=T)
            s_in = ref.get_ood_scores_clip(args, net, Loader(n_id, False), labels, in_dist=True)
            s_out = ref.get_ood_scores_clip(args, net, Loader(n_ood, True), labels)
            out[f"{score}_T{T}_in"] = np.asarray(s_in)
            out[f"{score}_T{T}_out"] = np.asarray(s_out)
            if score == "MCM":
                out[f"measures_T{T}"] = np.array(ref.get_measures(-s_in, -s_out), dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "scores_tiny.npz"), **out)
    print("scores", {k: (v.shape, v.dtype) for k, v in out.items() if v.ndim})


def capture_maha(ref, m, geo):
    """The reference's own get_mean_prec / get_Mahalanobis_score (utils/detection_util.py:146-207) on
    the tiny model: class means, precision, and the scores of an ID set and an OOD set (the OOD call
    drops the trailing partial batch, as the reference's loop does).  `tqdm` wraps the loaders in the
    reference; it is importable here."""
    import tempfile

    n_cls, n_train, n_id, n_ood, bs = 6, 192, 40, 37, 16
    torch.Tensor.cuda = lambda self, *a, **k: self  # shim (2): no GPU here

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

    net = Net4x(m)
    out = {"n_cls": np.array(n_cls), "n_train": np.array(n_train), "n_id": np.array(n_id),
           "n_ood": np.array(n_ood), "batch": np.array(bs)}
    for normalize in (False, True):
        with tempfile.TemporaryDirectory() as td:
            args = types.SimpleNamespace(n_cls=n_cls, feat_dim=geo.proj_dim, gpu="cpu", model="CLIP",
                                         normalize=normalize, template_dir=td, in_dataset="ImageNet10",
                                         max_count=250, batch_size=bs)
            mean, prec = ref.get_mean_prec(args, net, Loader(n_train, False, 7))
            s_in = ref.get_Mahalanobis_score(args, net, Loader(n_id, False, 1), mean, prec, in_dist=True)
            s_out = ref.get_Mahalanobis_score(args, net, Loader(n_ood, True, 2), mean, prec, in_dist=False)
        tag = "norm" if normalize else "raw"
        out[f"mean_{tag}"] = mean.numpy()
        out[f"prec_{tag}"] = prec.numpy()
        out[f"in_{tag}"] = np.asarray(s_in)
        out[f"out_{tag}"] = np.asarray(s_out)
        # the features the scores were computed from, so the scoring kernel can be checked alone
        with torch.no_grad():
            feats = []
            for px, _ in Loader(n_id, False, 1):
                f = net.get_image_features(pixel_values=px).float()
                feats.append(f / f.norm(dim=-1, keepdim=True) if normalize else f)
        out[f"feat_in_{tag}"] = torch.cat(feats).numpy()
    np.savez_compressed(os.path.join(HERE, "maha_tiny.npz"), **out)
    print("maha", {k: (v.shape, v.dtype) for k, v in out.items() if v.ndim})


def capture_measures(ref):
    rng = np.random.Generator(np.random.Philox(key=7))
    cases = {}
    cases["kat_pos"], cases["kat_neg"] = np.array([.9, .8, .7, .6]), np.array([.65, .5, .4, .3, .2])
    cases["gauss_pos"], cases["gauss_neg"] = rng.normal(1.0, 1.0, 500), rng.normal(0.0, 1.0, 700)
    # heavy ties: scores quantised to 12 levels
    cases["ties_pos"] = np.round(rng.normal(0.6, 0.2, 300), 1)
    cases["ties_neg"] = np.round(rng.normal(0.4, 0.2, 200), 1)
    # float32 MCM-like scores: narrow band around 1/K
    cases["narrow_pos"] = (0.1 + 1e-3 * rng.standard_normal(400)).astype(np.float32)
    cases["narrow_neg"] = (0.1 + 1e-3 * rng.standard_normal(600) - 4e-4).astype(np.float32)
    # all-equal scores, and perfectly separated
    cases["equal_pos"], cases["equal_neg"] = np.full(20, 0.5), np.full(30, 0.5)
    cases["sep_pos"], cases["sep_neg"] = np.linspace(2, 3, 50), np.linspace(0, 1, 60)
    out = dict(cases)
    for name in ("kat", "gauss", "ties", "narrow", "equal", "sep"):
        out[f"{name}_measures"] = np.array(
            ref.get_measures(cases[f"{name}_pos"], cases[f"{name}_neg"]), dtype=np.float64)
        print(name, out[f"{name}_measures"])
    np.savez_compressed(os.path.join(HERE, "measures.npz"), **out)


PREPROCESS_CASES = [(375, 500), (500, 333), (64, 48), (224, 300), (1000, 760), (231, 517), (224, 224),
                    (225, 224), (3000, 2000), (299, 299)]


def preprocess_input(h, w):
    """The test re-creates the inputs from this seed; only outputs are stored."""
    return np.random.default_rng(h * 10007 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)


def capture_preprocess(size=224):
    """Resize(224) + CenterCrop(224) exactly as the reference's loader transform applies them
    (utils/train_eval_util.py:27-33) to a PIL image: torchvision's size / crop arithmetic
    (torchvision itself is not installed; its two formulas are restated here) around Pillow's
    own Image.resize(BILINEAR) and Image.crop."""
    import hashlib

    from PIL import Image

    out = {"cases": np.array(PREPROCESS_CASES, dtype=np.int32)}
    for h, w in PREPROCESS_CASES:
        img = Image.fromarray(preprocess_input(h, w))
        short, long_ = (w, h) if w <= h else (h, w)
        if short != size:  # torchvision.transforms.functional.resize with an int size
            new_short, new_long = size, int(size * long_ / short)
            nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
            img = img.resize((nw, nh), Image.BILINEAR)
        nw, nh = img.size
        top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))  # center_crop
        arr = np.asarray(img.crop((left, top, left + size, top + size)))
        assert arr.shape == (size, size, 3)
        out[f"sha256_{h}x{w}"] = np.frombuffer(hashlib.sha256(arr.tobytes()).digest(), dtype=np.uint8)
        out[f"patch_{h}x{w}"] = arr[:24, :24].copy()       # a corner, for diagnostics on mismatch
        out[f"rowsum_{h}x{w}"] = arr.astype(np.int64).sum(axis=(1, 2))
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)


def capture_concept_banks():
    """The reference's four ImageNet concept banks (utils/common.py:16-73), produced by ITS functions from ITS
    data/ directory: ImageNet10 / ImageNet20 in full (what utils/common.py must reproduce without any data
    file), ImageNet100 / ImageNet-1k as sha-256 of the joined names (checked wherever the reference's data/
    directory is available, i.e. in the build container)."""
    import hashlib
    import json

    spec = importlib.util.spec_from_file_location("ref_common", "/root/reference/utils/common.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cwd = os.getcwd()
    os.chdir("/root/reference")  # the reference opens data/... relative to its own root
    try:
        banks = {"ImageNet10": list(mod.obtain_ImageNet10_classes()),
                 "ImageNet20": list(mod.obtain_ImageNet20_classes())}
        for name, fn in (("ImageNet100", mod.obtain_ImageNet100_classes), ("ImageNet", mod.obtain_ImageNet_classes)):
            names = [str(x) for x in fn()]
            banks[name + "_sha256"] = hashlib.sha256("\n".join(names).encode()).hexdigest()
            banks[name + "_n"] = len(names)
            banks[name + "_first3"] = names[:3]
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, "concept_banks.json"), "w") as f:
        json.dump(banks, f, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "banks":
        capture_concept_banks()
        sys.exit(0)
    ref = load_reference_detection_util()
    capture_measures(ref)
    m, geo = capture_clip("tiny", n_img=4, n_txt=6)
    capture_scores(ref, m, geo)
    capture_maha(ref, m, geo)
    capture_clip("B16-2L", n_img=2, n_txt=4, sample_rows=[0, 1, 57, 196])
    capture_clip("ViT-B/16", n_img=2, n_txt=4, sample_rows=[0, 196])
    capture_clip("ViT-L/14", n_img=1, n_txt=2, sample_rows=[0, 256])   # BASELINE config 4's checkpoint
    capture_clip("ViT-B/32", n_img=2, n_txt=2, sample_rows=[0, 49])
    capture_preprocess()
    capture_concept_banks()
