"""Host-side pieces of the drop-in boundary that need no GPU: concept banks vs the reference's own
(tests/golden/concept_banks.json, written by make_golden.py from /root/reference), the `--templates`
reader, the CLI's flag surface, the folder index order."""
import hashlib
import json
import os
import types

import numpy as np
import pytest


def _banks(golden_dir):
    return json.load(open(os.path.join(golden_dir, "concept_banks.json")))


@pytest.mark.parametrize("ds", ["ImageNet10", "ImageNet20"])
def test_hardcoded_banks_equal_the_reference(golden_dir, ds):
    from utils.common import get_test_labels

    got = get_test_labels(types.SimpleNamespace(in_dataset=ds))
    assert list(got) == _banks(golden_dir)[ds]


@pytest.mark.parametrize("ds", ["ImageNet100", "ImageNet"])
def test_file_banks_equal_the_reference(golden_dir, ds):
    """Needs the reference's data/ directory (class-name files are dataset artefacts, not shipped)."""
    from utils.common import get_test_labels

    data = "/root/reference/data"
    if not os.path.isdir(data):
        pytest.skip("reference data/ directory not present on this box")
    got = [str(x) for x in get_test_labels(types.SimpleNamespace(in_dataset=ds, data_dir=data))]
    want = _banks(golden_dir)
    assert len(got) == want[ds + "_n"] and got[:3] == want[ds + "_first3"]
    assert hashlib.sha256("\n".join(got).encode()).hexdigest() == want[ds + "_sha256"]


def test_missing_bank_files_warn_or_raise(tmp_path):
    from utils.common import get_test_labels

    a = types.SimpleNamespace(in_dataset="ImageNet", data_dir=str(tmp_path))
    with pytest.warns(RuntimeWarning, match="placeholder"):
        assert len(get_test_labels(a)) == 1000
    a.weights = "clip.safetensors"
    with pytest.raises(FileNotFoundError):
        get_test_labels(a)


def test_read_templates(tmp_path):
    from mcm_amd.detection import read_templates

    txt = tmp_path / "t.txt"
    txt.write_text("# comment\na photo of a {c}.\n\nitap of my {}.\n")
    assert read_templates(str(txt)) == ["a photo of a {c}.", "itap of my {}."]
    py = tmp_path / "templates.py"
    py.write_text("tpl = [\n    lambda c: f'a bad photo of a {c}.',\n    lambda c: f\"art of the {c}.\",\n]\n\n"
                  "subset = {0: [\n    lambda c: f'a photo of a {c}.',\n]}\n")
    assert read_templates(str(py)) == ["a bad photo of a {c}.", "art of the {c}."]   # first list only
    bad = tmp_path / "bad.txt"
    bad.write_text("no placeholder here\n")
    with pytest.raises(ValueError):
        read_templates(str(bad))
    ref = "/root/reference/utils/imagenet_templates.py"
    if os.path.exists(ref):  # the 80 OpenAI templates the reference ships as data (BASELINE config 5)
        t = read_templates(ref)
        assert len(t) == 80 and t[0] == "a bad photo of a {c}." and all("{c}" in x for x in t)


def test_cli_flags_and_ckpt_mapping(tmp_path, monkeypatch):
    """Reference flags keep their names/defaults (eval_ood_detection.py:15-45); `--generate False` parses
    as True there too (argparse type=bool), and the hub-id mapping of set_model_clip is applied."""
    monkeypatch.chdir(tmp_path)
    import eval_ood_detection as cli
    from mcm_amd.config import HUB_IDS

    a = cli.process_args([])
    assert (a.in_dataset, a.batch_size, a.T, a.score, a.CLIP_ckpt, a.seed) == ("ImageNet", 512, 1, "MCM", "ViT-B/16", 5)
    assert a.generate is True and a.normalize is False and a.max_count == 250 and a.n_cls == 1000
    assert cli.process_args(["--generate", "False"]).generate is True   # bool("False"): the reference's quirk
    assert cli.process_args(["--generate", ""]).generate is False
    assert HUB_IDS["ViT-B/16"] == "openai/clip-vit-base-patch16"
    t = tmp_path / "t.txt"
    t.write_text("a photo of a {c}.\na drawing of a {c}.\n")
    assert cli.process_args(["--templates", str(t)]).templates == ["a photo of a {c}.", "a drawing of a {c}."]


def test_folder_index_order(tmp_path):
    """ImageFolder semantics: classes sorted by name, files sorted inside a class, non-images skipped."""
    from mcm_amd.folder import FolderIndex

    for c, files in (("n02", ["b.JPEG", "a.jpg", "notes.txt"]), ("n01", ["z.png"]), ("n10", ["sub/k.jpeg"])):
        for f in files:
            p = tmp_path / c / f
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_bytes(b"x")
    idx = FolderIndex(str(tmp_path))
    assert idx.classes == ["n01", "n02", "n10"] and len(idx) == 4
    assert [os.path.relpath(p, tmp_path) for p, _ in idx.samples] == ["n01/z.png", "n02/a.jpg", "n02/b.JPEG",
                                                                      "n10/sub/k.jpeg"]
    assert idx.targets == [0, 1, 1, 2]


def test_device_pattern_loader_is_deterministic_and_shardable():
    torch = pytest.importorskip("torch")
    from mcm_amd.synth import DevicePatternLoader

    dev = torch.device("cpu")
    full = DevicePatternLoader(20, 32, 7, 8, dev, ood=False, tile=1.0)
    a = torch.cat([x for x, _ in full])
    b = torch.cat([x for x, _ in full])
    assert torch.equal(a, b) and a.shape == (20, 3, 32, 32)
    parts = torch.cat([x for x, _ in full.shard(0, 8)] + [x for x, _ in full.shard(8, 20)])
    assert torch.equal(a, parts)                      # shard boundaries on batch boundaries
    # ... and anywhere else: image i is a function of (seed, i) only (noise drawn in aligned 64-image blocks), so the
    # per-rank shards of mcm_amd.dist.shard_range and any other batch size see the same pixels
    big = DevicePatternLoader(150, 16, 7, 64, dev, ood=False)
    whole = torch.cat([x for x, _ in big])
    odd = torch.cat([x for x, _ in big.shard(0, 75)] + [x for x, _ in big.shard(75, 150)])
    rebatched = torch.cat([x for x, _ in DevicePatternLoader(150, 16, 7, 37, dev, ood=False)])
    assert torch.equal(whole, odd) and torch.equal(whole, rebatched)
    ood = torch.cat([x for x, _ in DevicePatternLoader(20, 32, 7, 8, dev, ood=True, tile=1.0)])
    assert not torch.equal(a, ood)
    labs = torch.cat([y for _, y in full])
    assert labs.tolist() == [i % 7 for i in range(20)]


def test_folder_subset_is_the_reference_rule(tmp_path):
    """`--subset`: the first max_count samples of every class in dataset order (reference
    utils/train_eval_util.py:56-64)."""
    from mcm_amd.folder import FolderIndex

    for c, n in (("a", 5), ("b", 2), ("c", 3)):
        for i in range(n):
            p = tmp_path / c / f"{i}.png"
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_bytes(b"x")
    idx = FolderIndex(str(tmp_path))
    sub = idx.first_per_class(3)
    assert len(idx) == 10 and len(sub) == 8 and sub.classes == idx.classes
    assert sub.targets == [0, 0, 0, 1, 1, 2, 2, 2]
    assert [os.path.basename(p) for p, _ in sub.samples] == ["0.png", "1.png", "2.png", "0.png", "1.png", "0.png",
                                                             "1.png", "2.png"]


def test_image_folder_decode_pool_keeps_order(tmp_path):
    """The pooled decode (worker processes; the next batches decoded while batch i is scored) yields the same batches, in
    order, as the serial path."""
    torch = pytest.importorskip("torch")
    from PIL import Image

    from mcm_amd.folder import ImageFolderU8

    rng = np.random.default_rng(0)
    for c in ("x", "y"):
        for i in range(5):
            p = tmp_path / c / f"{i}.png"
            p.parent.mkdir(parents=True, exist_ok=True)
            Image.fromarray(rng.integers(0, 256, (8 + i, 9, 3), dtype=np.uint8)).save(p)

    def walk(workers):  # the host half of the loader (the device half — packed upload + resize/crop — is GPU-tested)
        out = []
        for imgs, labels in ImageFolderU8(str(tmp_path), None, 3, workers=workers).decoded_batches():
            out.append((imgs, labels.tolist()))
        return out

    serial, pooled = walk(1), walk(4)
    assert len(serial) == len(pooled) == 4
    for (a, la), (b, lb) in zip(serial, pooled):
        assert la == lb and len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


def test_decode_worker_processes_overflow_and_errors(tmp_path):
    """DecodePool (mcm_amd/decode_pool.py): worker PROCESSES write the pixels into shared memory — same images as the serial
    decode, in order; an image larger than a shared place is decoded by the parent; a file that does not decode raises
    the decoder's own error in the parent, not a hang; `copy=False` hands out views of the shared slots."""
    pytest.importorskip("torch")
    from PIL import Image

    from mcm_amd.folder import DecodePool, ImageFolderU8, _decode_rgb

    rng = np.random.default_rng(1)
    paths = []
    for i in range(21):
        p = tmp_path / "c" / f"{i:02d}.png"
        p.parent.mkdir(parents=True, exist_ok=True)
        Image.fromarray(rng.integers(0, 256, (5 + i, 7 + (i % 3), 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    want = [_decode_rgb(p) for p in paths]
    pool = DecodePool(3, batch=8, slots=3, stride=300)   # 300-byte places: images 11 .. 20 do not fit
    try:
        got = []
        for b in range(3):
            pool.submit(b, paths[b * 8:(b + 1) * 8])
        for b in range(3):
            imgs = pool.collect(b)
            fits = [a.nbytes <= 300 for a in imgs]
            shared = np.frombuffer(pool.buf, dtype=np.uint8)
            assert all(np.shares_memory(a, shared) == f for a, f in zip(imgs, fits))  # views of the slot | parent-decoded
            got += [a.copy() for a in imgs]
        assert len(got) == 21 and all(np.array_equal(a, b) for a, b in zip(got, want))
        bad = tmp_path / "c" / "zz_bad.png"
        bad.write_bytes(b"not an image")
        pool.submit(0, [paths[0], str(bad)])
        with pytest.raises(Exception, match="cannot identify image file"):
            pool.collect(0)
    finally:
        pool.close()
    assert all(p.poll() is not None for p in pool.procs)   # the workers are gone
    bad.unlink()
    ld = ImageFolderU8(str(tmp_path), None, 4, workers=2)
    it = ld.decoded_batches()
    next(it)
    del it                       # a consumer that stops after one batch: the next pass waits out what was in flight
    views = [imgs for imgs, _ in ld.decoded_batches(copy=False)]
    assert sum(len(v) for v in views) == 21 and np.array_equal(views[-1][-1], want[-1])   # the last batch is still intact
    ld.close()


def test_effective_cpus_follows_the_cgroup_quota(monkeypatch):
    """hostinfo.effective_cpus: min(affinity, cgroup CPU quota) — what the decode pool and the CPU baseline size themselves by."""
    import builtins
    import io

    from mcm_amd import hostinfo

    real_open = builtins.open

    def fake(content):
        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                return io.StringIO(content)
            return real_open(path, *a, **k)
        return _open

    monkeypatch.setattr(hostinfo.os, "sched_getaffinity", lambda _pid: set(range(256)), raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert hostinfo.cpu_quota() == 16.0 and hostinfo.effective_cpus() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert hostinfo.cpu_quota() is None and hostinfo.effective_cpus() == 256
    monkeypatch.setattr(builtins, "open", fake("50000 100000\n"))
    assert hostinfo.effective_cpus() == 1


def test_hash_tokenizer_is_refused_by_a_net_with_real_weights():
    """A reference-shaped caller has only args.ckpt; whether the hash stand-in is acceptable is decided by what the
    net says about its weights, falling back to args.weights for nets that do not say."""
    from mcm_amd import detection
    from mcm_amd.tokenizer import TokenizerUnavailable

    args = types.SimpleNamespace(ckpt="openai/clip-vit-base-patch16")
    real = types.SimpleNamespace(synthetic_weights=False)
    synth = types.SimpleNamespace(synthetic_weights=True)
    with pytest.raises(TokenizerUnavailable):
        detection._tokenizer(args, real)
    with pytest.warns(RuntimeWarning, match="HashTokenizer"):
        assert detection._tokenizer(args, synth) is not None
    args.weights = "clip.safetensors"          # a net that does not say: the old rule
    with pytest.raises(TokenizerUnavailable):
        detection._tokenizer(args, types.SimpleNamespace())


def test_prompt_bank_is_encoded_once_per_net_and_key():
    torch = pytest.importorskip("torch")
    from mcm_amd import detection

    class Net:
        synthetic_weights = True
        calls = 0

        def get_text_features(self, input_ids, attention_mask=None, normalize=False):
            Net.calls += 1
            assert normalize is True   # the fused form is detected from the signature, not by catching TypeError
            return torch.ones(input_ids.shape[0], 4)

    class Plain:   # a net honouring only the HF contract: normalised here
        synthetic_weights = True

        def get_text_features(self, input_ids, attention_mask=None):
            return torch.full((input_ids.shape[0], 4), 2.0)

    args = types.SimpleNamespace(ckpt="x", templates=None)
    net = Net()
    with pytest.warns(RuntimeWarning):
        a = detection.prompt_bank(args, net, ["cat", "dog"])
        b = detection.prompt_bank(args, net, ["cat", "dog"])
        assert a is b and Net.calls == 1
        detection.prompt_bank(args, net, ["cat", "dog", "eel"])     # another bank: re-encoded
        assert Net.calls == 2
        args.templates = ["a {c}", "the {c}"]
        net.reduce_bank = lambda f, K, T: f.view(K, T, -1).mean(1)
        assert detection.prompt_bank(args, net, ["cat", "dog", "eel"]).shape == (3, 4) and Net.calls == 3
        f = detection.prompt_bank(types.SimpleNamespace(ckpt="x", templates=None), Plain(), ["cat"])
    assert torch.allclose(f.norm(dim=1), torch.ones(1))


def test_get_mean_prec_ignores_labels_beyond_n_cls(tmp_path):
    """A batch holding a label >= n_cls used to make the per-batch count arrays ragged; the reference's
    per-class loop simply never visits such labels."""
    torch = pytest.importorskip("torch")
    from mcm_amd import detection

    rng = np.random.default_rng(0)
    feats = torch.from_numpy(rng.standard_normal((12, 6)).astype(np.float32))

    class Net:
        def __init__(self):
            self.i = 0

        def get_image_features(self, pixel_values):
            out = feats[self.i:self.i + pixel_values.shape[0]]
            self.i += pixel_values.shape[0]
            return out

    batches = [(torch.zeros(4, 1), torch.tensor([0, 1, 2, 7])), (torch.zeros(4, 1), torch.tensor([1, 1, 0, 2])),
               (torch.zeros(4, 1), torch.tensor([2, 0, 9, 1]))]
    args = types.SimpleNamespace(n_cls=3, normalize=False, template_dir=None, model="CLIP", in_dataset="x",
                                 max_count=250)
    mean, prec = detection.get_mean_prec(args, Net(), batches)
    assert mean.shape == (3, 6) and prec.shape == (6, 6) and torch.isfinite(mean).all()
