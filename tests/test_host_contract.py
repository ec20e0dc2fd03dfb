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
    ood = torch.cat([x for x, _ in DevicePatternLoader(20, 32, 7, 8, dev, ood=True, tile=1.0)])
    assert not torch.equal(a, ood)
    labs = torch.cat([y for _, y in full])
    assert labs.tolist() == [i % 7 for i in range(20)]
