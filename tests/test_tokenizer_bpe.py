"""The library's C++ CLIP BPE (mcm_amd/csrc/tokenizer.cpp) vs HF transformers'
CLIPTokenizer — the tokenizer the reference calls (utils/detection_util.py:216,228) — built from the
SAME synthetic vocabulary (the real vocab.json / merges.txt exist in neither container); the Unicode side
(NFC, classes, lowercase) over every block.  Host code: runs without a GPU."""
import collections
import json
import os

import numpy as np
import pytest

transformers = pytest.importorskip("transformers")

CORPUS = """a photo of a tench goldfish great white shark tiger shark hammerhead electric ray stingray
cock hen ostrich brambling goldfinch house finch junco indigo bunting robin bulbul jay magpie
chickadee water ouzel kite bald eagle vulture great grey owl fire salamander common newt eft
spotted salamander axolotl bullfrog tree frog tailed frog loggerhead leatherback turtle mud turtle
terrapin box turtle banded gecko common iguana american chameleon whiptail agama frilled lizard
alligator lizard gila monster green lizard african chameleon komodo dragon african crocodile
a blurry photo of the a close-up photo of a a drawing of a a bright photo of a cropped small large
golden retriever labrador german shepherd poodle siamese cat persian cat tabby egyptian cat
the dog's photo isn't it's they're we've i'm you'll he'd 2 dogs 12 cats 365 days 1,000 prompts"""


def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def train_bpe(n_merges):
    """A few hundred merges learned from CORPUS: a vocabulary with realistic, overlapping merges."""
    b2u = bytes_to_unicode()
    words = collections.Counter()
    for w in CORPUS.lower().split():
        sym = [b2u[b] for b in w.encode()]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for sym, c in words.items():
            for a, b in zip(sym, sym[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = collections.Counter()
        for sym, c in words.items():
            out, i = [], 0
            while i < len(sym):
                if i + 1 < len(sym) and (sym[i], sym[i + 1]) == best:
                    out.append(sym[i] + sym[i + 1])
                    i += 2
                else:
                    out.append(sym[i])
                    i += 1
            new[tuple(out)] += c
        words = new
    base = list(b2u.values())
    vocab = base + [v + "</w>" for v in base] + [a + b for a, b in merges]
    vocab += ["<|startoftext|>", "<|endoftext|>"]
    return {t: i for i, t in enumerate(vocab)}, merges


@pytest.fixture(scope="module")
def toks(tmp_path_factory):
    from mcm_amd.tokenizer import NativeBPETokenizer, load_tokenizer

    vocab, merges = train_bpe(400)
    d = tmp_path_factory.mktemp("tok")
    json.dump(vocab, open(d / "vocab.json", "w"), ensure_ascii=True)
    open(d / "merges.txt", "w", encoding="utf-8").write(
        "#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    hf = transformers.CLIPTokenizer(vocab=vocab, merges=merges)
    native = load_tokenizer(str(d))
    assert isinstance(native, NativeBPETokenizer) and len(native) == len(vocab)
    return hf, native, vocab


PROMPTS = [
    "a photo of a tench", "a photo of a great white shark", "a photo of a golden retriever",
    "A Photo of a  Golden   Retriever", "  leading and trailing spaces  ", "the dog's photo, isn't it?",
    "they're here; we've won!! i'm sure you'll see he'd go", "365 days, 1,000 prompts & 12 cats (2 dogs)",
    "tab\tseparated\nand newline", "UPPER lower MiXeD", "hyphen-ated close-up e-mail", "x", "",
    "unseen wordzzz qqq", "a photo of a crème brûlée", "naïve café über straße", "ΑΒΓ αβγ Привет мир",
    "quotes “curly” and ‘single’ — dash", "emoji 🙂 and symbols ©®™ ±≠", "a photo of a <|endoftext|> token",
    "'sup 'tis 'twas", "don't can't won't", "a1b2c3", "10% off $5.00 #tag @user", "end.",
]


def test_matches_hf_clip_tokenizer(toks):
    hf, native, vocab = toks
    want = hf(PROMPTS, padding=True, return_tensors="np")
    got = native(PROMPTS, padding=True, return_tensors="np")
    assert got["input_ids"].shape == want["input_ids"].shape
    for i, p in enumerate(PROMPTS):
        np.testing.assert_array_equal(got["input_ids"][i], want["input_ids"][i], err_msg=repr(p))
        np.testing.assert_array_equal(got["attention_mask"][i], want["attention_mask"][i], err_msg=repr(p))
    eos = vocab["<|endoftext|>"]
    assert (got["input_ids"][:, 0] == vocab["<|startoftext|>"]).all()
    assert (got["input_ids"][got["attention_mask"] == 0] == eos).all()   # pad token = <|endoftext|>


def test_random_ascii_strings_match(toks):
    hf, native, _ = toks
    rng = np.random.default_rng(0)
    alphabet = list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789   ''.,;:!?-_/()&%$#@\"")
    texts = ["".join(rng.choice(alphabet, size=int(rng.integers(1, 60)))) for _ in range(300)]
    want = hf(texts, padding=True, return_tensors="np")
    got = native(texts, padding=True, return_tensors="np")
    np.testing.assert_array_equal(got["input_ids"], want["input_ids"])
    np.testing.assert_array_equal(got["attention_mask"], want["attention_mask"])


def _assert_same(hf, native, texts, capacity=77):
    for i in range(0, len(texts), 4000):
        chunk = texts[i:i + 4000]
        want = hf(chunk, padding=True, return_tensors="np")
        got = native(chunk, padding=True, return_tensors="np", capacity=capacity)
        if want["input_ids"].shape == got["input_ids"].shape and (want["input_ids"] == got["input_ids"]).all() \
                and (want["attention_mask"] == got["attention_mask"]).all():
            continue
        for j, t in enumerate(chunk):  # name the first offender
            a = want["input_ids"][j][want["attention_mask"][j] == 1].tolist()
            b = got["input_ids"][j][got["attention_mask"][j] == 1].tolist()
            assert a == b, [hex(ord(c)) for c in t]


def test_unicode_classes_every_block(toks):
    """\\p{L} / \\p{N} / \\s / lowercase / single-code-point NFC of the generated tables (mcm_amd/csrc/unicode_tables.inc,
    tools/gen_unicode_tables.py) against HF's tokenizer: every code point below U+3000, every 3rd up to U+323B0 (the end of
    the assigned ideographs) and a seeded sample of the rest — each as a letter neighbour, a digit neighbour, word-initial
    and doubled.  (The full sweep of all 1 112 064 code points: 0 mismatches, 3 minutes; this is the CPU suite's share.)"""
    hf, native, _ = toks
    rng = np.random.default_rng(5)
    cps = list(range(0x20, 0x3000)) + list(range(0x3000, 0x323B0, 3)) + \
        [int(c) for c in rng.integers(0x323B0, 0x110000, size=5000)]
    cps = [c for c in cps if not 0xD800 <= c <= 0xDFFF]
    _assert_same(hf, native, [f"a{chr(c)}b 1{chr(c)}2 {chr(c)}x {chr(c)}{chr(c)}" for c in cps])


def test_nfc_random_sequences(toks):
    """Canonical ordering and composition (NFC comes first in HF's normaliser): random strings of ASCII, combining marks,
    decomposable characters, Hangul jamo and arbitrary assigned code points."""
    import unicodedata

    hf, native, _ = toks
    rng = np.random.default_rng(3)
    assigned = [c for c in range(0x20, 0x30000) if unicodedata.category(chr(c)) not in ("Cn", "Co", "Cs")]
    marks = [c for c in assigned if unicodedata.combining(chr(c))]
    decomposable = [c for c in assigned if unicodedata.normalize("NFD", chr(c)) != chr(c)]
    ascii_pool = [ord(c) for c in "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789   '.,-"]
    assigned, marks, decomposable, ascii_pool, jamo = (np.asarray(list(p)) for p in (assigned, marks, decomposable, ascii_pool,
                                                                                       range(0x1100, 0x1200)))
    texts = []
    for _ in range(6000):
        s = []
        for _ in range(int(rng.integers(1, 24))):
            r = rng.random()
            pool = ascii_pool if r < 0.35 else marks if r < 0.55 else decomposable if r < 0.75 else \
                jamo if r < 0.85 else assigned
            s.append(chr(int(pool[rng.integers(len(pool))])))
        texts.append("".join(s))
    texts += ["a\u0301\u0323", "\u0301\u0323", "A\u030a", "\u212b", "\u1100\u1161\u11a8", "\uac00\u11a8", "\u0130stanbul",
              "\u03a3\u0391\u03a3", "e\u0302\u0301", "\u0915\u093c", "\u0958", "\ufb01 \ufb03", "\u2126 \u00b5 \u03bc"]
    _assert_same(hf, native, texts, capacity=256)


def test_reference_call_contract_and_errors(toks, tmp_path):
    import torch

    from mcm_amd.tokenizer import NativeBPETokenizer

    _, native, _ = toks
    out = native([f"a photo of a {c}" for c in ["tench", "goldfish", "great white shark"]], padding=True,
                 return_tensors="pt")   # the reference's call, utils/detection_util.py:228
    assert out["input_ids"].dtype == torch.int64 and out["input_ids"].shape == out["attention_mask"].shape
    assert out["attention_mask"].sum(1).tolist() == sorted(out["attention_mask"].sum(1).tolist())
    with pytest.raises(RuntimeError, match="capacity"):
        native(["word " * 50], capacity=16)
    with pytest.raises(RuntimeError):
        NativeBPETokenizer(str(tmp_path / "missing.json"), str(tmp_path / "missing.txt"))
    (tmp_path / "vocab.json").write_text('{"a": 0}')
    (tmp_path / "merges.txt").write_text("#version: 0.2\n")
    with pytest.raises(RuntimeError, match="startoftext"):
        NativeBPETokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))
