"""JPEG ingest on the MI355X: host entropy decode (mcm_jpeg_entropy_decode) -> device reconstruction (mcm_jpeg_reconstruct,
csrc/jpeg.hip) must give Pillow's pixels byte for byte (the reference loader decodes with Pillow), and the file pipe on top
(mcm_amd/ingest.py::JpegFilePipe: + Resize + CenterCrop) the crops the Pillow route gives — fallback files included."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from tests.test_jpeg_oracle import CASES, _photo, entropy_decode  # noqa: E402


@pytest.fixture(scope="module")
def net():
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry("B16-2L")
    n = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=48, max_prompt_tokens=1024)
    yield n
    n.close()


def _pil(path):
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def test_device_reconstruction_equals_pillow_on_a_mixed_batch(net, tmp_path):
    paths = []
    for k, (h, w, kw) in enumerate(CASES):
        p = str(tmp_path / f"{k}.jpg")
        Image.fromarray(_photo(h, w, k)).save(p, **kw)
        paths.append(p)
    for k, (h, w) in enumerate([(100, 130), (64, 64), (97, 31)]):
        p = str(tmp_path / f"g{k}.jpg")
        Image.fromarray(_photo(h, w, k)[:, :, 0]).save(p, quality=88)
        paths.append(p)
    p = str(tmp_path / "cmyk.jpg")
    Image.fromarray(_photo(90, 120, 5)).convert("CMYK").save(p, quality=90)   # not taken: skipped by the device call
    paths.append(p)
    meta, quant, buf = entropy_decode(paths, threads=4)
    n = len(paths)
    want = [_pil(p) for p in paths]
    assert [m.status for m in meta].count(0) == n - 1 and meta[n - 1].status == 1
    offs, o = [], 0
    for a in want:
        offs.append(o)
        o += (a.size + 15) // 16 * 16
    rgb = torch.full((o,), 7, dtype=torch.uint8, device="cuda")
    coef = torch.from_numpy(buf).cuda()
    for _ in range(2):   # twice: the second call re-uses the handle's workspace
        net.jpeg_reconstruct(coef, meta, quant, n, rgb, offs)
    got = rgb.cpu().numpy()
    for i, a in enumerate(want):
        g = got[offs[i]: offs[i] + a.size].reshape(a.shape)
        if meta[i].status == 0:
            np.testing.assert_array_equal(g, a, err_msg=os.path.basename(paths[i]))
        else:
            assert (g == 7).all()   # untouched


def test_file_pipe_equals_the_pillow_route_with_fallbacks_and_errors(net, tmp_path):
    from mcm_amd.ingest import JpegFilePipe
    from oracle import oracle as orc

    rng = np.random.default_rng(3)
    paths = []
    for k in range(70):
        h, w = int(rng.integers(226, 520)), int(rng.integers(226, 520))
        p = str(tmp_path / f"{k:03d}.jpg")
        kw = dict(quality=int(rng.integers(40, 98)), subsampling=int(rng.integers(0, 3)))
        kw["optimize"] = bool(k % 2) and kw["quality"] <= 85   # (Pillow's encoder buffer is too small for optimize at high quality)
        if k % 11 == 0:
            kw = dict(quality=85, progressive=True)   # progressive: taken by the entropy decoder as well
        if k % 17 == 3:
            kw = dict(quality=85, cmyk=True)          # CMYK: Pillow fallback inside the pipe
        if k % 13 == 5:
            p = p[:-4] + ".png"                 # not a JPEG at all: Pillow fallback
            kw = {}
        im = Image.fromarray(_photo(h, w, 100 + k))
        if kw.pop("cmyk", False):
            im = im.convert("CMYK")
        im.save(p, **kw)
        paths.append(p)
    want = np.stack([orc.resize_crop_u8(_pil(p), 224) for p in paths])
    pipe = JpegFilePipe(net, 32, threads=3)
    for rep in range(2):
        batches = [paths[0:32], paths[32:49], paths[49:70]]
        got = torch.cat([b.clone() for b in pipe.stream(batches)]).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    assert pipe.fallback_images == 2 * sum(1 for k in range(70) if k % 13 == 5 or k % 17 == 3)
    bad = str(tmp_path / "bad.jpg")
    open(bad, "wb").write(open(paths[1], "rb").read()[:500])
    with pytest.raises(Exception):
        for _ in pipe.stream([paths[:3] + [bad]]):
            pass
    got = torch.cat([b.clone() for b in pipe.stream([paths[:5]])]).cpu().numpy()   # the pipe is usable after an error
    np.testing.assert_array_equal(got, want[:5])


def test_two_slot_pipe_never_refills_a_slot_the_consumer_still_reads(net, tmp_path):
    """with depth 2 the queue alone let the producer re-enter slot 0 while batch 0 was still being uploaded /
    reconstructed.  Slots are now handed back explicitly after the consumer's last use: 9 small batches through a 2-slot pipe
    with a slow consumer give Pillow's crops, and a pass abandoned at its first batch leaves the pipe usable."""
    import time

    from mcm_amd.ingest import JpegFilePipe
    from oracle import oracle as orc

    paths = []
    for k in range(36):
        p = str(tmp_path / f"s{k:02d}.jpg")
        Image.fromarray(_photo(230 + 7 * (k % 5), 240 + 11 * (k % 7), 300 + k)).save(p, quality=60 + k)
        paths.append(p)
    want = np.stack([orc.resize_crop_u8(_pil(p), 224) for p in paths])
    pipe = JpegFilePipe(net, 4, depth=2, threads=2)
    assert len(pipe.slots) == 2
    got = []
    for b in pipe.stream([paths[i:i + 4] for i in range(0, 36, 4)]):
        time.sleep(0.02)              # the producer is ahead of the consumer the whole pass
        got.append(b.clone())
    np.testing.assert_array_equal(torch.cat(got).cpu().numpy(), want)
    g = pipe.stream([paths[i:i + 4] for i in range(0, 36, 4)])
    next(g)
    g.close()                         # abandoned mid-pass: producer joined, the slot in flight recorded
    got = torch.cat([b.clone() for b in pipe.stream([paths[i:i + 4] for i in range(0, 36, 4)])]).cpu().numpy()
    np.testing.assert_array_equal(got, want)


def test_reconstruction_with_an_odd_max_batch(tmp_path):
    """The staging ring of mcm_jpeg_reconstruct keeps the quantisation tables behind max_batch records; the kernel reads them
    in 16-byte rows, so their offset must be aligned whatever max_batch is (7 records end on an odd multiple of 8 bytes)."""
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry("B16-2L")
    net = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=7, max_prompt_tokens=1024)
    try:
        paths = []
        for k, (h, w, kw) in enumerate(CASES[:5]):
            p = str(tmp_path / f"{k}.jpg")
            Image.fromarray(_photo(h, w, 40 + k)).save(p, **kw)
            paths.append(p)
        meta, quant, buf = entropy_decode(paths)
        want = [_pil(p) for p in paths]
        offs, o = [], 0
        for a in want:
            offs.append(o)
            o += (a.size + 15) // 16 * 16
        rgb = torch.zeros(o, dtype=torch.uint8, device="cuda")
        net.jpeg_reconstruct(torch.from_numpy(buf).cuda(), meta, quant, len(paths), rgb, offs)
        got = rgb.cpu().numpy()
        for i, a in enumerate(want):
            np.testing.assert_array_equal(got[offs[i]: offs[i] + a.size].reshape(a.shape), a)
    finally:
        net.close()
