"""Host -> device ingest pipes (mcm_amd/ingest.py) on a real MI355X: what comes out of the
copy-stream pipelines is what the direct calls give, batch for batch, whatever the source memory (pinned / pageable), the
batch sizes, and the growth of the packed buffers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def net():
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = geometry("B16-2L")
    n = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=48, max_prompt_tokens=1024)
    yield n
    n.close()


def _bank(net, K=10):
    from mcm_amd.synth import make_token_ids

    ids, _ = make_token_ids(K, seed=2)
    return net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)


def test_pinned_batch_pipe_equals_direct_scoring(net):
    from mcm_amd.ingest import PinnedBatchPipe

    bank = _bank(net)
    g = torch.Generator().manual_seed(1)
    sizes = [48, 48, 17, 48, 1, 30]
    batches = [torch.randint(0, 256, (b, 224, 224, 3), dtype=torch.uint8, generator=g) for b in sizes]
    want = [net.score_images(b.cuda(), bank).clone() for b in batches]
    for source in ("pinned", "pageable", "numpy"):
        src = [b.pin_memory() if source == "pinned" else (b.numpy() if source == "numpy" else b) for b in batches]
        pipe = PinnedBatchPipe(net, 48)
        got = [net.score_images(px, bank).clone() for px in pipe.stream(src)]
        torch.cuda.synchronize()
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want)), source
        assert pipe.bytes_copied == sum(sizes) * 224 * 224 * 3
    with pytest.raises(ValueError):
        list(PinnedBatchPipe(net, 48).stream([torch.zeros((2, 224, 224, 4), dtype=torch.uint8)]))


def test_packed_image_pipe_equals_per_image_resize_crop(net):
    """Variable-size images: one packed upload per batch + resize/crop over pointers into it == the images uploaded one by
    one through `resize_crop` (itself bit-exact against Pillow, tests/test_gpu_preprocess.py); the packed buffers grow when
    a later batch is larger than anything seen before; 1 and 8 native pack threads."""
    from mcm_amd.ingest import PackedImagePipe

    rng = np.random.default_rng(5)

    def batch(n, lo, hi):
        return [rng.integers(0, 256, size=(int(rng.integers(lo, hi)), int(rng.integers(lo, hi)), 3), dtype=np.uint8)
                for _ in range(n)]

    batches = [batch(7, 224, 300), batch(48, 224, 400), batch(3, 600, 900), batch(20, 224, 260), batch(48, 500, 700)]
    want = [net.resize_crop([torch.from_numpy(a) for a in b]).clone() for b in batches]
    for threads in (1, 8):
        pipe = PackedImagePipe(net, 48, 1 << 20, pack_threads=threads)   # 1 MiB to start with: every batch outgrows it
        got = [x.clone() for x in pipe.stream(batches)]
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a.dtype == torch.uint8 and torch.equal(a, b)
        assert pipe.bytes_copied >= sum(x.size for b in batches for x in b)
    with pytest.raises(ValueError):
        list(PackedImagePipe(net, 4, 1 << 20).stream([batch(5, 224, 230)]))


def test_jpeg_folder_through_decode_processes_equals_serial_decode(net, tmp_path, monkeypatch):
    """The image-folder loader end to end in a process that holds a HIP context: JPEG / PNG files decoded by the worker
    PROCESSES of mcm_amd/decode_pool.py (shared-memory hand-over) -> packed copy -> Resize + CenterCrop on the device ->
    scores, against the same files decoded in this thread; two passes over the same loader (its pool and its pinned slots
    are kept), ragged last batch, one image larger than a shared place (decoded by the parent)."""
    from PIL import Image

    from mcm_amd.folder import ImageFolderU8

    rng = np.random.default_rng(4)
    yy, xx = np.mgrid[0:700, 0:700].astype(np.float32)
    n = 0
    for c in ("a", "b", "c"):
        (tmp_path / c).mkdir()
        for i in range(37):
            h, w = int(rng.integers(226, 520)), int(rng.integers(226, 520))
            im = np.stack([127 + 90 * np.sin(0.03 * (k + 1) * xx[:h, :w] + i) * np.cos(0.02 * yy[:h, :w] + n) for k in range(3)], -1)
            im = np.clip(im + rng.normal(0, 10, im.shape), 0, 255).astype(np.uint8)
            Image.fromarray(im).save(tmp_path / c / f"{i:03d}.{'png' if i % 9 == 0 else 'jpg'}", quality=92)
            n += 1
    Image.fromarray(rng.integers(0, 256, (1100, 1300, 3), dtype=np.uint8)).save(tmp_path / "c" / "zz_big.png")  # 4.3 MB > a 3-MB place
    n += 1
    bank = _bank(net)

    def scores(loader, passes=1):
        out = []
        for _ in range(passes):
            out.append(torch.cat([net.score_images(px, bank).clone() for px, _ in loader]))
        return out

    monkeypatch.setenv("MCM_GPU_JPEG", "0")   # the Pillow routes first: in this thread, then in the worker processes
    serial = scores(ImageFolderU8(str(tmp_path), net, 48, workers=1))[0]
    pooled_loader = ImageFolderU8(str(tmp_path), net, 48, workers=5)
    first, second = scores(pooled_loader, passes=2)
    pooled_loader.close()
    assert serial.numel() == n == 112
    assert torch.equal(first, serial) and torch.equal(second, serial)
    lo, hi = 50, 101   # a rank's shard of the same folder
    shard = ImageFolderU8(str(tmp_path), net, 48, workers=3).shard(lo, hi)
    assert torch.equal(scores(shard)[0], serial[lo:hi])
    shard.close()
    # the default route: entropy decode on host threads, the rest of the JPEG decode on the device (the PNG files and the
    # oversized PNG take Pillow inside the pipe) — the same scores, bit for bit, over two passes and over a shard
    monkeypatch.setenv("MCM_GPU_JPEG", "1")
    dev_loader = ImageFolderU8(str(tmp_path), net, 48, workers=3)
    first, second = scores(dev_loader, passes=2)
    assert torch.equal(first, serial) and torch.equal(second, serial)
    assert torch.equal(scores(dev_loader.shard(lo, hi))[0], serial[lo:hi])
