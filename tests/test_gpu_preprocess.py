"""mcm_resize_crop_u8 vs Pillow's outputs (tests/golden/preprocess.npz) and the C
oracle: integer work, so every comparison is bit-exact."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def preprocess_input(h, w):
    return np.random.default_rng(h * 10007 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.fixture(scope="module")
def net():
    from mcm_amd.config import TEST_GEOMETRIES
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.weights import synth_state_dict

    geo = TEST_GEOMETRIES["B16-2L"]  # 224 x 224 input, two layers
    n = NativeCLIP(geo, synth_state_dict(geo, seed=0), precision="bf16", max_batch=32,
                   max_prompt_tokens=64 * 16)
    yield n
    n.close()


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_matches_pillow_fixtures_and_oracle(net, golden_dir):
    from oracle import oracle as orc

    g = np.load(os.path.join(golden_dir, "preprocess.npz"))
    cases = [tuple(int(v) for v in c) for c in g["cases"]]
    imgs = [preprocess_input(h, w) for h, w in cases]
    out = net.resize_crop([_dev(i) for i in imgs]).cpu().numpy()   # one mixed-size batch
    assert out.shape == (len(cases), 224, 224, 3) and out.dtype == np.uint8
    for (h, w), img, got in zip(cases, imgs, out):
        np.testing.assert_array_equal(got, orc.resize_crop_u8(img, 224), err_msg=f"{h}x{w} vs oracle")
        digest = np.frombuffer(hashlib.sha256(got.tobytes()).digest(), dtype=np.uint8)
        np.testing.assert_array_equal(digest, g[f"sha256_{h}x{w}"], err_msg=f"{h}x{w} vs Pillow")


def test_random_sizes_match_oracle(net):
    from oracle import oracle as orc

    rng = np.random.default_rng(11)
    sizes = [(int(rng.integers(20, 900)), int(rng.integers(20, 900))) for _ in range(24)]
    sizes += [(224, 224), (224, 301), (299, 224), (1, 1), (2, 700), (448, 448)]
    # every branch of the kernel's form choice (preprocess.hip): 8-tap tables with 8 / 4 / 2 rows per LDS pass, 16-tap tables
    # with 1 row per pass, a window that fits no pass (fused form), more than 16 taps (fused form), upscaling
    sizes += [(375, 500), (600, 800), (768, 1024), (1200, 1600), (1600, 1200), (2000, 1500), (1800, 4000), (120, 160), (230, 229)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    out = np.concatenate([net.resize_crop([_dev(i) for i in imgs[k:k + 20]]).cpu().numpy() for k in range(0, len(imgs), 20)])
    for (h, w), img, got in zip(sizes, imgs, out):
        np.testing.assert_array_equal(got, orc.resize_crop_u8(img, 224), err_msg=f"{h}x{w}")


def test_packed_images_at_unaligned_offsets_match_oracle(net):
    """The LDS form reads the source window in 16-byte chunks aligned to the ADDRESS: images packed back to back at
    byte offsets of every residue mod 16, the first one at the start of the allocation + 1 and the last one ending at
    its last byte, must come out bit-equal (chunks that stick out of an image are read byte by byte)."""
    import torch

    from oracle import oracle as orc

    rng = np.random.default_rng(23)
    sizes = [(int(rng.integers(225, 520)), int(rng.integers(225, 520))) for _ in range(17)] + [(224, 224), (40, 31)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    offsets, cur = [], 1
    for i, im in enumerate(imgs):
        offsets.append(cur)
        cur += im.size + (i % 5)           # gaps of 0..4 bytes: every alignment shows up
    host = np.zeros(cur - ((len(imgs) - 1) % 5), dtype=np.uint8)    # the last image ends the buffer
    for o, im in zip(offsets, imgs):
        host[o:o + im.size] = im.reshape(-1)
    assert offsets[-1] + imgs[-1].size == host.size and len({o % 16 for o in offsets}) > 8
    out = net.resize_crop_packed(torch.from_numpy(host).cuda(), offsets, [h for h, _ in sizes],
                                 [w for _, w in sizes]).cpu().numpy()
    for (h, w), img, got in zip(sizes, imgs, out):
        np.testing.assert_array_equal(got, orc.resize_crop_u8(img, 224), err_msg=f"{h}x{w}")


def test_feeds_the_scoring_path(net):
    """resize/crop on the device → uint8 scoring route == the same route fed with the oracle's crops."""
    import torch

    from mcm_amd.synth import make_token_ids
    from oracle import oracle as orc

    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (int(rng.integers(230, 600)), int(rng.integers(230, 600)), 3), dtype=np.uint8)
            for _ in range(6)]
    ids, mask = make_token_ids(10, seed=1)
    txt = net.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                                normalize=True)
    dev = net.score_images(net.resize_crop([_dev(i) for i in imgs]), txt, 1.0, "MCM")
    ref = net.score_images(_dev(np.stack([orc.resize_crop_u8(i, 224) for i in imgs])), txt, 1.0, "MCM")
    assert torch.equal(dev, ref) and torch.isfinite(dev).all()


def test_errors(net):
    import torch

    small = torch.zeros((50, 60, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        net.resize_crop([small] * 33)                       # above max_batch
    with pytest.raises(ValueError):
        net.resize_crop([torch.zeros((50, 60), dtype=torch.uint8, device="cuda")])
    huge = torch.zeros((7200, 7100, 3), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="31"):
        net.resize_crop([huge])                             # 2*ceil(31.7)+1 taps > the kernel's 64
