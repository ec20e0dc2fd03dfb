"""JPEG ingest on the CPU: the product's host half (mcm_jpeg_entropy_decode, csrc/jpeg_entropy.cpp: markers + Huffman ->
quantised DCT coefficients) followed by the oracle's restatement of libjpeg's reconstruction (oracle/jpeg_ref.c: islow IDCT,
fancy upsampling, YCbCr -> RGB) must give, byte for byte, what Pillow's Image.open(path).convert("RGB") gives — the
reference loader's decoder (torchvision ImageFolder).  This pins BOTH: the entropy decoder that ships, and the checker the
GPU tests compare jpeg.hip against.  Integer work: every comparison is exact."""
import ctypes
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402


def _photo(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    f = rng.uniform(0.01, 0.2, 6)
    im = np.stack([127 + 100 * np.sin(f[2 * c] * xx + seed) * np.cos(f[2 * c + 1] * yy) for c in range(3)], -1)
    return np.clip(im + rng.normal(0, 25, im.shape), 0, 255).astype(np.uint8)


def entropy_decode(paths, threads=2):
    from mcm_amd.config import JpegImage
    from mcm_amd.engine import load_library

    lib = load_library()
    n = len(paths)
    arr = (ctypes.c_char_p * n)(*[os.fsencode(p) for p in paths])
    meta = (JpegImage * n)()
    quant = np.zeros((n, 3, 64), dtype=np.uint16)
    used = ctypes.c_int64(0)
    rc = lib.mcm_jpeg_entropy_decode(arr, n, None, 0, meta, quant.ctypes.data, threads, ctypes.byref(used))
    assert rc in (0, -7), rc   # MCM_ERANGE: the size query
    buf = np.zeros(max(16, used.value), dtype=np.uint8)
    rc = lib.mcm_jpeg_entropy_decode(arr, n, buf.ctypes.data, buf.size, meta, quant.ctypes.data, threads, ctypes.byref(used))
    assert rc == 0
    return meta, quant, buf


def reconstruct(meta_i, quant_i, buf):
    from oracle import oracle as orc

    m = meta_i
    coef = [np.frombuffer(buf, dtype=np.int16, count=m.hb[c] * m.wb[c] * 64, offset=m.coef_off[c]).reshape(m.hb[c], m.wb[c], 64)
            for c in range(m.ncomp)]
    return orc.jpeg_reconstruct(coef, quant_i[: m.ncomp], m.width, m.height, list(m.hs)[: m.ncomp], list(m.vs)[: m.ncomp],
                                list(m.wb)[: m.ncomp], list(m.hb)[: m.ncomp])


CASES = [  # (h, w, save kwargs)
    (375, 500, dict(quality=90)), (500, 375, dict(quality=75)), (333, 499, dict(quality=95, optimize=True)),
    (224, 224, dict(quality=50)), (17, 23, dict(quality=90)), (8, 8, dict(quality=90)), (1, 1, dict(quality=90)),
    (241, 319, dict(quality=90, subsampling=0)), (241, 319, dict(quality=90, subsampling=1)), (241, 319, dict(quality=90, subsampling=2)),
    (480, 641, dict(quality=100)), (600, 800, dict(quality=20)), (255, 257, dict(quality=85, optimize=True, subsampling=1)),
    (300, 301, dict(quality=90, restart_marker_blocks=7)), (300, 301, dict(quality=90, restart_marker_rows=1, subsampling=0)),
    # progressive (SOF2): spectral selection + successive approximation, DC / AC first and refinement scans
    (375, 500, dict(quality=90, progressive=True)), (241, 319, dict(quality=75, progressive=True, subsampling=0)),
    (241, 319, dict(quality=60, progressive=True, subsampling=1)), (97, 131, dict(quality=95, progressive=True, subsampling=2)),
    (17, 23, dict(quality=90, progressive=True)), (300, 301, dict(quality=85, progressive=True, restart_marker_blocks=5)),
    (600, 800, dict(quality=30, progressive=True)),
    # chroma planes at most two samples wide: libjpeg replicates instead of filtering (jdsample.c jinit_upsampler)
    (279, 2, dict(quality=21, subsampling=2)), (196, 4, dict(quality=25, subsampling=1)), (390, 1, dict(quality=15, subsampling=2)),
    (21, 4, dict(quality=79, subsampling=1)), (40, 3, dict(quality=90, subsampling=2, progressive=True)), (3, 5, dict(quality=90, subsampling=2)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_entropy_decode_plus_reference_reconstruction_equals_pillow(tmp_path, case):
    h, w, kw = CASES[case]
    p = str(tmp_path / "a.jpg")
    Image.fromarray(_photo(h, w, case)).save(p, **kw)
    meta, quant, buf = entropy_decode([p])
    assert meta[0].status == 0 and (meta[0].height, meta[0].width) == (h, w)
    with Image.open(p) as im:
        want = np.asarray(im.convert("RGB"))
    got = reconstruct(meta[0], quant[0], buf)
    np.testing.assert_array_equal(got, want)


def test_grayscale_batch_and_files_the_path_does_not_take(tmp_path):
    paths, want = [], []
    for i, (h, w) in enumerate([(100, 130), (64, 64), (97, 31)]):
        p = str(tmp_path / f"g{i}.jpg")
        Image.fromarray(_photo(h, w, i)[:, :, 0]).save(p, quality=88)
        paths.append(p)
    p = str(tmp_path / "prog_gray.jpg")
    Image.fromarray(_photo(90, 120, 5)[:, :, 1]).save(p, quality=90, progressive=True)
    paths.append(p)
    p = str(tmp_path / "cmyk.jpg")
    Image.fromarray(_photo(90, 120, 6)).convert("CMYK").save(p, quality=90)
    paths.append(p)
    p = str(tmp_path / "png_named.jpg")
    Image.fromarray(_photo(20, 20, 7)).save(p, format="PNG")
    paths.append(p)
    p = str(tmp_path / "truncated.jpg")
    q = str(tmp_path / "whole.jpg")
    Image.fromarray(_photo(200, 200, 8)).save(q, quality=90)
    open(p, "wb").write(open(q, "rb").read()[:3000])
    paths.append(p)
    paths.append(str(tmp_path / "missing.jpg"))
    paths.append(q)
    meta, quant, buf = entropy_decode(paths, threads=3)
    # (the truncated file is NOT taken: its scan does not end in EOI — Pillow, the fallback, raises for it as the reference would)
    assert [m.status for m in meta] == [0, 0, 0, 0, 1, 2, 2, 2, 0], [m.status for m in meta]
    for i in (0, 1, 2, 3, 8):
        with Image.open(paths[i]) as im:
            np.testing.assert_array_equal(reconstruct(meta[i], quant[i], buf), np.asarray(im.convert("RGB")))


def test_damaged_files_never_crash_the_entropy_decoder(tmp_path):
    """The parser reads untrusted bytes: 300 damaged variants of valid files (flipped bytes, truncations, garbage in the
    headers, lengths that lie) must come back as status 0 / 1 / 2 — and whatever is reported as taken must reconstruct
    without touching memory outside its planes."""
    rng = np.random.default_rng(17)
    good = []
    for k, kw in enumerate([dict(quality=90), dict(quality=60, subsampling=0, optimize=True), dict(quality=85, subsampling=1),
                            dict(quality=90, restart_marker_blocks=5), dict(quality=88, progressive=True)]):
        p = str(tmp_path / f"good{k}.jpg")
        Image.fromarray(_photo(70 + 9 * k, 90 + 7 * k, k)).save(p, **kw)
        good.append(open(p, "rb").read())
    paths = []
    for i in range(300):
        b = bytearray(good[i % len(good)])
        mode = i % 5
        if mode == 0:                                   # a few flipped bytes anywhere
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
        elif mode == 1:                                 # damage confined to the headers
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(2, min(len(b), 700)))] = int(rng.integers(0, 256))
        elif mode == 2:                                 # truncation
            b = b[: int(rng.integers(2, len(b)))]
        elif mode == 3:                                 # 0xFF bytes sprinkled into the scan
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(len(b) // 2, len(b)))] = 0xFF
        else:                                           # garbage tail / doubled file
            b = b + bytes(rng.integers(0, 256, 64, dtype=np.uint8)) + b[: len(b) // 3]
        p = str(tmp_path / f"bad{i:03d}.jpg")
        open(p, "wb").write(bytes(b))
        paths.append(p)
    meta, quant, buf = entropy_decode(paths, threads=4)
    status = [m.status for m in meta]
    assert set(status) <= {0, 1, 2} and status.count(0) > 20 and status.count(2) > 20, (status.count(0), status.count(1), status.count(2))
    # What is still reported as taken decoded without any anomaly (every restart marker and the EOI exactly where a clean
    # scan has them); where Pillow decodes the damaged file without complaint as well, the pixels must agree
    agree = differ = 0
    for i, m in enumerate(meta):
        if m.status == 0:
            out = reconstruct(m, quant[i], buf)
            assert out.shape == (m.height, m.width, 3)
            try:
                import warnings

                with warnings.catch_warnings():
                    warnings.simplefilter("error")
                    with Image.open(paths[i]) as im:
                        want = np.asarray(im.convert("RGB"))
            except Exception:
                continue
            if want.shape == out.shape and np.array_equal(want, out):
                agree += 1
            else:
                differ += 1
    # (a damaged scan that is still syntactically clean and keeps its coefficients in a plausible range can differ in the
    # damaged blocks: libjpeg-turbo's SIMD code works in 16 bits there, the restatement exactly — rare, and not a file any
    # two libjpeg builds agree on either)
    assert agree > 10 and differ <= 3, (agree, differ)


def test_dc_huffman_symbol_above_15_goes_to_the_fallback(tmp_path):
    """libjpeg (jpeg_make_d_derived_tbl) refuses a DC Huffman table with a symbol above 15 and Pillow raises
    'broken data stream'; the native entropy decoder used to take such a file when the bad symbol was never decoded.  It is
    now handed to the fallback (status 2), which raises what the reference's loader raises."""
    p = str(tmp_path / "ok.jpg")
    Image.fromarray(_photo(64, 80, 5)).save(p, quality=85)
    data = bytearray(open(p, "rb").read())
    i = 2
    patched = False
    while i + 4 <= len(data) and data[i] == 0xFF:
        mk, ln = data[i + 1], (data[i + 2] << 8) | data[i + 3]
        if mk == 0xC4:  # DHT: walk its tables, overwrite the LAST symbol of the first DC table
            o = i + 4
            end = i + 2 + ln
            while o < end:
                tc, cnt = data[o] >> 4, sum(data[o + 1:o + 17])
                if tc == 0 and not patched:
                    data[o + 17 + cnt - 1] = 0x1F
                    patched = True
                o += 17 + cnt
        if mk == 0xDA:
            break
        i += 2 + ln
    assert patched
    bad = str(tmp_path / "bad_dc.jpg")
    open(bad, "wb").write(bytes(data))
    meta, _, _ = entropy_decode([p, bad])
    assert meta[0].status == 0 and meta[1].status == 2
