"""INTEGRATION.md is executable: the code blocks a reference maintainer would paste are extracted from the
document and exec'd as written — the `ctypes` stub over an HF `CLIPModel`-shaped object, the fused loop, the
raw-image ingest, the device metrics and the split-activation re-scorer — and the reference's loop
(utils/detection_util.py:219-249, restated here) driven through the stub is held to the reference's own outputs
(tests/golden/scores_tiny.npz = its get_ood_scores_clip run in the build container).  Round 5's stub passed a
stale ABI version to mcm_create and nothing ran it."""
import os
import re
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from mcm_amd.config import SCORE_KINDS, geometry  # noqa: E402
from mcm_amd.synth import SyntheticImageSet, SyntheticLoader, make_token_ids  # noqa: E402
from mcm_amd.weights import synth_state_dict  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def doc_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    return re.findall(r"```python\n(.*?)```", text, re.S)


def test_doc_has_the_blocks_and_takes_the_abi_version_from_the_library():
    """CPU: the stub asks the library for its ABI version (never a constant), and its struct mirrors mcm_config."""
    import ctypes

    from mcm_amd.config import ABI_VERSION, CConfig
    from mcm_amd.engine import LIB_PATH

    blocks = doc_blocks()
    assert len(blocks) == 5
    stub = blocks[0]
    assert "_L.mcm_abi_version()" in stub and not re.search(r"McmConfig\(\s*\d", stub)
    os.environ["MCM_LIB"] = LIB_PATH
    ns = {}
    exec(compile(stub, "INTEGRATION.md[stub]", "exec"), ns)   # defines McmConfig / NativeNet; dlopen needs no GPU
    assert [f[0] for f in ns["McmConfig"]._fields_] == [f[0] for f in CConfig._fields_]
    assert ctypes.sizeof(ns["McmConfig"]) == ctypes.sizeof(CConfig)
    assert ns["_L"].mcm_abi_version() == ABI_VERSION


def _hf_model(geo, sd):
    from transformers import CLIPModel

    m = CLIPModel(geo.hf_configs()).eval()
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(np.array(v, dtype=np.float32)) for k, v in sd.items()},
                                            strict=False)
    assert not unexpected and all(k.endswith("position_ids") or k == "logit_scale" for k in missing)
    return m


def reference_loop(args, net, loader, test_labels, tokenizer):
    """The reference's get_ood_scores_clip body, utils/detection_util.py:219-249, restated (the prompt bank re-encoded
    every batch, the [b,K] softmax taken to the host, numpy / scipy reductions)."""
    from scipy import stats

    _score = []
    with torch.no_grad():
        for images, labels in loader:
            images = images.cuda()
            image_features = net.get_image_features(pixel_values=images).float()
            image_features /= image_features.norm(dim=-1, keepdim=True)
            text_inputs = tokenizer([f"a photo of a {c}" for c in test_labels], padding=True, return_tensors="pt")
            text_features = net.get_text_features(input_ids=text_inputs["input_ids"].cuda(),
                                                  attention_mask=text_inputs["attention_mask"].cuda()).float()
            text_features /= text_features.norm(dim=-1, keepdim=True)
            output = image_features @ text_features.T
            if args.score == "max-logit":
                smax = output.cpu().numpy()
            else:
                smax = torch.softmax(output / args.T, dim=1).cpu().numpy()
            if args.score == "energy":
                _score.append(-(args.T * torch.logsumexp(output / args.T, dim=1)).cpu().numpy())
            elif args.score == "entropy":
                _score.append(stats.entropy(smax, axis=1))
            elif args.score == "var":
                _score.append(-np.var(smax, axis=1))
            else:
                _score.append(-np.max(smax, axis=1))
    return np.concatenate(_score)[:len(loader.dataset)].copy()


@pytest.mark.gpu
@pytest.mark.parametrize("precision", [1, 2], ids=["fp32", "fp16"])
def test_the_documented_stub_runs_the_reference_loop(golden_dir, precision):
    from mcm_amd.engine import LIB_PATH
    from mcm_amd.synth import class_names

    pytest.importorskip("transformers")
    os.environ["MCM_LIB"] = LIB_PATH
    blocks = doc_blocks()
    ns = {}
    exec(compile(blocks[0], "INTEGRATION.md[stub]", "exec"), ns)
    g = np.load(os.path.join(golden_dir, "scores_tiny.npz"))
    K, n_id, n_ood, bs = int(g["K"]), int(g["n_id"]), int(g["n_ood"]), int(g["batch"])
    geo = geometry("tiny")
    hf = _hf_model(geo, synth_state_dict(geo, 0))
    net = ns["NativeNet"](hf, max_batch=bs, precision=precision)
    ids, mask = make_token_ids(K, seed=2)

    class FixedTok:  # the fixture was captured with these ids (no vocabulary offline)
        def __call__(self, texts, padding=True, return_tensors="pt"):
            assert len(texts) == K
            return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}

    try:
        assert net.eval() is net
        l_in = SyntheticLoader(SyntheticImageSet(n_id, geo.image_size, K, False, 1), bs)
        l_out = SyntheticLoader(SyntheticImageSet(n_ood, geo.image_size, K, True, 1), bs)
        tol32 = {s: dict(rtol=5e-5, atol=2e-6) for s in SCORE_KINDS}
        tol32["var"] = dict(rtol=5e-3, atol=1e-9)
        tol16 = {"MCM": dict(rtol=2e-4, atol=1e-6), "max-logit": dict(rtol=0, atol=2e-4), "energy": dict(rtol=2e-4, atol=2e-4),
                 "entropy": dict(rtol=2e-4, atol=1e-5), "var": dict(rtol=5e-2, atol=1e-8)}
        for score in SCORE_KINDS:
            for T in (1, 2):
                args = types.SimpleNamespace(ckpt="ViT-B/16", model="CLIP", score=score, T=T)
                s_in = reference_loop(args, net, l_in, class_names(K), FixedTok())
                s_out = reference_loop(args, net, l_out, class_names(K), FixedTok())
                assert s_in.dtype == np.float32 and s_in.shape == (n_id,) and s_out.shape == (n_ood,)
                tol = (tol32 if precision == 1 else tol16)[score]
                np.testing.assert_allclose(s_in, g[f"{score}_T{T}_in"], **tol)
                np.testing.assert_allclose(s_out, g[f"{score}_T{T}_out"], **tol)

        # block 2 of the document: the fused loop (bank hoisted, one mcm_score per batch), same scores
        args = types.SimpleNamespace(score="MCM", T=1)
        env = dict(ns, net=net, ids=np.ascontiguousarray(ids, dtype=np.int32), K=K, S=ids.shape[1], loader=l_in, args=args,
                   KIND=dict(SCORE_KINDS), _score=[], torch=torch)
        exec(compile(blocks[1], "INTEGRATION.md[loop]", "exec"), env)
        fused = torch.cat(env["_score"]).cpu().numpy()[:n_id]
        np.testing.assert_allclose(fused, g["MCM_T1_in"], **(tol32 if precision == 1 else tol16)["MCM"])
        text = env["text"]

        # block 4: AUROC / AUPR / FPR95 on the device == the reference's get_measures outputs on its own scores
        in_score = torch.cat(env["_score"])[:n_id].contiguous()
        env2 = dict(env, _score=[], loader=l_out)
        exec(compile(blocks[1], "INTEGRATION.md[loop]", "exec"), env2)
        out_score = torch.cat(env2["_score"])[:n_ood].contiguous()
        if precision == 1:
            from ctypes import POINTER, c_double, c_int32, c_int64, c_void_p  # noqa: F401  (block 3 imports them; block 4 assumes it)

            env4 = dict(env, in_score=in_score, out_score=out_score, POINTER=POINTER, c_double=c_double, c_int32=c_int32,
                        c_int64=c_int64, c_void_p=c_void_p)
            exec(compile(blocks[3], "INTEGRATION.md[measures]", "exec"), env4)
            np.testing.assert_allclose([env4["auroc"], env4["aupr"], env4["fpr"]], g["measures_T1"], atol=1e-4)

        # block 3: raw uint8 images -> Resize + CenterCrop on the device -> mcm_score_u8  (tiny geometry: 64-pixel crops,
        # so the document's 224 is the one thing substituted)
        from oracle import oracle as orc

        rng = np.random.default_rng(3)
        raws = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((90, 70), (64, 200), (131, 64))]
        env3 = dict(env, images=[torch.from_numpy(a).cuda() for a in raws])
        exec(compile(blocks[2].replace("224", str(geo.image_size)), "INTEGRATION.md[u8]", "exec"), env3)
        want_crops = np.stack([orc.resize_crop_u8(a, geo.image_size) for a in raws])
        assert np.array_equal(env3["crops"].cpu().numpy(), want_crops)
        mean = np.array([0.48145466, 0.4578275, 0.40821073], np.float32)
        std = np.array([0.26862954, 0.26130258, 0.27577711], np.float32)
        px = ((want_crops.astype(np.float32) / 255.0 - mean) / std).transpose(0, 3, 1, 2).copy()
        o = orc.OracleCLIP(geo, synth_state_dict(geo, 0))
        want = orc.score_features(o.encode_image(px), o.encode_text(ids), 1.0, 0)
        np.testing.assert_allclose(env3["scores"].cpu().numpy(), want, **(tol32 if precision == 1 else tol16)["MCM"])

        # block 5: the split-activation re-scorer (fp16 handles): fp32-grade scores for a window of the ID set
        if precision == 2:
            from ctypes import c_float, c_int32, c_void_p  # noqa: F811

            window_idx = torch.tensor([0, 3, 5, 11], device="cuda")
            allpx = torch.cat([b for b, _ in l_in])[:n_id].cuda()
            env5 = dict(env, window_pixels=allpx[window_idx].contiguous(), n=4, patched=torch.empty(4, device="cuda"),
                        in_score=in_score.clone(), window_idx=window_idx, c_float=c_float, c_int32=c_int32, c_void_p=c_void_p)
            exec(compile(blocks[4], "INTEGRATION.md[x2]", "exec"), env5)
            np.testing.assert_allclose(env5["in_score"][window_idx].cpu().numpy(), g["MCM_T1_in"][[0, 3, 5, 11]], **tol32["MCM"])
        # error behaviour of the contract: a wrong image size is a ValueError, like HF's (modeling_clip.py:204-207)
        with pytest.raises(ValueError):
            net.get_image_features(torch.zeros((1, 3, geo.image_size + 1, geo.image_size), device="cuda"))
        assert text.shape == (K, net.P)
    finally:
        net.close()
