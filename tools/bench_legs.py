"""bench_legs.py — the side legs of bench.py: everything the benchmark reports NEXT TO the timed region (CPU baseline, parity
against HF / the exact-fp32 arm, ingest rates, other arms and configurations, live PMC traffic, clock / power sampling).  None of
it runs inside the timed region; bench.py imports it lazily on rank 0.  Moved out of bench.py in round 5 so that the benchmark
proper (workload, timed loop, roofline, the one short JSON line) reads in one screen."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp16x2": 2500.0, "fp32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(family="gemm"):
    """L2<->fabric bytes per launch of a kernel family (and per GEMM shape class) from the committed rocprofv3
    PMC passes (profiles/*_traffic.json, written by tools/traffic_json.py); None when absent."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None, None
    try:
        per = json.load(open(files[-1]))["per_launch_bytes"]
        shapes = {k: {"l2_fabric_bytes": v["hbm_bytes"], "algorithmic_bytes": v.get("algorithmic_bytes"),
                      "ratio": v.get("ratio")} for k, v in per.items() if k.startswith("gemm_") and "text" not in k}
        return per[family]["hbm_bytes"], os.path.basename(files[-1]), shapes or None
    except Exception:
        return None, None, None


def live_pmc_traffic(args):
    """L2<->fabric bytes per GEMM launch measured IN THIS RUN: two short child runs of this script under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, /opt/skills/guides/MI355X_MICROARCH.md
    HBM section; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 with the gfx950 half-count correction of FETCH_SIZE), summarised by
    tools/traffic_json.py.  Returns (family bytes per launch, per-shape dict, note) — (None, None, why) when rocprofv3 is not
    there or a pass fails (the committed profiles/*_traffic.json is then used and labelled as such)."""
    import importlib.util
    import shutil
    import tempfile

    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, None, "rocprofv3 not found"
    spec = importlib.util.spec_from_file_location("traffic_json", os.path.join(ROOT, "tools", "traffic_json.py"))
    tj = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tj)
    child = [sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--quick", "--no-profile", "--ckpt", args.ckpt,
             "--batch", str(args.batch), "--prompts", str(args.prompts), "--precision", args.precision,
             "--weights-regime", args.weights_regime, "--weight-operands", args.weight_operands]
    env = dict(os.environ, TMPDIR="/tmp")
    found = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(d, counter)
            try:
                r = subprocess.run([rp, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--"] + child, cwd="/tmp", env=env,
                                   capture_output=True, text=True, timeout=240)
            except Exception as e:
                return None, None, f"rocprofv3 {counter} pass: {type(e).__name__}"
            hits = [os.path.join(dp, f) for dp, _d, fs in os.walk(out) for f in fs
                    if f.endswith("_results.db") or f.endswith("counter_collection.csv")]
            if r.returncode or not hits:
                return None, None, f"rocprofv3 {counter} pass failed (rc {r.returncode})"
            try:
                found[counter] = tj.mean_by_family(hits[0], counter)
            except Exception as e:
                return None, None, f"parsing the {counter} pass: {type(e).__name__}: {e}"
    fetch, write = found["FETCH_SIZE"], found["WRITE_SIZE"]
    if "gemm" not in fetch:
        return None, None, "no GEMM dispatches in the counter pass"
    per = {}
    for f, (fb, n) in fetch.items():
        if f == "gemm" or f.startswith("gemm_"):
            wb = write.get(f, (0.0, 0))[0]
            per[f] = {"l2_fabric_bytes": (2 * fb + wb) * 1024, "launches_sampled": n}
            if f in tj.ALGO and args.batch == 512 and args.ckpt == "ViT-B/16":
                per[f]["algorithmic_bytes"] = tj.ALGO[f]
                per[f]["ratio"] = per[f]["l2_fabric_bytes"] / tj.ALGO[f]
    fam = per.pop("gemm")["l2_fabric_bytes"]
    per.pop("gemm_fp32_text_tower", None)
    return fam, per, "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 3 steps each"


def physical_gpu_index(local):
    """Index rocm-smi / sysfs know device ordinal `local` of this process by: the launcher may have narrowed the visible set."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                ids = [int(x) for x in v.split(",") if x.strip() != ""]
                if local < len(ids):
                    return ids[local]
            except ValueError:
                pass
    return local


def _sysfs_card(local):
    """/sys/class/drm/cardN/device of device ordinal `local`, matched by PCI address (torch reports it); None when the
    container does not show amdgpu's sysfs or the address cannot be matched."""
    import glob

    try:
        import torch

        pr = torch.cuda.get_device_properties(local)
        want = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if os.path.basename(os.path.realpath(dev)).lower().startswith(want) and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                return dev
        except OSError:
            pass
    return None


class SmiSampler(threading.Thread):
    """sclk (MHz) and package power (W) of ONE device every `period` s: amdgpu's sysfs files when the container shows them
    (pp_dpm_sclk's starred level, hwmon power1_average / power1_input: no subprocess — every rank of an N-GPU run samples its
    own GPU), else `rocm-smi -d <index>`.  `device` = this process's device ordinal."""

    def __init__(self, period=0.5, device=0):
        super().__init__(daemon=True)
        self.period, self.samples, self._stop_evt = period, [], threading.Event()
        self.device = int(device)
        self.card = _sysfs_card(self.device)
        self.index = physical_gpu_index(self.device)
        self.source = "sysfs " + os.path.realpath(self.card) if self.card else f"rocm-smi -d {self.index}"

    def _read_sysfs(self):
        import glob
        import re

        clk = re.search(r"(\d+)\s*Mhz\s*\*", open(os.path.join(self.card, "pp_dpm_sclk")).read(), re.I)
        for f in sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*", "power1_average")) +
                        glob.glob(os.path.join(self.card, "hwmon", "hwmon*", "power1_input"))):
            try:
                uw = float(open(f).read().strip())
            except (OSError, ValueError):
                continue
            if clk and uw > 0:
                return int(clk.group(1)), uw * 1e-6
        return None

    def _read_smi(self):
        import re

        out = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower"], capture_output=True, text=True,
                             timeout=5).stdout
        clk = re.search(r"sclk clock level.*?\((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\):\s*([0-9.]+)", out)
        return (int(clk.group(1)), float(pw.group(1))) if clk and pw else None

    def run(self):
        while not self._stop_evt.is_set():
            try:
                got = self._read_sysfs() if self.card else None
                if got is None:
                    if self.card:   # the files are there but unreadable / empty: fall back for good
                        self.card, self.source = None, f"rocm-smi -d {self.index}"
                    got = self._read_smi()
                if got:
                    self.samples.append(got)
            except Exception:
                pass
            self._stop_evt.wait(self.period)

    def stop(self):
        self._stop_evt.set()
        self.join(timeout=10)
        # busy samples: within 30 % of the highest power seen (an idle MI355X draws 250 - 300 W, a loaded one up to 1.4 kW; a
        # fixed 300-W line mistook idle samples of a warm part for load, and dropped every sample of a light workload)
        top = max((s[1] for s in self.samples), default=0.0)
        busy = [s for s in self.samples if s[1] >= 0.7 * top]
        out = {"samples": len(self.samples), "busy_samples": len(busy), "source": self.source}
        if busy:
            out.update(sclk_mhz_mean=sum(s[0] for s in busy) / len(busy), power_w_mean=sum(s[1] for s in busy) / len(busy))
        return out


def cpu_baseline(geo, sd, ids, mask, K, px_batches, max_seconds, native_first):
    """The reference loop on the host cores (utils/detection_util.py:219-248): per batch, image features →
    normalise → (re-)encode the K prompts → normalise → matmul → softmax → -max, fp32 torch CPU through HF
    CLIPModel.  BASELINE.md §3: one warm-up batch, then >= 256 images at batch 64.  The image part and the
    text part of every batch are timed separately, so one pass gives both figures: `value` (what the
    reference does: text bank re-encoded per batch) and `value_hoisted` (bank encoded once)."""
    import numpy as np
    import torch

    from mcm_amd.hostinfo import cpu_quota, effective_cpus

    # torch defaults to one thread per visible core; a container is scheduled on its cgroup quota (the GPU boxes here: 256
    # CPUs visible, 16 cores of quota), and threads beyond it only take turns.  `cores` = the threads used = that allocation.
    threads_default = torch.get_num_threads()
    torch.set_num_threads(min(threads_default, effective_cpus()))
    info = {"cores": torch.get_num_threads(), "host_cpus": os.cpu_count(), "cpu_quota_cores": cpu_quota(),
            "torch_default_threads": threads_default, "unit": "images/sec"}
    bs = px_batches[0].shape[0]
    try:
        from oracle.hf_reference import HFReference

        h = HFReference(geo, sd, device="cpu")
        h.set_bank(ids, mask)

        def image_part(px):   # features, normalise, similarity, softmax, -max against the current bank
            return h.score_batch(px, 1.0, "MCM").numpy()

        def text_part():      # the loop-invariant work the reference repeats every batch (:228-231)
            h.set_bank(ids, mask)

        kind = "reference"
    except Exception as e:  # transformers missing on the box: time the C oracle instead
        from oracle import oracle as orc

        o = orc.OracleCLIP(geo, sd)
        bank = {"t": o.encode_text(ids)}

        def image_part(px):
            return orc.score_features(o.encode_image(px.numpy()), bank["t"], 1.0, 0)

        def text_part():
            bank["t"] = o.encode_text(ids)

        kind = "port"
        info["note"] = f"transformers unavailable ({type(e).__name__}); C oracle timed"
    t0 = time.perf_counter()
    text_part()
    image_part(px_batches[0][:8])  # warm-up: thread pool, allocator, oneDNN primitive caches (8 images: the timed batches follow)
    warm = time.perf_counter() - t0
    n, t_img, t_txt, i = 0, 0.0, 0.0, 0
    target = 256
    first = None
    while n < target and (t_img + t_txt) < max_seconds:
        px = px_batches[i % len(px_batches)]
        t0 = time.perf_counter()
        text_part()
        t1 = time.perf_counter()
        got = image_part(px)
        t2 = time.perf_counter()
        if first is None:
            first = got  # batch 0 = the pixels the native run scored first: the parity check below
        t_txt += t1 - t0
        t_img += t2 - t1
        n += px.shape[0]
        i += 1
    info.update(value=n / (t_img + t_txt) if n else None, value_hoisted=n / t_img if n else None, kind=kind,
                seconds={"warmup_batch": warm, "image_part": t_img, "text_part": t_txt},
                sample=f"{n} images, batch {bs} (K={K} prompts; value: bank re-encoded per batch as the reference "
                       f"does, value_hoisted: bank encoded once) after a warm-up (bank + 8 images) of {warm:.1f} s; same seeded "
                       f"weights and pixels as the native run" + ("" if n >= target else
                                                                  f"; stopped at the {max_seconds:.0f} s cap"))
    torch.set_num_threads(threads_default)
    if native_first is not None and first is not None:
        d = np.abs(first - native_first)
        info["parity_max_abs_dscore_vs_native"] = float(d.max())
        info["parity_images"] = int(d.size)
    return info


def parity_leg(args, K, B, device):
    """AUROC / AUPR / FPR95 of every native arm against the exact-fp32 arm AND against the HF CLIPModel fp32
    reference running on the same device over the same 50 000 + 10 000 device-generated images, in both weight
    regimes.  Outside the timed region; the HF scorer is the checker (oracle/hf_reference.py), never measured."""
    from mcm_amd.parity import CONFIG3_OOD_SETS, HEADLINE_PIXELS, measure_drift

    external, hf_note = None, None
    if not args.no_hf:
        try:
            from oracle.hf_reference import hf_available, hf_scorer_factory

            why = hf_available()
            if why is None:
                external = {"hf": hf_scorer_factory()}
            else:
                hf_note = f"transformers unavailable on this box ({why}): vs_hf not measured"
        except Exception as e:
            hf_note = f"HF reference scorer unavailable ({type(e).__name__}: {e}): vs_hf not measured"
    from mcm_amd.parity import REALISTIC_PIXELS, meets_bar

    arms = tuple(dict.fromkeys((args.precision, "fp16", "bf16", "fp16x2", "fp16+refine", "fp16+refine2")))
    c3 = tuple(args.drift_n) == (50000, 10000)  # default: BASELINE config 3 — ImageNet-1k vs the four OOD sets
    ood_sets = CONFIG3_OOD_SETS if c3 else None
    out = {"config": "BASELINE config 3: ImageNet-1k-sized ID set (50 000) vs iNaturalist / SUN / Places / Textures-sized "
                     "OOD sets (10 000 / 10 000 / 10 000 / 5 640), K = 1000; headline keys = the AVG row of the reference's "
                     "CSV, per_set = every OOD set on its own" if c3 else "one ID and one OOD set (--drift-n)",
           "n_id": args.drift_n[0], "n_ood": {n: c for n, c, _ in CONFIG3_OOD_SETS} if c3 else args.drift_n[1],
           "pixels": {k: HEADLINE_PIXELS[k] for k in ("amp", "tile")},
           "bar": "north_star: |dAUROC|, |dFPR95| <= 1e-4.  FPR95 of ONE set is a count of images on the ID side of one "
                  "threshold (quantum 1e-4 at 10 000 images): per_set carries it as d_fpr95_images",
           "reference_arms": "exact-fp32 MFMA arm of this library; HF transformers CLIPModel fp32 eager on this device"}
    if hf_note:
        out["vs_hf_note"] = hf_note
    keys = ("d_auroc", "d_aupr", "d_fpr95", "max_abs_dscore", "rms_dscore")
    regimes = [r for r in (("fp16_exact_weights", "fp16-exact"), ("fp32_valued_weights", "fp32"))
               if r[1] in args.parity_regimes.split(",")]
    for regime, weights in regimes:
        t0 = time.perf_counter()
        # HF on the device scores the first regime only (35 s per regime at config 3's sizes): the others are held against the
        # exact-fp32 arm, itself pinned to HF by the first
        # fp32-valued weights: the 16-bit arms run the split-weight GEMMs (weight_operands auto); "fp16:single" is what
        # rounds 1 - 3 did there (one rounded operand per weight), kept as the comparison
        arms_w = arms + (("fp16:single",) if weights == "fp32" else ())
        d = measure_drift(args.ckpt, K=K, n_id=args.drift_n[0], n_ood=args.drift_n[1], batch=B, arms=arms_w,
                          device=device, amp=HEADLINE_PIXELS["amp"], tile=HEADLINE_PIXELS["tile"], weights=weights,
                          external=external if regime == regimes[0][0] else None, ood_sets=ood_sets)
        r = {"auroc_fp32_arm": d["reference"]["auroc"], "fpr95_fp32_arm": d["reference"]["fpr95"],
             "score_std_id": d["reference"]["score_std_id"], "seconds": time.perf_counter() - t0,
             "fp16_saturation_events": d["fp16_saturation_events"].get("fp16"),
             "weight_operands": d["weight_operands"], "refine": d.get("refine"),
             "vs_fp32_arm": {p: {k: d["arms"][p][k] for k in keys + ("max_set",) + (("per_set",) if c3 else ())} for p in arms_w}}
        if "external" in d:
            r["auroc_hf"], r["fpr95_hf"] = d["external"]["hf"]["auroc"], d["external"]["hf"]["fpr95"]
            r["vs_hf"] = {"fp32_arm": d["reference"]["vs_external"]["hf"],
                          **{p: d["arms"][p]["vs_external"]["hf"] for p in arms_w}}
        out[regime] = r
    done = [r for r, _ in regimes]
    head = out[done[0]]
    # headline keys: the benchmarked dtype in the first regime (default fp16-exact weights: the reference's checkpoints' regime)
    out["weights"] = f"{regimes[0][1]} (headline keys below)" + ("; other regimes under their own keys" if len(done) > 1 else
                                                                "; the fp32-valued regime: --parity-regimes fp16-exact,fp32 "
                                                                "and tests/test_gpu_headline_parity.py")
    out["vs"] = "HF CLIPModel fp32 on this device" if "vs_hf" in head else "exact-fp32 MFMA arm"
    src = head["vs_hf"][args.precision] if "vs_hf" in head else head["vs_fp32_arm"][args.precision]
    out.update({k: src[k] for k in keys})
    # judged PER OOD SET (the AVG row lets opposite-sign drifts cancel): |dAUROC|, |dAUPR| <= 1e-4 on every set and FPR95
    # within N images of the reference on every set — N = 1 is the quantum of a 10 000-image set; on this ordering-stress
    # set a 16-bit arm's activation rounding moves 0 - 2 images depending on the draw
    for key, n_img in (("meets_1e-4", 1), ("meets_1e-4_fpr95_within_2_images", 2)):
        out[key] = {w: {p: meets_bar(v, 1e-4, n_img) for p, v in out[w]["vs_hf" if "vs_hf" in out[w] else "vs_fp32_arm"].items()}
                    for w in done}
    # the realistic operating point (mcm_amd/parity.py REALISTIC_PIXELS): reference AUROC 0.9, score noise ~0.1 % of the spread
    if c3:
        t0 = time.perf_counter()
        d = measure_drift(args.ckpt, K=K, n_id=10000, n_ood=10000, batch=500, arms=arms, device=device,
                          amp=REALISTIC_PIXELS["amp"], tile=REALISTIC_PIXELS["tile"], tile_ood=REALISTIC_PIXELS["tile_ood"],
                          weights="fp16-exact", operating_point=0.9)
        out["operating_point_auroc_0.9"] = dict(d["operating_point"], seconds=time.perf_counter() - t0,
                                                pixels=d["pixels"], vs="exact-fp32 MFMA arm")
    return out


def ingest_legs(net, txt, B, steps, which):
    """uint8 host → scores, end to end (§7 hard part 4).  `host_u8`: 224² uint8 crops sitting in PINNED
    host memory → double-buffered asynchronous copies on a copy stream → mcm_score_u8 (ToTensor + Normalize fused into the
    patch gather).  `host_raw`: variable-size decoded RGB images (an ImageNet-like size mix, mean ≈ 0.5 MB) → packed into
    one pinned buffer per batch by the host → ONE copy per batch → mcm_resize_crop_u8 (Resize 224 + CenterCrop 224,
    bit-exact vs Pillow) → mcm_score_u8.  JPEG decode itself is host-CPU work outside this path.  Outside the timed
    region of the headline number; reported next to it."""
    import numpy as np
    import torch

    from mcm_amd.ingest import PackedImagePipe, PinnedBatchPipe

    S = net.geo.image_size
    out = {}
    sc = torch.empty(B, device=net.device)
    if "host-u8" in which:
        g = torch.Generator().manual_seed(7)
        host = [torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, generator=g).pin_memory() for _ in range(3)]
        pipe = PinnedBatchPipe(net, B)
        for px in pipe.stream(host[:2]):  # warm-up: pinned buffers, copy stream, the u8 patchify kernel
            net.score_images(px, txt, 1.0, "MCM", out=sc)
        torch.cuda.synchronize()
        b0, t0 = pipe.bytes_copied, time.perf_counter()
        for px in pipe.stream(host[i % 3] for i in range(steps)):
            net.score_images(px, txt, 1.0, "MCM", out=sc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["host_u8"] = {"images_per_sec": steps * B / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
                          "pcie_gb_per_sec": (pipe.bytes_copied - b0) / dt / 1e9, "bytes_per_image": S * S * 3,
                          "source": "uint8 [B,224,224,3] crops in pinned host memory, one async copy per batch on a copy "
                                    "stream, 3 device buffers"}
        del pipe, host
    if "host-raw" in which:
        rng = np.random.default_rng(11)
        sizes = [(375, 500), (500, 375), (333, 500), (500, 333), (480, 640), (400, 400), (256, 341), (600, 800)]
        base = {hw: rng.integers(0, 256, size=(hw[0], hw[1], 3), dtype=np.uint8) for hw in sizes}
        batch = [base[sizes[i % len(sizes)]] for i in range(B)]
        nbytes = PackedImagePipe.packed_bytes(batch)
        pipe = PackedImagePipe(net, B, nbytes + (1 << 20), pack_threads=int(os.environ.get("MCM_PACK_THREADS", min(16, os.cpu_count() or 1))))
        for px in pipe.stream([batch, batch]):
            net.score_images(px, txt, 1.0, "MCM", out=sc)
        torch.cuda.synchronize()
        b0, t0 = pipe.bytes_copied, time.perf_counter()
        for px in pipe.stream(batch for _ in range(steps)):
            net.score_images(px, txt, 1.0, "MCM", out=sc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["host_raw"] = {"images_per_sec": steps * B / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
                           "pcie_gb_per_sec": (pipe.bytes_copied - b0) / dt / 1e9, "bytes_per_image": nbytes / B,
                           "pack_threads": pipe.pack_threads,
                           "source": "decoded RGB images of 8 sizes (256x341 ... 600x800) in pageable host memory, packed "
                                     "into one pinned buffer and copied once per batch, Resize + CenterCrop on the device"}
        del pipe
    if "host-jpeg" in which:
        # the CLI's own loader on an image folder of JPEG files: file read + Pillow decode in the loader's worker processes (host
        # work, like the reference's DataLoader workers) -> packed copy -> Resize + CenterCrop + scoring on the device
        try:
            import shutil
            import tempfile

            from PIL import Image

            from mcm_amd.folder import ImageFolderU8

            rng = np.random.default_rng(13)
            sizes = [(375, 500), (500, 375), (333, 500), (500, 333), (480, 640), (400, 400), (256, 341), (600, 800)]
            root = tempfile.mkdtemp(prefix="mcm_jpeg_")
            try:
                import io

                nfiles, fbytes = 16 * B, 0   # (a pass of 16 batches: the first batch of a pass pays for the pipe's start)
                yy, xx = np.mgrid[0:800, 0:800].astype(np.float32)
                blobs = []
                for i in range(2 * len(sizes)):  # photograph-like content: smooth structure + texture (noise alone does not compress)
                    h, w = sizes[i % len(sizes)]
                    f = rng.uniform(0.01, 0.06, 6)
                    im = np.stack([127 + 70 * np.sin(f[2 * c] * xx[:h, :w] + i) * np.cos(f[2 * c + 1] * yy[:h, :w]) for c in range(3)], -1)
                    im = np.clip(im + rng.normal(0, 12, im.shape), 0, 255).astype(np.uint8)
                    buf = io.BytesIO()
                    Image.fromarray(im).save(buf, format="JPEG", quality=90)
                    blobs.append(buf.getvalue())
                for c in range(8):
                    os.makedirs(os.path.join(root, f"class{c}"))
                for j, blob in enumerate(blobs):
                    with open(os.path.join(root, f"blob{j}.bin"), "wb") as fh:
                        fh.write(blob)
                for i in range(nfiles):  # hard links to the 16 files: a folder of 8 192 entries without 700 MB of writes
                    os.link(os.path.join(root, f"blob{i % len(blobs)}.bin"), os.path.join(root, f"class{i % 8}", f"{i:05d}.jpg"))
                    fbytes += len(blobs[i % len(blobs)])
                loader = ImageFolderU8(root, net, B)  # (MCM_DECODE_WORKERS overrides the loader's own choice: its CPU quota)
                route = ("entropy decode on host threads, inverse DCT + upsampling + colour on the device"
                         if os.environ.get("MCM_GPU_JPEG", "0") == "1" else "Pillow in worker processes (the default)")
                for px, _ in loader:  # warm-up pass: page cache, thread pool, slots
                    net.score_images(px, txt, 1.0, "MCM", out=sc[: px.shape[0]])
                torch.cuda.synchronize()
                passes, t0 = 1, time.perf_counter()
                for _ in range(passes):
                    for px, _ in loader:
                        net.score_images(px, txt, 1.0, "MCM", out=sc[: px.shape[0]])
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                pipe = next(iter(net.__dict__.get("_jpeg_pipes", {}).values()), None)
                loader.close()
                out["host_jpeg"] = {"images_per_sec": passes * nfiles / dt, "ms_per_step": 1e3 * dt / (passes * nfiles / B),
                                    "steps": passes * nfiles // B, "decode_workers": loader.workers, "host_cpus": os.cpu_count(),
                                    "cpu_quota_cores": __import__("mcm_amd.hostinfo", fromlist=["cpu_quota"]).cpu_quota(),
                                    "jpeg_bytes_per_image": fbytes / nfiles,
                                    "decoder": route,
                                    "files_decoded_by_pillow_inside_the_pipe": (pipe.fallback_images if pipe is not None else None),
                                    "pipe_seconds_per_batch": ({k: round(v / max(1, pipe.stats["batches"]), 4) for k, v in pipe.stats.items()
                                                                if k != "batches"} if pipe is not None else None),
                                    "source": f"{nfiles} JPEG files (quality 90, 8 sizes 256x341 ... 600x800) in an image folder, read + "
                                              "decoded by the CLI's loader (see decoder), Resize + CenterCrop + scoring on the device; "
                                              "bound by the host cores this container is given (decode_workers = its CPU quota)"}
            finally:
                shutil.rmtree(root, ignore_errors=True)
        except ImportError as e:
            out["host_jpeg"] = {"skipped": f"Pillow unavailable ({e})"}
    return out


def arm_leg(geo, sd, precision, weight_operands, B, K, ids, px, device, steps=10, x2=False):
    """Throughput of one more arm on the same workload, `steps` timed steps after two warm-up steps, outside the timed region
    of the headline number: (images/s, GEMM-family TFLOP/s by HIP events, fraction of that dtype's dense MFMA peak)."""
    import torch

    from mcm_amd.engine import NativeCLIP

    net = NativeCLIP(geo, sd, device=device, precision=precision, max_batch=B, weight_operands=weight_operands,
                     max_prompt_tokens=max(K * ids.shape[1], 77))
    try:
        txt = net.get_text_features(input_ids=torch.from_numpy(ids), normalize=True)
        out = torch.empty(B, device=px.device)
        run = net.score_images_x2 if x2 else net.score_images   # x2: the split-activation arm (chunks of mcm_x2_max_batch)
        run(px, txt, 1.0, "MCM", out=out)
        run(px, txt, 1.0, "MCM", out=out)
        net.profile(True)   # (creates the event pool outside the timed region)
        run(px, txt, 1.0, "MCM", out=out)
        net.profile_read()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            net.profile(i % 4 == 0)   # kernel events on every 4th step, as in bench.py's timed region: bracketing every launch of
            run(px, txt, 1.0, "MCM", out=out)   # every step costs 2.3 % (round 5's legs did, which is why they sat below `value`)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        g = net.profile_read()["gemm"]
        ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] else None
        peak = MFMA_PEAK_TFLOPS[precision]
        return {"images_per_sec": steps * B / dt, "ms_per_step": 1e3 * dt / steps, "steps": steps,
                "gemm_tflops": ach, "peak_tflops": peak, "frac": ach / peak if ach else None,
                "split_weight_gemms": net.split_weights, "finite": bool(torch.isfinite(out).all())}
    finally:
        net.close()


class ResidentSet:
    """One data set of BASELINE config 3 held in HBM as fp32 NCHW pixels (85 640 images = 51.6 GB of the part's 288 GB): what
    `value` assumes (inputs resident when the timed region starts), for a whole evaluation.  Iterates in batches and gathers by
    index (threshold refinement re-scores named images)."""

    def __init__(self, loader, batch):
        import torch

        self.dataset = loader.dataset
        self.batch_size = batch
        self.px = torch.cat([px for px, _ in loader])

    def __len__(self):
        return -(-self.px.shape[0] // self.batch_size)

    def __iter__(self):
        for s in range(0, self.px.shape[0], self.batch_size):
            yield self.px[s:s + self.batch_size], None

    def gather(self, idx):
        import torch

        return self.px[torch.as_tensor(idx, device=self.px.device, dtype=torch.long)]


def refined_leg(net, geo, sd, txt, ids, mask, B, K, device, two_level=True):
    """Throughput AT PARITY (`value_refined`): BASELINE config 3's sizes — 50 000 ID images once against the four OOD sets
    (10 000 / 10 000 / 10 000 / 5 640), all resident in HBM — scored by the benchmarked arm AND threshold-refined
    (mcm_amd/refine.py), then the three metrics per set on the device.  Wall clock of all of it / 85 640 images.
    The re-scorer is the SPLIT-ACTIVATION arm of the same handle (mcm_score_x2: no second model); with `two_level` the
    handful of images within a few fp32 ulps of the threshold additionally go through an exact-fp32 handle (created before
    the clock starts, like the scoring handle), so that the reported FPR95 is that arm's image for image.  A bf16 run has no
    split-activation arm: the exact-fp32 handle re-scores its whole window.  Afterwards (untimed) the exact arm scores
    every image and the refined FPR95 is checked against it: `fpr95_images_vs_fp32_arm_max_set` must be 0."""
    import torch

    from mcm_amd.engine import NativeCLIP
    from mcm_amd.parity import CONFIG3_OOD_SETS, HEADLINE_PIXELS
    from mcm_amd.refine import Rescorer, ThresholdRefiner
    from mcm_amd.synth import DevicePatternLoader

    dev = torch.device("cuda", device)
    sets = [("id", 50000, False, 1)] + [(n, c, True, s) for n, c, s in CONFIG3_OOD_SETS]
    t0 = time.perf_counter()
    data = {n: ResidentSet(DevicePatternLoader(c, geo.image_size, K, B, dev, ood=ood, seed=s, amp=HEADLINE_PIXELS["amp"],
                                               tile=HEADLINE_PIXELS["tile"]), B) for n, c, ood, s in sets}
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    exact = NativeCLIP(geo, sd, device=device, precision="fp32", max_batch=min(B, 256), max_prompt_tokens=max(K * ids.shape[1], 77))
    use_x2 = net.x2_max_batch > 0
    try:
        n_img = sum(c for _, c, _, _ in sets)

        def score_set(scorer, name):
            out = torch.empty(data[name].px.shape[0], device=dev)
            for i, (px, _) in enumerate(data[name]):
                scorer.score_images(px, txt, 1.0, "MCM", out=out[i * B:i * B + px.shape[0]])
            return out

        score_set(net, "dtd")  # warm-up of the pass (clocks, the kernels' first launches at this batch)
        if use_x2:
            net.score_images_x2(data["dtd"].px[:net.x2_max_batch], txt)
        exact.score_images(data["dtd"].px[:32], txt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scores = {n: score_set(net, n) for n, _, _, _ in sets}
        torch.cuda.synchronize()
        t_score = time.perf_counter() - t0
        def refine(two):  # on a copy of the 16-bit scores: (seconds, refined scores, measures per set, refiner stats, part times)
            sc = {n: v.clone() for n, v in scores.items()}
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            first = Rescorer(net.x2_scorer() if use_x2 else exact, txt, data, 1.0, "MCM")
            second = Rescorer(exact, txt, data, 1.0, "MCM") if (use_x2 and two) else None
            refiner = ThresholdRefiner(first, rescore_exact=second)
            refiner.fit_id(sc["id"])
            torch.cuda.synchronize()
            t_fit = time.perf_counter() - t1
            meas, t_meas = {}, 0.0
            for n, _, ood, _ in sets:
                if ood:
                    refiner.apply(n, sc[n])
                    torch.cuda.synchronize()
                    tm = time.perf_counter()
                    meas[n] = net.measures(sc["id"], sc[n], negate=True)
                    t_meas += time.perf_counter() - tm
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            return dt, meas, refiner.stats, {"id_calibration_and_window": t_fit, "device_metrics_4_sets": t_meas,
                                             "ood_windows": dt - t_fit - t_meas}

        # `value_refined`: what the CLI does by default — the split-activation arm of the same handle re-scores the window
        # (one level); `exact`: additionally the inner window through the exact-fp32 handle (--refine-threshold exact)
        refine(two_level)   # warm-up (untimed, like the scoring pass's): first launches of the small-batch kernels the windows
                            # run, torch's sort / nonzero — 0.25 s the first time, nothing afterwards
        t_ref, meas, st, parts = refine(False)
        t_all = t_score + t_ref
        out = {"images_per_sec": n_img / t_all, "images": n_img, "seconds": t_all, "seconds_scoring": t_score,
               "seconds_refine": t_ref, "images_per_sec_unrefined": n_img / t_score, "seconds_refine_parts": parts,
               "rescored": st.get("rescored_total"), "rescored_per_set": st["rescored"],
               "rescorer": "split-activation arm of the same handle (mcm_score_x2)" if use_x2 else "exact-fp32 handle",
               "delta": st["delta"], "noise_max_abs": st["noise_max_abs"], "threshold_interval": st.get("threshold_interval"),
               "seconds_generating_pixels": t_gen, "measures": {n: list(m) for n, m in meas.items()},
               "workload": "BASELINE config 3 sizes, fp32 NCHW pixels resident in HBM, batch %d; timed: scoring of the 5 sets + "
                           "threshold refinement + device metrics" % B}
        meas2 = None
        if use_x2 and two_level:
            t_ref2, meas2, st2, parts2 = refine(True)
            out["exact"] = {"images_per_sec": n_img / (t_score + t_ref2), "seconds_refine": t_ref2, "seconds_refine_parts": parts2,
                            "rescored": st2.get("rescored_total"), "rescored_exact": st2.get("rescored_exact_total"),
                            "rescored_exact_per_set": st2.get("rescored_exact"), "delta2": st2.get("delta2"),
                            "noise2_max_abs": st2.get("noise2_max_abs"),
                            "rescorer": "split-activation arm, then the exact-fp32 handle for the inner window"}
            out["rescored_exact"] = st2.get("rescored_exact_total")
        # the check (untimed): every image through the exact arm, FPR95 per set against the refined scores'
        ref = {n: score_set(exact, n) for n, _, _, _ in sets}
        m_ref = {n: net.measures(ref["id"], ref[n], negate=True) for n, c, ood, _ in sets if ood}

        def moved(ms):
            return {n: {"fpr95_images": round(abs(m_ref[n][2] - ms[n][2]) * c), "d_auroc": abs(m_ref[n][0] - ms[n][0])}
                    for n, c, ood, _ in sets if ood}

        out["vs_fp32_arm"] = moved(meas)
        out["fpr95_images_vs_fp32_arm_max_set"] = max(v["fpr95_images"] for v in out["vs_fp32_arm"].values())
        if meas2 is not None:
            out["exact"]["vs_fp32_arm"] = moved(meas2)
            out["exact"]["fpr95_images_vs_fp32_arm_max_set"] = max(v["fpr95_images"] for v in out["exact"]["vs_fp32_arm"].values())
        return out
    finally:
        exact.close()
        del data
        torch.cuda.empty_cache()


def config_legs(device, steps=12):
    """BASELINE configs 4 and 2 on this device, `steps` timed steps each after two warm-up steps (like `arms`; 12 since round 6 —
    3 steps could not carry a 2 % claim): ViT-L/14 fp16 batch 256 K = 1000 (config 4's per-GPU work) and
    ViT-B/16 K = 100 batch 512 in bf16 (the dtype config 2 names) and fp16; plus configs 4 and 3 at the nearest full-round batch
    (255, 665) NEXT TO the same work at BASELINE's batch measured by the same leg (c3_B16_fp16_b512: the control of the
    full-round-batch claim, EXPERIMENTS.md R5.11 / R6)."""
    import torch

    from mcm_amd.config import geometry
    from mcm_amd.synth import make_token_ids
    from mcm_amd.weights import synth_state_dict

    out = {}
    for name, ckpt, prec, B, K in (("c4_L14_fp16_b256", "ViT-L/14", "fp16", 256, 1000),
                                   ("c2_B16_K100_bf16", "ViT-B/16", "bf16", 512, 100),
                                   ("c2_B16_K100_fp16", "ViT-B/16", "fp16", 512, 100),
                                   # the same work at a batch whose GEMMs fill every tile round (ClipGeometry.full_round_batches,
                                   # EXPERIMENTS.md R5.11): what a caller free to choose its batch gets per image
                                   ("c4_L14_fp16_b255", "ViT-L/14", "fp16", 255, 1000),
                                   ("c3_B16_fp16_b512", "ViT-B/16", "fp16", 512, 1000),
                                   ("c3_B16_fp16_b665", "ViT-B/16", "fp16", 665, 1000)):
        try:
            geo = geometry(ckpt)
            sd = synth_state_dict(geo, 0, "fp16-exact")
            ids, _ = make_token_ids(K, seed=2)
            px = torch.randn((B, 3, geo.image_size, geo.image_size), device=torch.device("cuda", device),
                             generator=torch.Generator(device=torch.device("cuda", device)).manual_seed(99))
            # bf16 on fp16-exact weights: one rounded operand per weight is what BASELINE.md's dtype means (arms.bf16_single_operand)
            out[name] = arm_leg(geo, sd, prec, "single" if prec == "bf16" else "auto", B, K, ids, px, device, steps=steps)
            out[name].update(ckpt=ckpt, batch=B, prompts=K)
            del px, sd
        except Exception as e:
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        torch.cuda.empty_cache()
    return out
