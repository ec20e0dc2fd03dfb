"""bench.py — images/sec MCM-scored on MI355X (BASELINE.json's metric).

A "step" is one pass of the hot path (reference utils/detection_util.py:223-248) over one
batch of synthetic, device-resident, already-normalised fp32 pixels: vision tower → cosine
vs the pre-encoded prompt bank → softmax/T → max → [B] scores, through the C ABI of
libmcm_hip.so.  Workload at any N: CLIP-ViT-B/16, K=1000 prompts (the ImageNet-1k concept
bank the metric is quoted on), batch 512 per GPU, 16-bit MFMA operands / fp32 accumulate;
random-init weights of that architecture (no checkpoint offline).

N > 1: one process per GPU.  Launched by the driver under `torch.distributed.run` (RANK / LOCAL_RANK /
WORLD_SIZE in the env) or, when `--gpus N` is given WITHOUT that env, bench.py re-executes itself under
`torch.distributed.run --nproc-per-node N` — `--gpus N` never silently measures one GPU.  Images are
sharded with no data-path collective; the per-dataset all-gather of the score shards (RCCL) is inside
the timed region (weak scaling).  If the box has fewer devices than ranks (a 1-GPU box running the
2-rank logic check) the ranks share devices and the collective falls back to gloo, because RCCL refuses
two ranks on one device; the JSON line says so (`"collective"`).

Output (rank 0): ONE JSON line of at most 4 KB (`short_line`; tests/test_bench_line.py holds it to that) — the contract
keys, `roofline`, `cpu_baseline` and one scalar per side leg — printed last on stdout; every leg's full record goes to
`bench_detail.json` beside this file (and to gpurun_out/ when that directory exists).  Legs (tools/bench_legs.py), all
outside the timed region, N = 1 only:
  roofline      GEMM kernel family: algorithmic FLOP ÷ HIP-event time of the launches of every 4th timed
                step (`profiled_steps`; bracketing every launch of every step costs 2.3 % of throughput);
                `traffic` from two rocprofv3 --pmc child passes of this script;
  cpu_baseline  the reference's own arithmetic — HF transformers CLIPModel, fp32 — driven by a re-statement
                of the reference loop on the host cores (BASELINE.md §3 protocol);
  sustained     the same step repeated for >= 5 s with sclk / package power sampled through rocm-smi;
  refined       `value_refined`: BASELINE config 3's sizes (50 000 ID + 35 640 OOD device-resident images) scored AND
                threshold-refined (mcm_amd/refine.py: the route that holds FPR95 to the reference), wall-clock images/sec;
  arms/configs  3 timed steps each of the other precision arms and of BASELINE configs 2 and 4 (K = 100; ViT-L/14);
  parity        AUROC / AUPR / FPR95 on config 3 of every native arm against the exact-fp32 arm and against the HF
                CLIPModel fp32 reference on this device over the same pixels (mcm_amd/parity.py).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp16x2": 2500.0, "fp32": 157.3}  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
DEFAULT_PRECISION = "fp16"  # the 16-bit mode that holds AUROC/FPR95 to the fp32 arm
LINE_LIMIT = 4096           # bytes of the printed line (a 33.6 KB line left the driver with parsed = null)
# harness A/B switches (libmcm_hip_harness.so, mcm_debug_*): `--harness gemm_variant=3,ln_tail=1`; never the shipped policy
HARNESS_KEYS = {"gemm_variant": "mcm_debug_gemm_variant", "attn_variant": "mcm_debug_attention_variant",
                "ln_fold": "mcm_debug_ln_fold", "ln_tail": "mcm_debug_ln_tail", "ln_cluster": "mcm_debug_ln_cluster", "patch_fold": "mcm_debug_patch_fold",
                "group_n": "mcm_debug_gemm_group_n", "nsplit": "mcm_debug_nsplit", "qkv_chunks": "mcm_debug_qkv_chunks",
                "gemm_dbg": "mcm_debug_gemm_dbg", "persistent_grid": "mcm_debug_persistent_grid",
                "ln_cluster_spin": "mcm_debug_ln_cluster_spin", "ln_row": "mcm_debug_ln_row"}


def _r(x, sig=5):
    """Floats to `sig` significant digits (the line is read by a parser, the full precision is in bench_detail.json)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{sig}g}")
    except (TypeError, ValueError):
        return None


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def short_line(d):
    """The ONE line the driver parses: the contract keys verbatim, `roofline` / `cpu_baseline` as the task defines them, and
    one scalar (or a flat dict of scalars) per side leg.  `d` is the full record (what bench_detail.json holds).  Anything
    that could grow with the number of OOD sets, arms or shapes stays in the detail file."""
    line = {k: d.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                  "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 7), _r(line["ms_per_step"], 6)
    line["config"] = {k: _get(d, "config", k) for k in ("workload", "batch_per_gpu", "prompts", "parallelism")}
    if d.get("collective"):
        line["collective"] = str(d["collective"])[:120]
    for k in ("harness", "note"):
        if d.get(k):
            line[k] = str(d[k])[:160]
    for k in ("lnc_timeouts", "lnc_deferred_segments"):
        if k in d:
            line[k] = d[k]
    if d.get("kernel_faults"):
        line["kernel_faults"] = d["kernel_faults"]
    if d.get("per_rank"):   # N > 1: one compact record per rank — own images/s in the timed region, own sustained clock and power
        line["per_rank"] = [{"img_s": _r(r.get("img_s")), "sclk_mhz": _r(r.get("sclk_mhz"), 4), "power_w": _r(r.get("power_w"), 4),
                             "ag_ms": _r(r.get("ms_allgather"), 3)} for r in d["per_rank"]]
        line["allgather_ms"] = [_r(d.get("allgather_ms_min"), 3), _r(d.get("allgather_ms"), 3)]   # [min, max] over ranks
    # throughput AT PARITY and the exact-grade split-activation arm, each with its own end-to-end fraction of the dense MFMA peak
    # (images/s x nominal GFLOP per image / peak: SURVEY section 8d's definition) — always-kept keys
    gpi, peak = d.get("gflop_per_image"), MFMA_PEAK_TFLOPS.get(d.get("dtype"), 2500.0)
    x2 = _get(d, "arms", "fp16x2_split_activations", "images_per_sec")
    if x2 and gpi:
        line["value_fp16x2"], line["frac_fp16x2"] = _r(x2, 7), _r(x2 * gpi / 1e3 / MFMA_PEAK_TFLOPS["fp16x2"], 3)
    rf = d.get("refined")
    if rf:
        line["value_refined"] = _r(rf.get("images_per_sec"), 7)
        if rf.get("images_per_sec") and gpi:
            line["frac_refined"] = _r(rf["images_per_sec"] * gpi / 1e3 / peak, 3)
        line["refined"] = {k: _r(rf.get(k)) for k in ("images", "rescored", "seconds", "seconds_refine", "rescorer",
                                                       "fpr95_images_vs_fp32_arm_max_set", "error") if k in rf}
        if rf.get("exact"):  # --refine-threshold exact: the inner window also through an exact-fp32 handle
            ex = rf["exact"]
            line["refined"]["exact"] = {"img_s": _r(ex.get("images_per_sec")), "rescored_exact": ex.get("rescored_exact"),
                                        "fpr95_images_vs_fp32_arm_max_set": ex.get("fpr95_images_vs_fp32_arm_max_set")}
    ro = d.get("roofline")
    if ro:
        line["roofline"] = {k: _r(ro.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                        "avg_launch_us", "flop_per_launch", "sustained_sclk_mhz")}
        line["roofline"]["kernel"] = str(ro.get("kernel"))[:64]
        line["roofline"]["traffic_source"] = "live rocprofv3 --pmc" if str(ro.get("traffic_source", "")).startswith("measured in this run") \
            else ("committed profiles/" if ro.get("traffic") else None)
    hb = d.get("roofline_hbm_kernels")
    if hb:
        line["roofline_hbm"] = {k: {"gbs": _r(v.get("achieved"), 4), "frac": _r(v.get("frac"), 3), "us": _r(v.get("avg_launch_us"), 4)}
                                for k, v in hb.items()}
    if d.get("kernel_ms_per_step"):
        line["kernel_ms_per_step"] = {k: _r(v, 4) for k, v in d["kernel_ms_per_step"].items() if v}
    cb = d.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "value_hoisted", "unit", "cores", "kind",
                                                            "parity_max_abs_dscore_vs_native", "error") if k in cb}
        if cb.get("sample"):
            line["cpu_baseline"]["sample"] = str(cb["sample"])[:96]
    if d.get("sustained_images_per_sec"):
        line["sustained"] = {"img_s": _r(d["sustained_images_per_sec"]), "sclk_mhz": _r(_get(d, "sustained", "sclk_mhz_mean"), 4),
                             "power_w": _r(_get(d, "sustained", "power_w_mean"), 4)}
    for key in ("arms", "configs"):
        if d.get(key):
            line[key] = {n: ({"img_s": _r(a.get("images_per_sec")), "frac": _r(a.get("frac"), 3)} if "error" not in a
                             else {"error": str(a["error"])[:60]}) for n, a in d[key].items()}
    if d.get("ingest"):
        line["ingest"] = {n: _r(v.get("images_per_sec")) if "images_per_sec" in v else str(v.get("error", v.get("skipped")))[:60]
                          for n, v in d["ingest"].items()}
    p = d.get("parity")
    if p and "error" in p:
        line["parity"] = {"error": str(p["error"])[:160]}
    elif p:
        sp = {"vs": p.get("vs"), "meets_1e-4": p.get("meets_1e-4"), "max_set": {}}
        for regime in ("fp16_exact_weights", "fp32_valued_weights"):
            src = _get(p, regime, "vs_hf") or _get(p, regime, "vs_fp32_arm")
            if src:
                sp["max_set"][regime] = {arm: [_r(_get(v, "max_set", "d_auroc"), 3), _get(v, "max_set", "d_fpr95_images")]
                                         for arm, v in src.items()}
        sp["max_set_is"] = "[max over OOD sets |dAUROC|, FPR95 images moved]"
        op = p.get("operating_point_auroc_0.9")
        if op:
            sp["op_auroc_0.9"] = {arm: [_r(v.get("d_auroc"), 3), v.get("d_fpr95_images")] for arm, v in op.get("arms", {}).items()}
        line["parity"] = sp
    line["seconds"] = _r(sum((d.get("leg_seconds") or {}).values()), 4)
    line["detail"] = d.get("detail_file", "bench_detail.json")
    return line


def fit_line(short):
    """The line as text, never above LINE_LIMIT: should a leg ever produce more than the worst case the CPU test builds, whole
    optional sections are dropped (least important first; the detail file has them) rather than losing the run's measurement."""
    short = dict(short)
    for k in (None, "kernel_ms_per_step", "ingest", "configs", "sustained", "arms", "roofline_hbm", "parity", "refined", "collective", "per_rank"):
        if k is not None:
            if k not in short:
                continue
            short.pop(k)
            short["dropped_for_length"] = short.get("dropped_for_length", []) + [k]
        out = json.dumps(short, separators=(",", ":"))
        if len(out) <= LINE_LIMIT:
            return out
    return json.dumps({k: short.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                                 "scaling", "vs_baseline", "dtype", "data", "roofline", "cpu_baseline", "value_refined",
                                                 "frac_refined", "value_fp16x2", "frac_fp16x2", "detail")},
                      separators=(",", ":"))[:LINE_LIMIT]


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: become N ranks."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="images per GPU per step")
    ap.add_argument("--prompts", type=int, default=1000, help="K: size of the concept bank")
    ap.add_argument("--ckpt", default="ViT-B/16")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["bf16", "fp16", "fp16x2", "fp32"],
                    help="fp16x2: the whole step through the split-activation arm (hi + lo fp16 operands; FLOP counted on the LOGICAL problem)")
    ap.add_argument("--weights-regime", default="fp16-exact", choices=["fp16-exact", "fp32"],
                    help="seeded weights rounded to fp16 values (the reference's checkpoints were trained and released in "
                         "fp16: one fp16 operand per weight is lossless) or as drawn (fp32-valued: the 16-bit arms then run "
                         "the split-weight GEMMs unless --weight-operands single)")
    ap.add_argument("--weight-operands", default="auto", choices=["auto", "single", "split"])
    ap.add_argument("--quick", action="store_true", help="the timed region and its roofline only: every side leg off")
    ap.add_argument("--cpu-seconds", type=float, default=120.0,
                    help="hard cap on the CPU baseline's timed part (it stops after 256 images); 0 disables it")
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--sustain-seconds", type=float, default=5.0, help="0 disables the sustained-throughput leg")
    ap.add_argument("--ingest", default="host-u8,host-raw",
                    help="comma list of ingest legs reported next to the device-resident number (N = 1): host-u8 (pinned "
                         "224x224 uint8 crops -> copy stream -> mcm_score_u8), host-raw (variable-size decoded images -> one "
                         "packed copy -> resize/crop on the device -> mcm_score_u8), host-jpeg (an image folder of JPEG files "
                         "through the CLI's loader; not in the default: host-core bound); 'none' skips them")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a captured hipGraph (mcm_amd.engine.GraphedScorer) instead of launching its "
                         "~110 kernels: what a small-batch caller would do (launch-bound below batch ~64); per-kernel events off")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc child passes); the committed "
                         "profiles/*_traffic.json is reported instead")
    ap.add_argument("--no-arms", action="store_true", help="skip the 3-step runs of the other precision arms (N = 1)")
    ap.add_argument("--no-configs", action="store_true", help="skip the 3-step runs of BASELINE configs 2 and 4 (N = 1)")
    ap.add_argument("--no-refined", action="store_true", help="skip the config-3-sized scored-and-refined pass (value_refined)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the process group even for one rank, so that the score all-gather really runs "
                         "(torchrun --nproc-per-node 1: RCCL on the one device)")
    ap.add_argument("--no-drift", action="store_true", help="skip the AUROC/FPR95 parity leg (N = 1 only)")
    ap.add_argument("--no-hf", action="store_true", help="parity leg without the HF-on-device reference scorer")
    ap.add_argument("--parity-regimes", default="fp16-exact",
                    help="comma list of weight regimes of the parity leg: fp16-exact (with HF on the device), fp32 (vs the "
                         "exact-fp32 arm only; tests/test_gpu_headline_parity.py asserts it on every round-end run)")
    ap.add_argument("--drift-n", type=int, nargs=2, default=[50000, 10000], metavar=("N_ID", "N_OOD"))
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP events")
    ap.add_argument("--harness", default="",
                    help="A/B hooks of libmcm_hip_harness.so as k=v[,k=v]: " + ", ".join(sorted(HARNESS_KEYS)) +
                         " (EXPERIMENTS.md); the line is then labelled and is not the shipped policy")
    ap.add_argument("--idle-ms", type=float, default=-1.0,
                    help="measurement hook (EXPERIMENTS.md, the energy reading of the step): >= 0 = synchronise after every "
                         "timed step and leave the device idle for this long; the line is then NOT a throughput figure")
    ap.add_argument("--profile-every", type=int, default=4,
                    help="bracket every kernel of every N-th timed step with HIP events (each pair costs "
                         "~3 us of stream serialisation: all steps = -2.3 %% throughput, every 4th = -0.6 %%)")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes")
    args = ap.parse_args(argv)
    args.harness_kv = {}
    for kv in filter(None, args.harness.split(",")):
        k, _, v = kv.partition("=")
        if k not in HARNESS_KEYS:
            ap.error(f"unknown --harness key {k!r}")
        args.harness_kv[k] = int(v)
    if args.quick:
        args.cpu_seconds = args.sustain_seconds = 0.0
        args.ingest = "none"
        args.no_live_traffic = args.no_arms = args.no_configs = args.no_refined = args.no_drift = True
    return args


def hbm_kernels(prof, n_prof, geo, B, precision):
    """The HBM-bound kernels of the step against the HBM roof (north_star: "rocprof HBM GB/s ... against peak"):
    ALGORITHMIC bytes of one step ÷ HIP-event time of that family in one step.  out-proj is the one GEMM shape that is
    HBM-bound (N = K = width: its fp32 read-modify-write of the residual outweighs its FLOP at the machine balance): X in,
    W in, residual read + written in fp32."""
    M, D, L, es = B * geo.v_tokens, geo.v_width, geo.v_layers, 2 if precision in ("fp16", "bf16") else 4   # (fp16x2: hi + lo = 4 bytes)
    ln_bytes = ((2 * (L - 1) + 1) * M * D * (4 + es)      # layer_norm1/2 of the full layers + the last layer_norm1
                + B * D * (4 + es)                          # the last layer's layer_norm2: CLS rows only
                + M * D * 4)                                # the fused pre_layrnorm pass also rewrites x in fp32
    at_bytes = (L - 1) * M * 4 * D * es + (M * 2 * D * es + 2 * B * D * es)   # qkv in + out; last layer: K, V + CLS
    op_bytes = (L - 1) * (M * D * es + D * D * es + 2 * M * D * 4)              # full layers; the last one is CLS-only
    hbm = {}
    for name, key, nbytes in (("layernorm", "layernorm", ln_bytes), ("attention", "attention", at_bytes),
                              ("outproj", "gemm_outproj", op_bytes)):
        p = prof.get(key)
        if not p or not p.get("ms"):
            continue
        ms = p["ms"] / n_prof
        gbs = nbytes / (ms * 1e-3) / 1e9
        hbm[name] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                     "frac_of_measured_copy_rate": gbs / 6290.0, "algorithmic_bytes_per_step": nbytes, "ms_per_step": ms,
                     "avg_launch_us": 1e3 * p["ms"] / p["launches"] if p.get("launches") else None}
    return hbm


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        respawn_under_torchrun(args.gpus)

    import torch

    from mcm_amd import dist as mdist
    from mcm_amd.config import geometry
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.synth import make_token_ids
    from mcm_amd.weights import synth_state_dict

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    ws_env = int(os.environ.get("WORLD_SIZE", "1"))
    if ws_env != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws_env}; launch with "
                         f"`python bench.py --gpus N` or torch.distributed.run --nproc-per-node N")
    ndev = torch.cuda.device_count()
    shared = ws_env > ndev  # more ranks than devices: logic check only
    rank, ws, local = mdist.init_from_env(backend="gloo" if shared else None, force=args.force_collective)
    coll = mdist.group_active()  # a process group exists (N > 1, or --force-collective): the all-gather runs
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    legs, t_leg = {}, time.perf_counter()  # wall clock per leg of this run -> line["leg_seconds"]

    def leg_done(name):
        nonlocal t_leg
        now = time.perf_counter()
        legs[name] = round(legs.get(name, 0.0) + now - t_leg, 2)
        t_leg = now

    geo = geometry(args.ckpt)
    sd = synth_state_dict(geo, 0, args.weights_regime)
    K, B = args.prompts, args.batch
    ids, mask = make_token_ids(K, seed=2)
    net = NativeCLIP(geo, sd, device=local, precision=args.precision, max_batch=B, max_prompt_tokens=max(K * ids.shape[1], 77),
                     weight_operands=args.weight_operands, harness=bool(args.harness_kv))
    late = {k: v for k, v in args.harness_kv.items() if k in ("qkv_chunks", "gemm_dbg")}  # after the text tower ran
    for k, v in args.harness_kv.items():
        if k not in late and getattr(net._lib, HARNESS_KEYS[k])(v) != 0:
            raise SystemExit(f"--harness {k}={v} refused")
    txt = net.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), normalize=True)
    for k, v in late.items():
        if getattr(net._lib, HARNESS_KEYS[k])(v) != 0:
            raise SystemExit(f"--harness {k}={v} refused")

    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    nbuf = min(4, max(1, args.steps))
    bufs = [torch.randn((B, 3, geo.image_size, geo.image_size), generator=gen, device=dev) for _ in range(nbuf)]
    scores = torch.empty((args.steps, B), device=dev)

    def barrier():
        if coll:
            torch.distributed.barrier()

    graphs = None
    if args.graph:  # one graph per input buffer (a graph replays the pointers it captured): no copy in the timed loop
        from mcm_amd.engine import GraphedScorer

        args.no_profile = True
        graphs = [GraphedScorer(net, B, txt, 1.0, "MCM", input=b) for b in bufs]

    def step(i, out):
        if graphs is None:
            net.score_images(bufs[i % nbuf], txt, 1.0, "MCM", out=out)
        else:
            g = graphs[i % nbuf]
            g.graph.replay()
            out.copy_(g.output, non_blocking=True)

    for i in range(args.warmup):
        if not args.no_profile and i == args.warmup - 1:
            net.profile(True)  # creates the event pool outside the timed region
        step(i, scores[0])
    if coll:  # warm the collective too (RCCL builds its rings on first use)
        mdist.all_gather_scores(scores[0], ws * B)
    torch.cuda.synchronize()
    if not args.no_profile:
        net.profile_read()  # drop the warm-up samples
        net.profile(False)
    pe = max(1, args.profile_every)
    n_prof = 0 if args.no_profile else len(range(0, args.steps, pe))
    leg_done("setup_and_warmup")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if n_prof:
            net.profile(i % pe == 0)  # a host-side flag: events are recorded on profiled steps only
        step(i, scores[i])
        if args.idle_ms >= 0:  # measurement hook: an idle device between steps (kernel times come from the HIP events)
            torch.cuda.synchronize()
            time.sleep(args.idle_ms * 1e-3)
    t_own = t_ag = None
    if coll:  # the path's only exchange: per-dataset all-gather of the score shards
        torch.cuda.synchronize()
        t_own = time.perf_counter() - t0          # this rank's own steps (no collective yet)
        full = mdist.all_gather_scores(scores.reshape(-1), ws * args.steps * B)
        assert full.numel() == ws * args.steps * B
        torch.cuda.synchronize()
        t_ag = time.perf_counter() - t0 - t_own   # the all-gather as THIS rank saw it: its own cost + waiting for the slowest rank
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if coll:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    prof = None
    if not args.no_profile:
        prof = net.profile_read()
        net.profile(False)
    assert torch.isfinite(scores).all()
    nb_cpu = min(args.cpu_batch, B)
    first_scores = scores[0][:nb_cpu].clone()  # step 0 scored bufs[0]; the sustained leg overwrites scores[]
    leg_done("timed_steps")

    side = rank == 0 and ws == 1   # the side legs: one GPU, rank 0
    from tools import bench_legs as bl

    sustained = None
    if args.sustain_seconds > 0:  # every rank runs it (the chip-level power state is what is being measured)
        n_sus = max(args.steps, int(args.sustain_seconds / (dt / args.steps)) + 1)
        # EVERY rank samples ITS OWN GPU (sysfs / rocm-smi -d): on a chassis where eight 1.35-kW parts share a power budget a
        # sub-linear curve must be attributable to clocks from the line alone
        sampler = bl.SmiSampler(period=0.5 if ws == 1 else 1.0, device=local)
        sampler.start()
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_sus):
            step(i, scores[i % args.steps])
        torch.cuda.synchronize()
        dts_own = time.perf_counter() - t1
        barrier()
        dts = time.perf_counter() - t1
        sustained = {"steps": n_sus, "seconds": dts, "images_per_sec": ws * n_sus * B / dts}
        mine = sampler.stop()
        sustained.update(mine)
        sustained["own_images_per_sec"] = n_sus * B / dts_own
        leg_done("sustained")

    per_rank = None
    if coll:  # one record per rank, gathered outside the timed region: what each GPU did on its own, at what clock and power
        rec = {"rank": rank, "device": local, "img_s": args.steps * B / t_own, "ms_own_steps": 1e3 * t_own, "ms_allgather": 1e3 * t_ag,
               "kernel_faults": net.kernel_faults}
        if sustained:
            rec.update(sustained_img_s=sustained.get("own_images_per_sec"), sclk_mhz=sustained.get("sclk_mhz_mean"),
                       power_w=sustained.get("power_w_mean"), smi=sustained.get("source"))
        per_rank = [None] * ws
        torch.distributed.all_gather_object(per_rank, rec)

    ingest = None
    if side and args.ingest != "none":
        ingest = {}
        for leg in args.ingest.split(","):  # a side leg that fails says so in the line; it never costs the headline number
            try:
                ingest.update(bl.ingest_legs(net, txt, B, max(6, min(args.steps, 12)), {leg}))
            except Exception as e:
                ingest[leg.replace("-", "_")] = {"error": f"{type(e).__name__}: {e}"[:400]}
                torch.cuda.synchronize()
        leg_done("ingest")

    line = None
    if rank == 0:
        value = ws * args.steps * B / dt
        nominal = geo.vision_flops_per_image() / 1e9 + 2e-9 * geo.proj_dim * K
        line = {
            "metric": "images/sec MCM-scored (CLIP-B/16, 1000 prompts)",
            "value": value, "unit": "images/sec", "n_gpus": ws, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"MCM scoring, CLIP-{args.ckpt} ({geo.v_layers}L vision tower, random-init weights), "
                                   f"K={K} prompts pre-encoded, batch {B}/GPU, fp32 NCHW pixels resident in HBM -> [B] scores",
                       "batch_per_gpu": B, "prompts": K, "parallelism": f"image-sharded x{ws}"},
            "gflop_per_image": nominal,
            "weights": {"regime": args.weights_regime, "gemm_weight_elements_not_operand_numbers": net.weights_inexact,
                        "split_weight_gemms": net.split_weights},
        }
        if args.harness_kv:
            line["harness"] = f"libmcm_hip_harness.so with {args.harness} forced: an A/B run, not the shipped policy"
            if args.harness_kv.get("ln_cluster"):   # the LNC arm's own counters: waits that gave up, segments left to the clean-up launch
                import ctypes
                for key, fn in (("lnc_timeouts", "mcm_debug_ln_tail_timeouts"), ("lnc_deferred_segments", "mcm_debug_ln_cluster_deferred")):
                    n = ctypes.c_uint64(0)
                    if getattr(net._lib, fn)(net._h, ctypes.byref(n)) == 0:
                        line[key] = int(n.value)
        if args.graph:
            line["hip_graph"] = "every step is one replay of a captured hipGraph (mcm_amd.engine.GraphedScorer)"
        if args.idle_ms >= 0:
            line["idle_ms_between_steps"] = args.idle_ms
            line["note"] = "measurement run with an idle device between steps: `value` is not a throughput figure"
        if coll:
            line["collective"] = ("gloo: %d ranks share %d device(s), RCCL refuses duplicate devices — logic "
                                  "check, not a scaling number" % (ws, ndev)) if shared else \
                "%s all_gather_into_tensor of the score shards (device tensors, no host bounce), inside the timed region" % \
                ("nccl (RCCL)" if torch.distributed.get_backend() == "nccl" else torch.distributed.get_backend())
        line["kernel_faults"] = net.kernel_faults   # bounded waits of the persistent kernels that ran out (0 in a correct run)
        if per_rank:
            line["per_rank"] = per_rank
            line["allgather_ms"] = max(r["ms_allgather"] for r in per_rank)   # incl. waiting for the slowest rank
            line["allgather_ms_min"] = min(r["ms_allgather"] for r in per_rank)  # ~ the collective's own cost (the slowest rank waits for nobody)
        if ingest:
            line["ingest"] = ingest
        if sustained:
            line["sustained_images_per_sec"] = sustained.pop("images_per_sec")
            line["sustained"] = sustained
        if prof:
            g = prof["gemm"]
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] else None
            peak = MFMA_PEAK_TFLOPS[args.precision]
            line["roofline"] = {
                "bound": "mfma", "kernel": "GEMM family (gemm_pp + gemm_p256 + tile kernels: every GEMM launch of a step)",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak if ach else None, "traffic": None,
                "traffic_unit": "L2<->fabric bytes per launch (L2 misses + write-backs, Infinity-Cache hits included: an "
                                "upper bound on HBM bytes) = (2*FETCH_SIZE+WRITE_SIZE)*1024, rocprofv3 PMC",
                "avg_launch_us": 1e3 * g["ms"] / g["launches"] if g["launches"] else None,
                "flop_per_launch": g["flops"] / g["launches"] if g["launches"] else None,
            }
            if side and B == 512:
                tr, src, shapes = bl.pmc_traffic("gemm")
                line["roofline"].update(traffic=tr, traffic_source=src, traffic_per_shape=shapes)
            if sustained and "sclk_mhz_mean" in sustained:
                # the part runs this kernel family at its package power limit: the clock it sustains, not the
                # 2.4 GHz the 2.5 PF/s peak assumes, is what `frac` is achieved at
                line["roofline"]["sustained_sclk_mhz"] = sustained["sclk_mhz_mean"]
                line["roofline"]["sustained_power_w"] = sustained["power_w_mean"]
                line["roofline"]["frac_of_peak_at_sustained_clock"] = \
                    (ach / (peak * sustained["sclk_mhz_mean"] / 2400.0)) if ach else None
            line["roofline_hbm_kernels"] = hbm_kernels(prof, n_prof, geo, B, args.precision)
            fam = {k: v for k, v in prof.items() if not k.startswith("gemm_")}   # gemm_<shape> entries are subsets of "gemm"
            tot = sum(v["ms"] for v in fam.values())
            executed = sum(v["flops"] for v in fam.values()) / n_prof / B / 1e9
            line["kernel_ms_per_step"] = {k: round(v["ms"] / n_prof, 4) for k, v in prof.items()}
            line["profiled_steps"] = n_prof
            line["kernel_time_frac"] = {k: round(v["ms"] / tot, 4) for k, v in fam.items() if tot}
            # effective: nominal tower FLOP per image over wall time.  The last layer runs its MLP / out-proj
            # for the CLS row only (identical results), so the FLOP actually executed are lower:
            line["end_to_end_mfma_frac_effective"] = value * nominal / 1e3 / ws / peak
            line["gflop_per_image_executed"] = executed
            line["end_to_end_mfma_frac_executed"] = value * executed / 1e3 / ws / peak
        leg_done("line")
        if side and args.cpu_seconds > 0:
            pxs = [b[:nb_cpu].cpu() for b in bufs]  # the batches the native run scored (bufs[0] first)
            native = first_scores.cpu().numpy() if args.steps >= 1 else None
            try:
                line["cpu_baseline"] = bl.cpu_baseline(geo, sd, ids, mask, K, pxs, args.cpu_seconds, native)
            except Exception as e:
                line["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:400]}
            leg_done("cpu_baseline")
    px0 = bufs[0]
    del bufs, scores
    if side and not args.no_refined and args.precision in ("fp16", "bf16"):   # (fp32 / fp16x2 runs ARE exact-grade arms)
        # throughput AT PARITY: config 3's sizes scored by this arm and threshold-refined, wall clock, same handle
        try:
            line["refined"] = bl.refined_leg(net, geo, sd, txt, ids, mask, B, K, local)
        except Exception as e:
            line["refined"] = {"error": f"{type(e).__name__}: {e}"[:400]}
            torch.cuda.synchronize()
        leg_done("refined")
    net.close()
    if side and line.get("roofline") and not args.no_live_traffic and not args.no_profile:
        # roofline.traffic witnessed by this very run (the handle is closed: the children have the device to themselves)
        torch.cuda.empty_cache()
        try:
            fam, per, note = bl.live_pmc_traffic(args)
        except Exception as e:
            fam, per, note = None, None, f"{type(e).__name__}: {e}"[:200]
        if fam is not None:
            line["roofline"].update(traffic=fam, traffic_source=note, traffic_per_shape=per)
        else:
            line["roofline"]["traffic_source"] = f"{line['roofline'].get('traffic_source')} (committed record; live PMC pass unavailable: {note})"
        leg_done("live_pmc_traffic")
    if side and not args.no_arms:
        # The other arms on the same workload, witnessed by the same run (3 timed steps each): the exact-fp32 arm (the
        # reference's own precision: fp32 everywhere), bf16 (the dtype BASELINE.md names), and the split-weight fp16 arm on
        # fp32-VALUED seeded weights (what a checkpoint that is not fp16-exact runs, include/mcm.h MCM_WEIGHTS_*)
        arms = {}
        for name, prec, regime, wo in (("fp32", "fp32", args.weights_regime, "auto"),
                                       ("bf16_single_operand", "bf16", args.weights_regime, "single"),  # BASELINE.md's dtype, rounded weights
                                       ("bf16_split", "bf16", args.weights_regime, "auto"),       # fp16 values are not bf16 numbers: split
                                       ("fp16_split_weights", "fp16", "fp32", "split"),
                                       # the split-activation arm (mcm_score_x2: the re-scorer of threshold refinement), as a
                                       # scorer of its own: within one fp32 ulp of the fp32 arm's score
                                       ("fp16x2_split_activations", "fp16", args.weights_regime, "auto")):
            if prec == args.precision and wo == args.weight_operands and regime == args.weights_regime and "x2" not in name:
                continue
            torch.cuda.empty_cache()
            try:
                arms[name] = bl.arm_leg(geo, sd if regime == args.weights_regime else synth_state_dict(geo, 0, regime), prec, wo, B, K,
                                        ids, px0, local, x2="x2" in name)
            except Exception as e:
                arms[name] = {"error": f"{type(e).__name__}: {e}"[:400]}
            arms[name]["weights_regime"] = regime
        line["arms"] = arms
        leg_done("arms")
    del px0
    torch.cuda.empty_cache()
    if side and not args.no_configs:
        # BASELINE configs 4 and 2 on this device (3 timed steps each): ViT-L/14 fp16 batch 256, K = 1000; B/16 K = 100 bf16 / fp16
        line["configs"] = bl.config_legs(local)
        leg_done("configs")
    if rank == 0:
        if side and not args.no_drift and args.precision in ("fp16", "bf16"):
            try:
                line["parity"] = bl.parity_leg(args, K, B, local)
            except Exception as e:
                line["parity"] = {"error": f"{type(e).__name__}: {e}"[:400]}
            leg_done("parity")
        line["leg_seconds"] = legs
        written = []
        for path in dict.fromkeys([args.detail] + ([os.path.join(ROOT, "gpurun_out", "bench_detail.json")]
                                                    if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])):
            try:
                with open(path, "w") as fh:
                    json.dump(line, fh, indent=1)
                written.append(os.path.relpath(path, ROOT))
            except OSError:
                pass
        line["detail_file"] = written[0] if written else None
        print(fit_line(short_line(line)), flush=True)
    if coll:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
