"""eval_ood_detection.py — same CLI as the reference (eval_ood_detection.py:15-51): every flag
and default is kept; additive flags only (`--weights`, `--dtype`, `--templates`, `--tokenizer-dir`,
`--data-dir`, `--host-metrics`, `--synthetic-n`, `--synthetic`).

Drives the MI355X-native hot path: model → loaders → `get_ood_scores_clip` (ID once, then per OOD
set) → AUROC / AUPR / FPR95 → log + CSV.

Data.  `--root-dir` is honoured the way the reference's loaders read it (utils/train_eval_util.py:87-146):
`<root>/<in_dataset>/val/<class>/*` (ImageNet: `<root>/ImageNet/val`) for the ID set and
`<root>/ImageNet_OOD_dataset/{iNaturalist,SUN,Places,dtd/images}` for the OOD sets, read by
mcm_amd.folder.ImageFolderU8 (Pillow decode on the host; Resize + CenterCrop + ToTensor + Normalize on the
GPU, bit-exact with the reference's transform).  When a folder is missing — always the case offline — that
set is the seeded synthetic set of mcm_amd.synth with the reference's dataset size (generated in HBM, image i a
function of (seed, i) only, so per-rank shards see the same pixels), every set with its own seed, and the log,
the console and `<log_directory>/data_sources.json` say so per set.  The fine-grained
ID suites (bird200 / car196 / food101 / pet37) have their own archive formats in the reference
(dataloaders/*.py); only their synthetic form exists here.
Under `torchrun` each rank scores a contiguous shard and the shards are all-gathered (rank 0 reports)."""
import argparse
import logging
import os

import numpy as np

from utils.common import get_num_cls, get_test_labels, setup_seed
from utils.detection_util import (get_and_print_results, get_Mahalanobis_score, get_mean_prec,
                                  get_ood_scores_clip, print_measures)

# dataset sizes (external knowledge, as in the reference's loaders)
N_ID = {"ImageNet": 50000, "ImageNet10": 500, "ImageNet20": 1000, "ImageNet100": 5000,
        "bird200": 5794, "food101": 25250, "pet37": 3669, "car196": 8041}
N_OOD = {"iNaturalist": 10000, "SUN": 10000, "places365": 10000, "dtd": 5640,
         "ImageNet10": 500, "ImageNet20": 1000}


def process_args(argv=None):
    p = argparse.ArgumentParser(description="Evaluates MCM Score for CLIP",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--in_dataset", default="ImageNet", type=str,
                   choices=["ImageNet", "ImageNet10", "ImageNet20", "ImageNet100", "pet37", "food101",
                            "car196", "bird200"], help="in-distribution dataset")
    p.add_argument("--root-dir", default="datasets", type=str, help="root dir of datasets")
    p.add_argument("--name", default="eval_ood", type=str, help="unique ID for the run")
    p.add_argument("--seed", default=5, type=int, help="random seed")
    p.add_argument("--gpu", default=0, type=int, help="the GPU indice to use")
    p.add_argument("-b", "--batch-size", default=512, type=int, help="mini-batch size")
    p.add_argument("--T", type=int, default=1, help="temperature parameter")
    p.add_argument("--model", default="CLIP", type=str, help="model architecture")
    p.add_argument("--CLIP_ckpt", type=str, default="ViT-B/16", choices=["ViT-B/32", "ViT-B/16", "ViT-L/14"],
                   help="which pretrained img encoder to use")
    p.add_argument("--score", default="MCM", type=str,
                   choices=["MCM", "energy", "max-logit", "entropy", "var", "maha"], help="score options")
    # Mahalanobis baseline (--score maha), reference eval_ood_detection.py:38-44
    p.add_argument("--feat_dim", type=int, default=512)
    p.add_argument("--normalize", type=bool, default=False)
    p.add_argument("--generate", type=bool, default=True)
    p.add_argument("--template_dir", type=str, default="img_templates")
    p.add_argument("--subset", default=False, type=bool)
    p.add_argument("--max_count", default=250, type=int)
    # additive
    p.add_argument("--weights", default=None, help="CLIP checkpoint (.safetensors / state_dict); default: seeded synthetic")
    p.add_argument("--dtype", default="fp16", choices=["bf16", "fp16", "fp16x2", "fp32"],
                   help="MFMA operand format of the vision tower (fp16 holds AUROC / AUPR to the fp32 arm at 1e-4 and, with threshold "
                        "refinement, FPR95 too; fp16x2 = every activation as a hi + lo fp16 pair: every score within one fp32 ulp of "
                        "the fp32 arm's at half the fp16 arm's speed, 3.7 x the fp32 arm's; the text tower is always fp32)")
    p.add_argument("--synthetic-weights", default="fp16-exact", choices=["fp16-exact", "fp32"],
                   help="without --weights: seeded parameters rounded to fp16 values, as the reference's checkpoints are "
                        "(default), or as drawn (fp32-valued: the 16-bit arms then run the split-weight GEMMs)")
    p.add_argument("--weight-operands", default="auto", choices=["auto", "single", "split"],
                   help="16-bit modes: one operand per GEMM weight, or W_hi + W_lo (exact for fp32-valued weights, twice "
                        "the GEMM work); auto = split exactly when a weight is not a number of the operand dtype")
    p.add_argument("--refine-threshold", default="auto", choices=["auto", "on", "exact", "off"],
                   help="re-score the images whose 16-bit score lies within a few noise widths of the FPR95 threshold with a "
                        "better arm (a few hundred images per run), so that FPR95 is the exact arithmetic's number and not merely "
                        "within 1 - 8 images of it (mcm_amd/refine.py).  on: --dtype fp16 re-scores with the split-activation arm "
                        "of the SAME handle (mcm_score_x2: within one fp32 ulp of the exact-fp32 arm's score, no second model in "
                        "HBM); --dtype bf16 with an exact-fp32 handle.  exact: additionally the handful of images within a few "
                        "fp32 ulps of the threshold go through an exact-fp32 handle, so the count is that arm's image for image.  "
                        "auto = on for --dtype fp16 / bf16 with an MCM-family --score.  Under torchrun every rank re-scores only "
                        "the window images of its own shard")
    p.add_argument("--host-metrics", action="store_true",
                   help="AUROC/AUPR/FPR95 with sklearn on the host (the reference's route) instead of the device kernels")
    p.add_argument("--synthetic-n", default=None, type=int, help="cap synthetic dataset sizes (smoke runs)")
    p.add_argument("--synthetic", action="store_true", help="never look under --root-dir; seeded synthetic sets only")
    p.add_argument("--templates", default=None, metavar="FILE",
                   help="prompt-ensemble bank (BASELINE config 5): one template per line with {c} or {}, or a .py "
                        "file in the style of the reference's utils/imagenet_templates.py (its 80-template list)")
    p.add_argument("--tokenizer-dir", default=None, help="directory with the checkpoint's vocab.json + merges.txt")
    p.add_argument("--data-dir", default="data", help="the reference's data/ directory (class-name files)")
    p.add_argument("--decoder", default=None, choices=["device", "pillow"],
                   help="how image files under --root-dir are decoded: pillow (default: Pillow in worker processes — the reference's "
                        "decoder, utils/train_eval_util.py:96-146 through torchvision's ImageFolder) or device (opt-in: JPEG entropy "
                        "decoding on host threads, the rest of libjpeg's work + Resize + CenterCrop on the GPU, Pillow for files that "
                        "are not Huffman YCbCr / grayscale JPEGs); same pixels either way")
    p.add_argument("--full-round-batch", action="store_true",
                   help="raise --batch-size to the next batch (within 4 x) at which every vision GEMM fills its last tile round of "
                        "the persistent grid on this device (mcm_amd.config.ClipGeometry.full_round_batches; ViT-B/16 on 256 CUs: "
                        "512 -> 665, +2.6 ... 3.2 %% images/sec; ViT-L/14: 256 -> 318; the chosen batch, or that there is none, is logged).  "
                        "Scores do not depend on the batch they were computed in")
    args = p.parse_args(argv)
    if args.decoder:
        os.environ["MCM_GPU_JPEG"] = "1" if args.decoder == "device" else "0"
    if args.templates:
        from mcm_amd.detection import read_templates

        args.templates = read_templates(args.templates)
    args.n_cls = get_num_cls(args)
    args.log_directory = f"results/{args.in_dataset}/{args.score}/{args.model}_{args.CLIP_ckpt}_T_{args.T}_ID_{args.name}"
    os.makedirs(args.log_directory, exist_ok=True)
    return args


def setup_log(args):
    log = logging.getLogger(__name__)
    log.handlers.clear()
    fmt = logging.Formatter("%(asctime)s : %(message)s")
    for h in (logging.FileHandler(os.path.join(args.log_directory, "ood_eval_info.log"), mode="w"),
              logging.StreamHandler()):
        h.setFormatter(fmt)
        log.addHandler(h)
    log.setLevel(logging.DEBUG)
    return log


# where the reference's loaders look (utils/train_eval_util.py:96-146), relative to --root-dir
OOD_DIRS = {"iNaturalist": ("ImageNet_OOD_dataset", "iNaturalist"), "SUN": ("ImageNet_OOD_dataset", "SUN"),
            "places365": ("ImageNet_OOD_dataset", "Places"), "dtd": ("ImageNet_OOD_dataset", "dtd", "images"),
            "ImageNet10": ("ImageNet10", "train"), "ImageNet20": ("ImageNet20", "val")}
SEEDS = {"id": 1, "train": 7, "iNaturalist": 11, "SUN": 12, "places365": 13, "dtd": 14, "ImageNet10": 15,
         "ImageNet20": 16}


def _loader(args, net, what, n, ood, sources):
    """The loader for one set: the real folder when it exists under --root-dir, else the seeded synthetic
    set (recorded in `sources`, which ends up in the log and in data_sources.json)."""
    from mcm_amd.synth import DevicePatternLoader

    if what in ("id", "train"):
        sub = (args.in_dataset, "val" if what == "id" else "train")
    else:
        sub = OOD_DIRS[what]
    path = os.path.join(args.root_dir, *sub)
    if not args.synthetic and os.path.isdir(path) and args.in_dataset.startswith("ImageNet"):
        from mcm_amd.folder import ImageFolderU8

        from mcm_amd.folder import FolderIndex

        index = FolderIndex(path)
        if what == "train" and args.subset and args.in_dataset == "ImageNet":
            # reference utils/train_eval_util.py:54-64: the first max_count images of every class — and, like
            # there, only for ImageNet (its ImageNet10/20/100 branch ignores `subset`)
            index = index.first_per_class(args.max_count)
        loader = ImageFolderU8(index, net, args.batch_size)
        sources[what] = {"kind": "folder", "path": path, "n": len(loader.dataset)}
        return loader
    if args.synthetic_n:
        n = min(n, args.synthetic_n)
    seed = SEEDS[what]
    sources[what] = {"kind": "synthetic", "n": n, "seed": seed, "why": f"{path} not found"}
    return DevicePatternLoader(n, net.geo.image_size, args.n_cls, args.batch_size, net.device, ood=ood, seed=seed)


def _note_pillow_files(net, sources, what, log):
    """How many files of the set just scored were decoded by Pillow inside the JPEG pipe (files that are not Huffman
    YCbCr / grayscale JPEGs, or that show any irregularity): recorded with the set's data source."""
    total = sum(p.fallback_images for p in net.__dict__.get("_jpeg_pipes", {}).values())
    seen = net.__dict__.get("_pillow_files_seen", 0)
    if what in sources and sources[what].get("kind") == "folder":
        sources[what]["decoded_by_pillow"] = total - seen
        if total - seen:
            log.debug(f"{what}: {total - seen} file(s) decoded by Pillow (not taken by the device JPEG route)")
    net.__dict__["_pillow_files_seen"] = total


def main(argv=None):
    import json

    import torch

    from mcm_amd import dist as mdist
    from mcm_amd.config import HUB_IDS
    from mcm_amd.detection import maha_file_name
    from mcm_amd.engine import build_model

    args = process_args(argv)
    setup_seed(args.seed)
    assert torch.cuda.is_available()
    ndev = torch.cuda.device_count()
    ws_env = int(os.environ.get("WORLD_SIZE", "1"))
    # more ranks than devices (a 1-GPU box running the 2-rank logic check): the ranks share devices and the
    # score all-gather goes over gloo with a host bounce, because RCCL refuses two ranks on one device
    # (LOCAL_WORLD_SIZE: the ranks of THIS node — a multi-node torchrun has WORLD_SIZE > device_count() with one GPU per rank)
    ws_local = int(os.environ.get("LOCAL_WORLD_SIZE", ws_env))
    rank, ws, local = mdist.init_from_env(backend="gloo" if ws_local > ndev else None)
    log = setup_log(args)
    dev = (local % ndev) if ws > 1 else args.gpu
    torch.cuda.set_device(dev)
    if args.full_round_batch:   # (resolved here, not in process_args: the tile rounds depend on THIS device's CU count)
        from mcm_amd.config import geometry

        cus = torch.cuda.get_device_properties(dev).multi_processor_count // 8 * 8   # the persistent GEMMs' grid
        better = geometry(args.CLIP_ckpt).full_round_batches(args.batch_size, 4 * args.batch_size, cus=cus) if cus >= 8 else []
        if better:
            log.debug(f"--full-round-batch: batch {args.batch_size} -> {better[0]} (every vision GEMM of {args.CLIP_ckpt} fills its last "
                      f"tile round on {cus} CUs; the activation workspace grows with it)")
            args.batch_size = better[0]
        else:
            log.debug(f"--full-round-batch: no batch in [{args.batch_size}, {4 * args.batch_size}] fills every GEMM's last tile round "
                      f"for {args.CLIP_ckpt} on {cus} CUs; keeping batch {args.batch_size}")
    # the split-activation workspace of an fp16 handle (include/mcm.h mcm_config.x2_max_batch): the whole run goes
    # through that arm with --dtype fp16x2 (full batch: twice the activation bytes); threshold refinement re-scores a few hundred
    # images, for which half the batch costs nothing extra; otherwise the workspace is not allocated at all
    will_refine = args.refine_threshold != "off" and args.score != "maha"
    x2_batch = 0 if args.dtype == "fp16x2" else (max(1, args.batch_size // 2) if (args.dtype == "fp16" and will_refine) else -1)
    net = build_model(args.CLIP_ckpt, weights=args.weights, device=dev, precision=args.dtype,
                      max_batch=args.batch_size, synthetic_regime=args.synthetic_weights,
                      weight_operands=args.weight_operands, x2_max_batch=x2_batch)
    net.eval()
    if args.dtype != "fp32":
        log.debug(f"vision GEMM weights: {net.weights_inexact} element(s) are not {args.dtype} numbers -> "
                  f"{'split-weight GEMMs (W_hi + W_lo, exact for the weight operand)' if net.split_weights else 'one operand per weight'}")
        if net.weights_inexact and not net.split_weights:
            log.debug("NOTE: --weight-operands single ROUNDS those weights: measured 1.5 - 2e-4 in AUROC against the fp32 "
                      "reference in this regime (DESIGN.md section 2.1)")
    if args.dtype == "bf16":
        log.debug("NOTE: bf16 operands do not hold the reference's AUROC / FPR95 to 1e-4 (measured 1e-4 ... 1e-3 against "
                  "HF fp32, DESIGN.md section 2.1); --dtype fp16 (the default) does")
    args.ckpt = HUB_IDS[args.CLIP_ckpt]  # reference utils/train_eval_util.py:19-22 (ckpt_mapping)
    if args.in_dataset == "ImageNet10":
        out_datasets = ["ImageNet20"]
    elif args.in_dataset == "ImageNet20":
        out_datasets = ["ImageNet10"]
    else:
        out_datasets = ["iNaturalist", "SUN", "places365", "dtd"]
    sources = {}
    test_loader = _loader(args, net, "id", N_ID[args.in_dataset], False, sources)
    test_labels = get_test_labels(args, test_loader)
    on_dev = not args.host_metrics and args.score != "maha"  # scores stay in HBM; three metrics come back
    if args.score == "maha":  # reference eval_ood_detection.py:72-79
        args.feat_dim = net.geo.proj_dim
        os.makedirs(args.template_dir, exist_ok=True)
        # world_size > 1: the fit (a host-side float64 step over the training features) runs on rank 0 only; the two
        # statistics are broadcast (RCCL) and every rank scores its own shard of each test set (get_Mahalanobis_score)
        if args.generate and rank == 0:
            n_train = min(N_ID[args.in_dataset], args.max_count * args.n_cls) if args.subset else N_ID[args.in_dataset]
            train_loader = _loader(args, net, "train", n_train, False, sources)
            get_mean_prec(args, net, train_loader)
        # like the reference (:77-78) the statistics are always read back from the files get_mean_prec wrote —
        # or that an earlier run wrote, which is what `--generate ""` (argparse's only falsy bool) is for
        stats = {"classwise_mean": torch.empty((args.n_cls, args.feat_dim)), "precision": torch.empty((args.feat_dim, args.feat_dim))}
        missing = None
        if rank == 0:
            for what in ("classwise_mean", "precision"):
                f = os.path.join(args.template_dir, maha_file_name(args, what))
                if not os.path.exists(f):
                    missing = f
                    break
                stats[what] = torch.load(f, map_location="cpu").float().contiguous()
        if ws > 1:  # every rank learns whether rank 0 found the files BEFORE anybody waits in the broadcast
            ok = torch.tensor([0.0 if missing else 1.0])
            mdist.broadcast_tensors([ok], src=0)
            if float(ok[0]) == 0.0 and missing is None:
                missing = "the statistics files (see rank 0's message)"
        if missing:
            raise SystemExit(f"--generate is off and {missing} does not exist: run once with --generate True")
        if ws > 1:
            mdist.broadcast_tensors([stats["classwise_mean"], stats["precision"]], src=0)
        classwise_mean, precision = stats["classwise_mean"], stats["precision"]
        in_score = get_Mahalanobis_score(args, net, test_loader, classwise_mean, precision, in_dist=True)
    else:
        in_score = get_ood_scores_clip(args, net, test_loader, test_labels, in_dist=True, device_out=on_dev)
        _note_pillow_files(net, sources, "id", log)
    net.warn_if_saturated(f"the ID set {args.in_dataset}")
    refiner, net32 = None, None
    refinable = args.score != "maha" and args.dtype in ("fp16", "bf16")  # (fp16x2 / fp32 runs ARE exact-grade arms; with
                                                                       # --host-metrics the scores are host arrays: refined in place too)
    if args.refine_threshold != "off" and refinable:
        from mcm_amd.detection import prompt_bank
        from mcm_amd.refine import Rescorer, ThresholdRefiner

        bank = prompt_bank(args, net, test_labels)  # (the 16-bit handle's text tower is exact fp32: one bank for every arm)
        set_loaders = {"id": test_loader}
        use_x2 = net.x2_max_batch > 0   # fp16 handles: the split-activation arm, same weights, same workspace
        if not use_x2 or args.refine_threshold == "exact":
            # the exact-fp32 arm over the same weights: the only re-scorer of a bf16 run (its batch as large as the window
            # asks for), the inner-window re-scorer of `exact` (a handful of images: a small workspace)
            net32 = build_model(args.CLIP_ckpt, weights=args.weights, device=dev, precision="fp32",
                                max_batch=min(args.batch_size, 256 if not use_x2 else 32), synthetic_regime=args.synthetic_weights)
        first = Rescorer(net.x2_scorer() if use_x2 else net32, bank, set_loaders, args.T, args.score)
        second = Rescorer(net32, bank, set_loaders, args.T, args.score) if (use_x2 and net32 is not None) else None
        refiner = ThresholdRefiner(first, rescore_exact=second)
        refiner.fit_id(in_score if on_dev else torch.from_numpy(in_score))
        how = "the split-activation arm of this handle" if use_x2 else "an exact-fp32 handle"
        log.debug("threshold refinement: 16-bit score noise (max over %d calibration images) %.2e, window +-%.2e around the "
                  "FPR95 threshold, %d ID images re-scored by %s" % (refiner.stats["calibration_images"],
                  refiner.stats["noise_max_abs"], refiner.stats["delta"], refiner.stats["rescored"]["id"], how))
        if second is not None:
            log.debug("threshold refinement, inner window +-%.2e: %d ID images re-scored by the exact-fp32 arm"
                      % (refiner.stats["delta2"], refiner.stats["rescored_exact"]["id"]))
    elif args.refine_threshold in ("on", "exact"):
        raise SystemExit("--refine-threshold on needs a 16-bit --dtype and an MCM-family --score")
    auroc_list, aupr_list, fpr_list = [], [], []
    result = {"in_score": in_score, "out_scores": {}, "rank": rank, "world_size": ws, "sources": sources,
              "log_directory": args.log_directory, "batch_size": args.batch_size}
    for out_dataset in out_datasets:
        log.debug(f"Evaluting OOD dataset {out_dataset}")
        ood_loader = _loader(args, net, out_dataset, N_OOD[out_dataset], True, sources)
        log.debug(f"data source: {sources[out_dataset]}")
        if args.score == "maha":
            out_score = get_Mahalanobis_score(args, net, ood_loader, classwise_mean, precision, in_dist=False)
        else:
            out_score = get_ood_scores_clip(args, net, ood_loader, test_labels, device_out=on_dev)
            _note_pillow_files(net, sources, out_dataset, log)
            if refiner is not None:
                set_loaders[out_dataset] = ood_loader
                refiner.apply(out_dataset, out_score if on_dev else torch.from_numpy(out_score))
                log.debug(f"threshold refinement: {refiner.stats['rescored'][out_dataset]} images of {out_dataset} re-scored"
                          + (f", {refiner.stats['rescored_exact'][out_dataset]} of them by the exact-fp32 arm" if second is not None else ""))
        result["out_scores"][out_dataset] = out_score
        net.warn_if_saturated(out_dataset)
        if rank == 0:
            get_and_print_results(args, log, in_score, out_score, auroc_list, aupr_list, fpr_list, net=net)
    result["measures"] = {d: (a, p, f) for d, a, p, f in zip(out_datasets, auroc_list, aupr_list, fpr_list)}
    if rank == 0:
        log.debug("\n\nMean Test Results")
        print_measures(log, np.mean(auroc_list), np.mean(aupr_list), np.mean(fpr_list), method_name=args.score)
        import pandas as pd

        rows = {d: [100 * f, 100 * a, 100 * p] for d, f, a, p in zip(out_datasets, fpr_list, auroc_list, aupr_list)}
        rows["AVG"] = [100 * np.mean(fpr_list), 100 * np.mean(auroc_list), 100 * np.mean(aupr_list)]
        pd.DataFrame.from_dict(rows, orient="index", columns=["FPR95", "AUROC", "AUPR"]).round(2).to_csv(
            os.path.join(args.log_directory, f"{args.name}.csv"))
        synthetic = [k for k, v in sources.items() if v["kind"] == "synthetic"]
        if synthetic or not args.weights:
            log.debug(f"NOTE: synthetic inputs ({', '.join(synthetic) or 'none'}; weights: "
                      f"{'checkpoint' if args.weights else 'seeded synthetic'}) — these numbers are arithmetic "
                      "checks, not the paper's accuracy")
        with open(os.path.join(args.log_directory, "data_sources.json"), "w") as f:
            json.dump({"weights": args.weights or "seeded synthetic", "sets": sources}, f, indent=1)
    if refiner is not None:
        result["refine"] = dict(refiner.stats, rescorer="x2" if use_x2 else "fp32", rescored_by_this_rank=first.scored_here,
                                rank=rank, world_size=ws)
        with open(os.path.join(args.log_directory, f"refine_rank{rank}.json"), "w") as f:  # (every rank: who re-scored what)
            json.dump(result["refine"], f, indent=1)
        if net32 is not None:
            net32.close()
    # programmatic callers (tests) get the scores back; as device tensors they outlive the handle (torch owns them)
    net.close()
    if ws > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
