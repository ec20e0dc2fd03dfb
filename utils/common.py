"""Concept banks and CLI helpers (reference utils/common.py:9-87), re-stated.

ImageNet-10 / ImageNet-20 banks are the reference's hand-written names (utils/common.py:36-60), kept as
(wnid, name) data and ordered by wnid as the reference orders them.  ImageNet-1k / ImageNet-100 read the
class-name files of the reference's `data/` directory (dataset artefacts, not part of this repo; point
`--data-dir` at them).  When a file is missing the bank is K placeholder names AND a RuntimeWarning: with
synthetic weights only K matters, with a real checkpoint the scores would be meaningless."""
import json
import os
import random
import warnings

import numpy as np

N_CLS = {"ImageNet": 1000, "ImageNet10": 10, "ImageNet20": 20, "ImageNet100": 100,
         "bird200": 200, "car196": 196, "food101": 101, "pet37": 37}

# (wnid, concept name) — reference utils/common.py:38-42 and :50-54
IMAGENET10 = (("n01530575", "brambling bird"), ("n01641577", "bull frog"), ("n02107574", "swiss mountain dog"),
              ("n02123597", "Siamese cat"), ("n02389026", "horse"), ("n02422699", "antelope"),
              ("n03095699", "container ship"), ("n03417042", "garbage truck"), ("n04285008", "sports car"),
              ("n04552348", "warplane"))
IMAGENET20 = (("n01630670", "common newt"), ("n01631663", "eft"), ("n01632458", "spotted salamander"),
              ("n01693334", "green lizard"), ("n01697457", "African crocodile"), ("n02114367", "timber wolf"),
              ("n02120079", "Arctic fox"), ("n02132136", "brown bear"), ("n02317335", "starfish"),
              ("n02391049", "zebra"), ("n02782093", "balloon"), ("n02917067", "bullet train"),
              ("n02951358", "canoe"), ("n03773504", "missile"), ("n03785016", "moped"),
              ("n04147183", "sailboat"), ("n04252077", "snowmobile"), ("n04266014", "space shuttle"),
              ("n04310018", "steam locomotive"), ("n04389033", "tank"))


def setup_seed(seed):
    """reference utils/common.py:9-13 (no effect on the eval-only path)."""
    import torch

    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def get_num_cls(args):
    """reference utils/common.py:75-87."""
    return N_CLS[args.in_dataset]


def _placeholders(args, missing):
    from mcm_amd.synth import class_names

    msg = f"concept bank for {args.in_dataset}: {missing} not found; using {N_CLS[args.in_dataset]} placeholder names"
    if getattr(args, "weights", None):
        raise FileNotFoundError(msg + " is not allowed with --weights (pass --data-dir)")
    warnings.warn(msg + " (valid only with synthetic weights)", RuntimeWarning, stacklevel=3)
    return class_names(N_CLS[args.in_dataset])


def get_test_labels(args, loader=None):
    """K concept names for `--in_dataset` (reference utils/common.py:16-73)."""
    ds = args.in_dataset
    if ds == "ImageNet10":
        return [name for _, name in sorted(IMAGENET10)]
    if ds == "ImageNet20":
        return [name for _, name in sorted(IMAGENET20)]
    if ds in ("bird200", "car196", "food101", "pet37"):  # reference dataloaders/*.py expose the names
        names = getattr(getattr(loader, "dataset", None), "class_names_str", None)
        return names if names is not None else _placeholders(args, "loader.dataset.class_names_str")
    data = getattr(args, "data_dir", None) or "data"
    if ds == "ImageNet":
        p = os.path.join(data, "ImageNet", "imagenet_class_clean.npy")
        return list(np.load(p)) if os.path.exists(p) else _placeholders(args, p)
    p = os.path.join(data, ds, "class_list.txt")                       # ImageNet100: wnids -> names
    idx = os.path.join(data, "ImageNet", "imagenet_class_index.json")
    if not (os.path.exists(p) and os.path.exists(idx)):
        return _placeholders(args, f"{p} / {idx}")
    name_of = {wnid: name for wnid, name in json.load(open(idx)).values()}
    return [name_of[line.strip()].replace("_", " ") for line in open(p) if line.strip()]
