"""Concept banks and CLI helpers (reference utils/common.py:9-87), re-stated.

The reference reads class-name files shipped in its `data/` directory; those are dataset
artefacts, not part of this repo, so when a bank file is absent the bank falls back to K
placeholder names of the right size (the hot path only needs K strings)."""
import json
import os
import random

import numpy as np

N_CLS = {"ImageNet": 1000, "ImageNet10": 10, "ImageNet20": 20, "ImageNet100": 100,
         "bird200": 200, "car196": 196, "food101": 101, "pet37": 37}


def setup_seed(seed):
    """reference utils/common.py:9-13 (no effect on the eval-only path)."""
    import torch

    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def get_num_cls(args):
    """reference utils/common.py:75-87."""
    return N_CLS[args.in_dataset]


def get_test_labels(args, loader=None):
    """K concept names for `--in_dataset` (reference utils/common.py:16-73)."""
    from mcm_amd.synth import class_names

    if loader is not None and hasattr(getattr(loader, "dataset", None), "class_names_str"):
        return loader.dataset.class_names_str  # reference dataloaders/*.py
    data = os.path.join(getattr(args, "data_dir", "data"))
    if args.in_dataset == "ImageNet":
        p = os.path.join(data, "ImageNet", "imagenet_class_clean.npy")
        if os.path.exists(p):
            return list(np.load(p))
    elif args.in_dataset.startswith("ImageNet"):
        p = os.path.join(data, args.in_dataset, "class_list.txt")
        idx = os.path.join(data, "ImageNet", "imagenet_class_index.json")
        if os.path.exists(p) and os.path.exists(idx):
            wnid = {v[0]: v[1] for v in json.load(open(idx)).values()}
            return [wnid[line.strip()].replace("_", " ") for line in open(p) if line.strip()]
    return class_names(N_CLS[args.in_dataset])
