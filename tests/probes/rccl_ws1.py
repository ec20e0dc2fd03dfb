"""rccl_ws1.py — every collective of the path through RCCL on the one GPU of the test box (run under
`python -m torch.distributed.run --nproc-per-node 1`; tests/test_gpu_rccl.py drives it).

A 1-rank process group with backend "nccl" (= RCCL on ROCm): get_ood_scores_clip's score all-gather
(`all_gather_into_tensor` on device tensors, no host bounce), the histogram all-reduce, the Mahalanobis statistics
broadcast and the sharded Mahalanobis scorer, each compared with the same computation done without a process group.
Prints one JSON line."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    from mcm_amd import dist as mdist
    from mcm_amd.config import geometry
    from mcm_amd.detection import get_Mahalanobis_score, get_ood_scores_clip
    from mcm_amd.engine import NativeCLIP
    from mcm_amd.synth import DevicePatternLoader, class_names
    from mcm_amd.weights import synth_state_dict

    geo = geometry("tiny")
    dev = torch.device("cuda", 0)
    net = NativeCLIP(geo, synth_state_dict(geo, 0, "fp16-exact"), precision="fp16", max_batch=64, max_prompt_tokens=4096)
    net.synthetic_weights = True
    K, n, bs = 10, 333, 64
    args = types.SimpleNamespace(ckpt="x", model="CLIP", score="MCM", T=1, normalize=False, batch_size=bs)
    labels = class_names(K)
    own = DevicePatternLoader(n, geo.image_size, K, bs, dev, ood=False, seed=3)
    px = torch.cat([b for b, _ in DevicePatternLoader(n, geo.image_size, K, bs, dev, ood=False, seed=3)])
    plain = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(px.cpu(), torch.zeros(n, dtype=torch.long)),
                                        batch_size=bs, shuffle=False)
    edges = np.linspace(-0.2, 0.0, 33).astype(np.float32)
    mu = torch.randn(K, geo.proj_dim, generator=torch.Generator().manual_seed(1))
    a = torch.randn(geo.proj_dim, geo.proj_dim, generator=torch.Generator().manual_seed(2))
    prec = (a @ a.T / geo.proj_dim + torch.eye(geo.proj_dim)).float()

    # without a process group
    assert not mdist.group_active()
    base = {"own": get_ood_scores_clip(args, net, own, labels), "plain": get_ood_scores_clip(args, net, plain, labels),
            "maha": get_Mahalanobis_score(args, net, plain, mu, prec, in_dist=True)}
    base["hist"] = mdist.all_gather_histograms(torch.from_numpy(base["own"]).to(dev), edges, net=net)

    rank, ws, local = mdist.init_from_env(backend="nccl", force=True)
    assert mdist.group_active() and dist.get_backend() == "nccl" and ws == 1
    got = {"own": get_ood_scores_clip(args, net, own, labels), "plain": get_ood_scores_clip(args, net, plain, labels),
           "maha": get_Mahalanobis_score(args, net, plain, mu, prec, in_dist=True)}
    got["hist"] = mdist.all_gather_histograms(torch.from_numpy(got["own"]).to(dev), edges, net=net)
    dev_scores = get_ood_scores_clip(args, net, own, labels, device_out=True)  # stays in HBM through the all-gather
    t = [mu.clone(), prec.clone().to(dev)]
    mdist.broadcast_tensors(t, src=0)
    ok = {k: bool(np.array_equal(base[k], got[k])) for k in base}
    # threshold refinement with the sharded re-scorer: the patches ride an all-reduce (RCCL here) — same scores as without a group
    from mcm_amd.detection import prompt_bank
    from mcm_amd.refine import Rescorer, refine_threshold_scores

    def refined():
        sid = get_ood_scores_clip(args, net, own, labels, device_out=True).clone()
        r = Rescorer(net.x2_scorer(), prompt_bank(args, net, labels), {"id": own, "o": own}, 1.0, "MCM")
        so = sid.flip(0).clone()
        _, _, st = refine_threshold_scores(sid, {"o": so}, r, calib=64)
        return sid.cpu().numpy(), st["rescored_total"], r.scored_here

    with_group = refined()
    ok["device_out"] = bool(dev_scores.is_cuda and np.array_equal(dev_scores.cpu().numpy(), base["own"]))
    ok["broadcast"] = bool(torch.equal(t[0], mu) and torch.equal(t[1].cpu(), prec))
    ok["hist_is_numpy_histogram"] = bool(np.array_equal(got["hist"], np.histogram(base["own"], bins=edges)[0]))
    dist.barrier()
    dist.destroy_process_group()
    without = refined()
    ok["refine_sharded_rescorer"] = bool(np.array_equal(with_group[0], without[0]) and with_group[1:] == without[1:] and with_group[1] > 0)
    net.close()
    print(json.dumps({"backend": "nccl", "world_size": ws, "ok": ok, "all_ok": all(ok.values())}))
    sys.exit(0 if all(ok.values()) else 1)


if __name__ == "__main__":
    main()
