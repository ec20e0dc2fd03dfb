"""End-to-end device chain on synthetic decoded images: uint8 [375,500,3] → resize/crop → uint8 scoring
(ToTensor+Normalize fused) → scores kept in HBM → AUROC/AUPR/FPR95 on the device.  Everything after
the JPEG decoder of the reference's pipeline (utils/train_eval_util.py:27-33 + utils/detection_util.py:
209-265).  Usage: python tools/e2e_probe.py [n_id n_ood batch]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from mcm_amd.engine import build_model  # noqa: E402
from mcm_amd.synth import make_token_ids  # noqa: E402

n_id, n_ood, B = (int(v) for v in (sys.argv[1:4] + ["8192", "4096", "512"][len(sys.argv) - 1:]))
net = build_model("ViT-B/16", max_batch=B, max_prompt_tokens=1000 * 16)
ids, mask = make_token_ids(1000, seed=2)
txt = net.get_text_features(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), normalize=True)
g = torch.Generator(device="cuda").manual_seed(0)
pool = [torch.randint(0, 256, (375, 500, 3), dtype=torch.uint8, device="cuda", generator=g) for _ in range(B)]
shift = [torch.clamp(p.int() + 40, 0, 255).to(torch.uint8) for p in pool]  # a brighter "OOD" set


def run(images, n):
    out = []
    for s in range(0, n, B):
        b = min(B, n - s)
        out.append(net.score_images(net.resize_crop(images[:b]), txt, 1.0, "MCM"))
    return torch.cat(out)


run(pool, B)  # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
s_in = run(pool, n_id)
s_out = run(shift, n_ood)
auroc, aupr, fpr = net.measures(s_in, s_out, negate=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n_id}+{n_ood} images of 375x500 in {dt:.3f} s = {(n_id + n_ood) / dt:,.0f} img/s end to end "
      f"(resize/crop + score + metrics on device); AUROC {auroc:.4f} AUPR {aupr:.4f} FPR95 {fpr:.4f}")

# ---- the PCIe-inclusive rate: batches start in pinned HOST memory (what a DataLoader hands over), are copied
# on a side stream into a double buffer and scored on the main stream.  fp32 NCHW (the reference's loader
# output, 602 112 B per image) and uint8 NHWC crops (150 528 B per image).
copy_stream = torch.cuda.Stream()
for name, shape, dtype in (("fp32 NCHW", (B, 3, 224, 224), torch.float32), ("uint8 NHWC", (B, 224, 224, 3), torch.uint8)):
    host = [torch.empty(shape, dtype=dtype).pin_memory() for _ in range(2)]
    for hbuf in host:
        if dtype == torch.uint8:
            hbuf.random_(0, 256)
        else:
            hbuf.normal_()
    dev = [torch.empty(shape, dtype=dtype, device="cuda") for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    nb = 24
    out = torch.empty((nb, B), device="cuda")

    def loop():
        for i in range(nb):
            k = i & 1
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[k])          # the previous use of this device buffer has been scored
                dev[k].copy_(host[k], non_blocking=True)
                ready[k].record(copy_stream)
            torch.cuda.current_stream().wait_event(ready[k])
            net.score_images(dev[k], txt, 1.0, "MCM", out=out[i])
            done[k].record()

    for e in done:
        e.record()
    loop()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = host[0].numel() * host[0].element_size()
    print(f"host-resident {name} batches (pinned, H2D on a side stream, double-buffered): {nb * B / dt:,.0f} img/s, "
          f"{nb * nbytes / dt / 1e9:.1f} GB/s over PCIe")
net.close()
