import json, sys
sys.path.insert(0, '/root/repo')
from mcm_amd.parity import measure_drift, CONFIG3_OOD_SETS, HEADLINE_PIXELS
for w in ("fp32", "fp16-exact"):
    d = measure_drift("ViT-B/16", K=100, n_id=5000, batch=512, arms=("fp16", "fp16x2", "fp16+refine", "fp16+refine2"),
                      ood_sets=CONFIG3_OOD_SETS, amp=HEADLINE_PIXELS["amp"], tile=HEADLINE_PIXELS["tile"], weights=w)
    print(w, json.dumps({a: {k: v[k] for k in ("max_abs_dscore", "rms_dscore", "max_set")} for a, v in d["arms"].items()}))
    print(json.dumps(d["refine"]))
