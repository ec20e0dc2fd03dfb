"""pmc_limiters.py — which pipe bounds each kernel of the step: rocprofv3 PMC passes over `bench.py --quick`, per kernel family.

VERDICT r4 item 4 asks, for the two kernels below their HBM roof (attention, out-proj): "if both are again 'measured, nothing',
say which counter proves HBM was busy".  This tool collects, in separate `--pmc` passes (SQ has 8 slots, TCC 4; never combined
with a trace domain other than --kernel-trace), the SQ issue / wait split, the VALU, transcendental, MFMA, LDS and VMEM
instruction counts, and the L2's memory-side request counts, queue levels and stalls, and prints per family:

  valu_util     SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE x SIMDs)     share of SIMD-cycles the vector ALU is issuing
  mfma_util     SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs)
  wait / stall  SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY as shares of SQ_WAVE_CYCLES
  ea_rd_latency TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (cycles a read request spends beyond the L2), ea stalls

Usage (on the GPU box, from the repo root; ~3 minutes):  python tools/pmc_limiters.py [out.json]
out-proj and fc2 share an instantiation (gemm_pp_kernel<2, 2>); they alternate inside a layer (out-proj first), which is how
the launches are told apart (dispatch order)."""
import collections
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
     "SQ_ACTIVE_INST_LDS", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
     "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_FMA_F32",
     "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"],
    ["TCC_EA0_RDREQ", "TCC_EA0_RDREQ_LEVEL", "TCC_BUSY", "TCC_CYCLE"],
    ["TCC_EA0_WRREQ", "TCC_EA0_WRREQ_LEVEL", "TCC_EA0_WRREQ_STALL", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL"],
    ["TCC_HIT", "TCC_MISS", "TCC_REQ", "TCC_TAG_STALL"],
]
SIMDS = 1024  # 256 CUs x 4
XCDS = 8      # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked: QKV's mfma_util = 0.58, the known 55 - 60 %)


def family(name):
    m = re.search(r"gemm_\w+_kernel<(\d+), (\d+)", name)
    if m:
        if m.group(1) == "1":
            return None  # exact-fp32 kernels: the text tower's one pass
        return {"0": "gemm_qkv", "1": "gemm_fc1", "2": "gemm_resid", "3": "gemm_patch"}.get(m.group(2))
    for k in ("attn", "layernorm", "score", "pool_project"):
        if k in name:
            return "attention" if k == "attn" else k
    return None


def run_pass(counters, child):
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        r = subprocess.run([rp, "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "p", "--"] + child, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
        dbs = [os.path.join(dp, f) for dp, _d, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
        if r.returncode or not dbs:
            return None, (r.stderr or r.stdout)[-400:]
        db = sqlite3.connect(dbs[0])
        cols = [c[1] for c in db.execute("pragma table_info(counters_collection)")]
        order = "dispatch_id" if "dispatch_id" in cols else "rowid"
        per = collections.defaultdict(lambda: collections.defaultdict(list))   # family -> counter -> [values in dispatch order]
        seen_resid = collections.defaultdict(int)
        rows = db.execute(f"select {order}, kernel_name, counter_name, value from counters_collection order by {order}").fetchall()
        # the text tower's pass and the warm-up step come first; keep everything (means), but split the residual GEMMs by parity
        by_dispatch = collections.OrderedDict()
        for did, name, cname, val in rows:
            by_dispatch.setdefault(did, (name, {}))[1][cname] = by_dispatch.get(did, (name, {}))[1].get(cname, 0.0) + float(val)
        for did, (name, vals) in by_dispatch.items():
            f = family(name)
            if f is None:
                continue
            if f == "gemm_resid":
                big = "gemm_pp" in name or "gemm_p256" in name   # (the CLS-only last layer runs tile kernels: skipped)
                if not big:
                    continue
                f = "gemm_outproj" if seen_resid["n"] % 2 == 0 else "gemm_fc2"
                seen_resid["n"] += 1
            for c, v in vals.items():
                per[f][c].append(v)
        return {f: {c: sum(v) / len(v) for c, v in cs.items()} | {"_launches": len(next(iter(cs.values())))} for f, cs in per.items()}, None


def main():
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--quick", "--steps", "3", "--warmup", "1", "--no-profile"]
    merged = collections.defaultdict(dict)
    notes = []
    for counters in PASSES:
        got, err = run_pass(counters, child)
        if got is None:
            notes.append({"pass": counters, "error": err})
            continue
        for f, cs in got.items():
            merged[f].update(cs)
    out = {"source": "rocprofv3 --kernel-trace --pmc <pass> -- python bench.py --quick --steps 3 (B/16, batch 512, fp16), one pass per "
                     "counter group; means per launch; SQ_* cycle counters are quad-cycles summed over waves (guide: MI355X_MICROARCH.md)",
           "families": {}, "notes": notes}
    for f, c in sorted(merged.items()):
        d = dict(c)
        gui = c.get("GRBM_GUI_ACTIVE")
        if gui:
            gui = gui / XCDS
            d["gpu_cycles"] = gui
            if "SQ_ACTIVE_INST_VALU" in c:
                d["valu_util"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (gui * SIMDS)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                d["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS)
            if "SQ_ACTIVE_INST_LDS" in c:
                d["lds_issue_util"] = c["SQ_ACTIVE_INST_LDS"] * 4 / (gui * SIMDS)
            if "SQ_ACTIVE_INST_VMEM" in c:
                d["vmem_issue_util"] = c["SQ_ACTIVE_INST_VMEM"] * 4 / (gui * SIMDS)
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if k in c:
                    d["share_" + k[3:].lower()] = c[k] / wc
        if c.get("TCC_EA0_RDREQ"):
            d["ea_read_latency_cycles"] = c.get("TCC_EA0_RDREQ_LEVEL", 0.0) / c["TCC_EA0_RDREQ"]
        if c.get("TCC_EA0_WRREQ"):
            d["ea_write_latency_cycles"] = c.get("TCC_EA0_WRREQ_LEVEL", 0.0) / c["TCC_EA0_WRREQ"]
        if c.get("TCC_CYCLE"):
            d["tcc_busy_share"] = c.get("TCC_BUSY", 0.0) / c["TCC_CYCLE"]
        if c.get("TCC_REQ"):
            d["l2_hit_rate"] = c.get("TCC_HIT", 0.0) / max(1.0, c.get("TCC_HIT", 0.0) + c.get("TCC_MISS", 0.0))
        if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        out["families"][f] = d
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_limiters.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    json.dump(out, open(path, "w"), indent=1)
    keys = ("_launches", "valu_util", "mfma_util", "lds_issue_util", "vmem_issue_util", "share_wait_any", "share_wait_inst_any",
            "share_active_inst_any", "ea_read_latency_cycles", "ea_write_latency_cycles", "tcc_busy_share", "l2_hit_rate",
            "valu_insts_per_wave", "SQ_INSTS_VALU_TRANS_F32", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL", "TCC_EA0_WRREQ_STALL")
    for f, d in out["families"].items():
        print(f, {k: (round(d[k], 4) if isinstance(d[k], float) else d[k]) for k in keys if k in d})
    for n in notes:
        print("pass failed:", n)


if __name__ == "__main__":
    main()
