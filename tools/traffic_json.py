"""Turn rocprofv3 FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_traffic.json.

L2<->fabric bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and, on
gfx950, FETCH_SIZE reports half the bytes of a wide coalesced streaming read
(/opt/skills/guides/MI355X_MICROARCH.md §HBM).  WRITE_SIZE is uncalibrated there and used as is.
These are the L2's memory-side request counters: traffic that misses the XCD's L2 and goes to the fabric —
Infinity-Cache (MALL) hits INCLUDED, so this is an upper bound on HBM bytes, not HBM bytes.
GEMM launches are additionally reported per shape class (kernel template arguments <precision, epilogue>:
epilogue 0 = QKV (+ the last layer's K/V-only launch), 1 = fc1, 2 = the residual GEMMs out-proj and fc2, which
share an instantiation, 3 = patch embedding) next to their algorithmic bytes at B/16 batch 512.
Usage: traffic_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def rows(path):
    """(kernel name, counter name, value) per dispatch from a rocprofv3 counter_collection CSV or from the
    rocpd SQLite database rocprofv3 7.2 writes by default (view `counters_collection`)."""
    if path.endswith(".db"):
        import sqlite3

        db = sqlite3.connect(path)
        for name, cname, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
            yield {"Kernel_Name": name, "Counter_Name": cname, "Counter_Value": val}
    else:
        yield from csv.DictReader(open(path))


def family(name):
    if "gemm_" in name:  # <1, ...> = exact-fp32 kernels: the text tower's one pass, not the vision step
        return "gemm_fp32_text_tower" if "_kernel<1," in name else "gemm"
    for k in ("attn", "layernorm", "patchify", "score", "pool_project"):
        if k in name:
            return k
    return None


GEMM_SHAPES = {"0": "gemm_qkv", "1": "gemm_fc1", "2": "gemm_resid_outproj_fc2", "3": "gemm_patch"}
# algorithmic bytes per launch at B/16 batch 512 (M = 100 864 rows, 16-bit operands, fp32 residual):
# X + W read once + output (+ residual read for the read-modify-write form)
ALGO = {"gemm_qkv": 100864 * 768 * 2 + 2304 * 768 * 2 + 100864 * 2304 * 2,
        "gemm_fc1": 100864 * 768 * 2 + 3072 * 768 * 2 + 100864 * 3072 * 2,
        "gemm_resid_outproj_fc2": ((100864 * 768 * 2 + 768 * 768 * 2 + 2 * 100864 * 768 * 4)
                                   + (100864 * 3072 * 2 + 768 * 3072 * 2 + 2 * 100864 * 768 * 4)) // 2,
        # patch embedding, pixel-gathering form (round 4): fp32 NCHW pixels read once + W + the fp32 residual rows written
        "gemm_patch": 512 * 3 * 224 * 224 * 4 + 768 * 768 * 2 + 100352 * 768 * 4}


def gemm_shape(name):
    import re

    m = re.search(r"gemm_\w+_kernel<(\d+), (\d+)", name)
    if not m or m.group(1) == "1":  # exact-fp32 kernels = the text tower
        return None
    return GEMM_SHAPES.get(m.group(2))


def mean_by_family(path, counter):
    acc = collections.defaultdict(list)
    for r in rows(path):
        if r["Counter_Name"] != counter:
            continue
        for f in (family(r["Kernel_Name"]), gemm_shape(r["Kernel_Name"])):
            if f:
                acc[f].append(float(r["Counter_Value"]))
    return {f: (sum(v) / len(v), len(v)) for f, v in acc.items()}


def main():
    fetch = mean_by_family(sys.argv[1], "FETCH_SIZE")
    write = mean_by_family(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py, B/16 batch 512",
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950 FETCH half-count correction)",
           "meaning": "L2<->fabric bytes (L2 misses and write-backs; Infinity-Cache hits included) - an upper bound on HBM bytes",
           "per_launch_bytes": {}}
    for f in fetch:
        fb, n = fetch[f]
        wb = write.get(f, (0.0, 0))[0]
        out["per_launch_bytes"][f] = {"launches_sampled": n, "fetch_kib_raw": fb, "write_kib_raw": wb,
                                      "l2_fabric_bytes": (2 * fb + wb) * 1024, "hbm_bytes": (2 * fb + wb) * 1024}
        if f in ALGO:
            out["per_launch_bytes"][f]["algorithmic_bytes"] = ALGO[f]
            out["per_launch_bytes"][f]["ratio"] = (2 * fb + wb) * 1024 / ALGO[f]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["per_launch_bytes"], indent=1))


if __name__ == "__main__":
    main()
