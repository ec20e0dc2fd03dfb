"""Turn rocprofv3 FETCH_SIZE / WRITE_SIZE passes into profiles/<tag>_traffic.json.

HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB and, on
gfx950, FETCH_SIZE reports half the bytes of a wide coalesced streaming read
(/opt/skills/guides/MI355X_MICROARCH.md §HBM).  WRITE_SIZE is uncalibrated there and used as is.
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


def mean_by_family(path, counter):
    acc = collections.defaultdict(list)
    for r in rows(path):
        f = family(r["Kernel_Name"])
        if f and r["Counter_Name"] == counter:
            acc[f].append(float(r["Counter_Value"]))
    return {f: (sum(v) / len(v), len(v)) for f, v in acc.items()}


def main():
    fetch = mean_by_family(sys.argv[1], "FETCH_SIZE")
    write = mean_by_family(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py, B/16 batch 512",
           "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch (gfx950 FETCH half-count correction)",
           "per_launch_bytes": {}}
    for f in fetch:
        fb, n = fetch[f]
        wb = write.get(f, (0.0, 0))[0]
        out["per_launch_bytes"][f] = {"launches_sampled": n, "fetch_kib_raw": fb, "write_kib_raw": wb,
                                      "hbm_bytes": (2 * fb + wb) * 1024}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["per_launch_bytes"], indent=1))


if __name__ == "__main__":
    main()
