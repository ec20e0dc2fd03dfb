"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch).
Usage: pmc_summary.py <counter_collection.csv> [...]; GEMM rows are split by grid/LDS size."""
import collections
import csv
import re
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


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
    return name[:48]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        for r in rows(path):
            if "rocclr" in r["Kernel_Name"] or "at::native" in r["Kernel_Name"]:
                continue
            key = short(r["Kernel_Name"])
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in agg.values() for c in v})
    derived = []
    if {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"} <= set(names):
        derived.append("MFMA_BUSY_%")      # busy cycles summed over 1024 SIMDs / (1024 x kernel cycles)
    if {"SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"} <= set(names):
        derived.append("LDS_CONFLICT_%")   # extra LDS-array cycles / all LDS-array cycles
    print(f"{'kernel':50s} {'launches':>8s} " + " ".join(f"{n:>22s}" for n in names + derived))
    for k, v in sorted(agg.items()):
        n = max(len(x) for x in v.values())
        mean = {c: (sum(v[c]) / len(v[c]) if v.get(c) else float("nan")) for c in names}
        extra = []
        for d in derived:
            if d == "MFMA_BUSY_%":
                g = mean["GRBM_GUI_ACTIVE"]
                extra.append(100.0 * mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * g) if g else float("nan"))
            else:
                a = mean["SQ_LDS_IDX_ACTIVE"]
                extra.append(100.0 * mean["SQ_LDS_BANK_CONFLICT"] / a if a else 0.0)
        print(f"{k:50s} {n:8d} " + " ".join(f"{mean[c]:22.4g}" for c in names) + " " +
              " ".join(f"{x:22.2f}" for x in extra))
    if derived:
        print("# MFMA_BUSY_% = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE): GRBM_GUI_ACTIVE is summed over the 8 XCDs "
              "(checked against kernel duration x clock), so kernel cycles = GRBM / 8 and 1024 SIMDs x GRBM / 8 = 128 x GRBM; "
              "LDS_CONFLICT_% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (means per launch)")


if __name__ == "__main__":
    main()
