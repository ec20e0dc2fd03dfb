"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch).
Usage: pmc_summary.py <counter_collection.csv> [...]; GEMM rows are split by grid/LDS size."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
    return name[:48]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            if "rocclr" in r["Kernel_Name"] or "at::native" in r["Kernel_Name"]:
                continue
            key = short(r["Kernel_Name"])
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in agg.values() for c in v})
    print(f"{'kernel':50s} {'launches':>8s} " + " ".join(f"{n:>22s}" for n in names))
    for k, v in sorted(agg.items()):
        n = max(len(x) for x in v.values())
        print(f"{k:50s} {n:8d} " + " ".join(f"{(sum(v[c]) / len(v[c]) if v.get(c) else float('nan')):22.4g}" for c in names))


if __name__ == "__main__":
    main()
