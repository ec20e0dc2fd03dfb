"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel stats table:
calls, total / average / min / max duration, share of GPU time; GEMM rows are split by grid
size so each logical GEMM shape gets its own line.  Usage: rocpd_summary.py <db> [skip_calls]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, grid_x, duration from kernels order by start").fetchall()
    agg = {}
    for name, gx, dur in rows:
        key = (short(name), gx if "gemm" in name else 0)
        a = agg.setdefault(key, [0, 0, 10 ** 18, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':72s} {'grid':>9s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s}")
    for (name, gx), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:72s} {gx:9d} {a[0]:6d} {a[1] / 1e6:10.3f} {a[1] / a[0] / 1e3:9.1f} {a[2] / 1e3:9.1f} {a[3] / 1e3:9.1f} {100 * a[1] / tot:6.2f}")
    print(f"total GPU kernel time: {tot / 1e6:.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
