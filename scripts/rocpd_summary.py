"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, grouped by template.

    python scripts/rocpd_summary.py <results.db> [--top N] > profiles/<name>.txt
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:110]


def main():
    db = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = con.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
    fam = defaultdict(lambda: [0, 0.0])
    for n, s, e in rows:
        d = (e - s) / 1e3  # us
        k = short(n)
        a = agg[k]
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
        f = fam[k.split("<")[0]]
        f[0] += 1; f[1] += d
    total = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {db}: {len(rows)} dispatches, {total/1e3:.2f} ms of kernel time")
    print("\n## by kernel family")
    print(f"{'family':40s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}")
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:40s} {c:8d} {t/1e3:10.3f} {t/c:10.2f} {100*t/total:6.2f}%")
    print(f"\n## top {top} kernel instantiations")
    print(f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>7s}")
    for k, (c, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k:112s} {c:7d} {t/1e3:10.3f} {t/c:9.2f} {mn:9.2f} {mx:9.2f} {100*t/total:6.2f}%")


if __name__ == "__main__":
    main()
