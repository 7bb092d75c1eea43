"""GroupNorm family time per bench pass from a rocprofv3 --kernel-trace database of `bench.py --precision <policy> --steps K --warmup W
--no-cpu-baseline --no-profile-pass` (K + W passes): profiles/<name>.json, stamped with the kernel-source digest (bench.py reports it
only while the sources are unchanged).  The hipEvent brackets of bench.py's profile pass cannot resolve 6 - 14 us kernels (VERDICT r2
weak #5); the trace can.

    python scripts/collect_gn_trace.py <results.db> <passes in the trace> <out.json>
"""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import build as _b

db, passes, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
namecol = "name" if "name" in cols else "kernel_name"
tot = {}
for n, s, e in con.execute(f"select {namecol}, start, end from kernels"):
    for k in ("gn_stats_kernel", "gn_apply_kernel", "gn_fused_kernel"):
        if k in n:
            t = tot.setdefault(k, [0, 0.0]); t[0] += 1; t[1] += (e - s) / 1e6
res = {"passes": passes, "kernels": {k: {"launches_per_pass": v[0] / passes, "ms_per_pass": v[1] / passes} for k, v in tot.items()},
       "launches_per_pass": sum(v[0] for v in tot.values()) / passes, "ms_per_pass": sum(v[1] for v in tot.values()) / passes,
       "kernel_source_digest": _b._digest()[:16]}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
