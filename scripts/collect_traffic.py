"""Turn two rocprofv3 --pmc CSVs (FETCH_SIZE pass, WRITE_SIZE pass) of bench.py into profiles/<name>.json with the HBM
traffic of the MFMA implicit-GEMM kernel family (incl. the fused Swin kernels, as in bench.py's roofline) per launch.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM):
counter unit = KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> x2; WRITE_SIZE taken as is.

    python scripts/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import build as _b  # kernel-source digest: bench.py ignores the file when the sources have changed since

def total(path, counter, keys=("igemm", "wino_kernel", "swin_mlp", "win_attn_qkv", "ae_flash_attn")):
    s = 0.0; n = 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in keys):
            s += float(r["Counter_Value"]); n += 1
    return s, n

f, nf = total(sys.argv[1], "FETCH_SIZE")
w, nw = total(sys.argv[2], "WRITE_SIZE")
out = {"kernel_family": "igemm*_kernel + wino_kernel + swin_mlp*_kernel + win_attn_qkv*_kernel + ae_flash_attn_kernel", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_per_launch": 2.0 * f * 1024 / max(1, nf), "write_bytes_per_launch": w * 1024 / max(1, nw),
       "note": "FETCH_SIZE x2 (gfx950 wide-read correction), KiB units, separate --pmc passes"}
out["hbm_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
# the HBM-bound GroupNorm family (gn_stats / gn_apply / gn_fused kernels), same passes, same corrections
GN = ("gn_stats_kernel", "gn_apply_kernel", "gn_fused_kernel")
gf, gnf = total(sys.argv[1], "FETCH_SIZE", GN)
gw, gnw = total(sys.argv[2], "WRITE_SIZE", GN)
out["groupnorm"] = {"kernel_family": " / ".join(GN), "launches_fetch_pass": gnf, "launches_write_pass": gnw,
                    "fetch_bytes_per_launch": 2.0 * gf * 1024 / max(1, gnf), "write_bytes_per_launch": gw * 1024 / max(1, gnw)}
out["groupnorm"]["hbm_bytes_per_launch"] = out["groupnorm"]["fetch_bytes_per_launch"] + out["groupnorm"]["write_bytes_per_launch"]
# per kernel instantiation of both families (attribute the traffic): launches, HBM MB fetched (x2 corrected) / written per launch
import re
def per_kernel(path, counter, scale):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        n = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"]).strip()
        if not any(k in n for k in ("igemm", "wino_kernel", "swin_mlp", "win_attn", "ae_flash", "gn_", "splitk_reduce", "head_conv")): continue
        a = acc.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024 * scale
    return acc
pf, pw = per_kernel(sys.argv[1], "FETCH_SIZE", 2.0), per_kernel(sys.argv[2], "WRITE_SIZE", 1.0)
out["per_kernel"] = {n: {"launches": pf[n][0], "fetch_mb_per_launch": round(pf[n][1] / pf[n][0] / 1e6, 3),
                         "write_mb_per_launch": round(pw.get(n, [1, 0.0])[1] / max(1, pw.get(n, [1, 0.0])[0]) / 1e6, 3),
                         "total_gb": round((pf[n][1] + pw.get(n, [0, 0.0])[1]) / 1e9, 3)} for n in sorted(pf, key=lambda k: -(pf[k][1] + pw.get(k, [0, 0.0])[1]))}
out["kernel_source_digest"] = _b._digest()[:16]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
