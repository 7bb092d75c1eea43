"""Turn ONE rocprofv3 --pmc CSV of bench.py (counters: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE - eight SQ slots + one GRBM slot: one pass, MI355X_MICROARCH.md "rocprofv3 PMC
slots") into profiles/<name>.json: matrix-pipe and LDS utilisation per kernel instantiation of the MFMA families (north_star: "choices
evidenced by rocprof HBM GB/s and MFMA-busy").

    python scripts/collect_mfma_busy.py <counter_collection.csv> <out.json> [top N, default 12] [policy label, default "parity"]

(Round 6: `effective_clock_ghz` only for dispatches of >= 100 us and never above the part's 2.4 GHz; `mfma_busy_lower_bound` = busy cycles / (dispatch
duration x 2.4 GHz), a normalisation that needs no clock estimate.)
Normalisation (as profiles/r2_igemm4_ablation.txt §5): SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1 024 SIMDs, SQ_LDS_IDX_ACTIVE over
its 256 CUs, GRBM_GUI_ACTIVE over its 8 XCDs, so
    mfma_busy   = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (GRBM_GUI_ACTIVE / 8)      fraction of the kernel's cycles the matrix pipes were busy
    lds_active  = (SQ_LDS_IDX_ACTIVE / 256) / (GRBM_GUI_ACTIVE / 8)
    lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                         extra LDS cycles per LDS-array cycle
    wait_inst / wait_any = SQ_WAIT_INST_ANY, SQ_WAIT_ANY / SQ_WAVE_CYCLES           issue stalls / parked (s_waitcnt, barrier) share of wave time
One v_mfma_f32_16x16x32_f16 keeps its SIMD's pipe busy for 16 cycles; split storage issues three of them per algorithmic product, so
`mfma_busy` is the MFMA-ISSUE utilisation (bench.py's mfma_issue_frac at the clock the kernel really ran at), not the algorithmic fraction.
Stamped with the kernel-source digest like the traffic file: bench.py replays it only while the sources are unchanged.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from resshift_amd import build as _b  # noqa: E402

KEYS = ("igemm", "wino_kernel", "swin_mlp", "win_attn", "ae_flash")
acc = defaultdict(lambda: defaultdict(float))
launches = defaultdict(set)
dur_ns = defaultdict(dict)   # kernel -> dispatch -> duration of that dispatch in THIS (counter-collecting) pass
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"]).strip()
    if not any(k in n for k in KEYS):
        continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    did = r.get("Dispatch_Id", r.get("Correlation_Id", len(launches[n])))
    launches[n].add(did)
    try:
        dur_ns[n][did] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    except (KeyError, ValueError):
        pass
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rows = []
for n, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if gui <= 0:
        continue
    lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    wav = c.get("SQ_WAVE_CYCLES", 0.0)
    tns = sum(dur_ns[n].values())
    rows.append({"kernel": n, "launches": len(launches[n]), "gpu_cycles_per_launch": round(gui / max(1, len(launches[n])), 1),
                 # effective shader clock inside this kernel = GRBM_GUI_ACTIVE per XCD / the dispatches' own durations in the same pass (MI355X_MICROARCH.md, DVFS)
                 # (round 6, VERDICT r5 8a: GRBM_GUI_ACTIVE also counts the dispatch's ramp-up / drain outside [start, end], so for SHORT dispatches the
                 # quotient overshoots - 2.5 - 3.0 GHz on a 2.4 GHz part in round 5's file; it is reported for dispatches of >= 100 us only, where
                 # that edge is < 2 % of the window.  mfma_busy_lower_bound normalises by the dispatch duration at the MAXIMUM clock instead.)
                 "avg_us_in_this_pass": round(tns / 1e3 / max(1, len(dur_ns[n])), 2) if tns else None,
                 "effective_clock_ghz": (round(min(gui / tns, 2.4), 3) if tns and tns / max(1, len(dur_ns[n])) >= 1e5 else None),
                 "mfma_busy_lower_bound": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / (tns * 2.4), 4) if tns else None,
                 "share_of_family_cycles": gui,
                 "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / gui, 4),
                 "mfma_instructions_per_launch": round(c.get("SQ_INSTS_MFMA", 0.0) / max(1, len(launches[n])), 1),
                 "lds_active": round(lds / 256.0 / gui, 4), "lds_conflict": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds, 4) if lds else None,
                 "wait_inst": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wav, 4) if wav else None,
                 "wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / wav, 4) if wav else None})
tot = sum(r["share_of_family_cycles"] for r in rows) or 1.0
for r in rows:
    r["share_of_family_cycles"] = round(r["share_of_family_cycles"] / tot, 4)
rows.sort(key=lambda r: -r["share_of_family_cycles"])
fam_busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for c in acc.values()) / 1024.0 / tot
label = sys.argv[4] if len(sys.argv) > 4 else "parity"
out = {"what": f"rocprofv3 --pmc of `bench.py --steps 1 --warmup 1` ({label} policy, batch 32): matrix-pipe / LDS utilisation per kernel instantiation of the MFMA "
               "families, heaviest first; see scripts/collect_mfma_busy.py for the normalisation",
       "family_mfma_busy": round(fam_busy, 4), "instantiations": len(rows), "per_kernel": rows[:top],
       "kernel_source_digest": _b._digest()[:16]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}))
for r in rows[:top]:
    print(f"{r['kernel'][:70]:70s} n={r['launches']:4d} share {r['share_of_family_cycles']:.3f} mfma_busy {r['mfma_busy']:.3f} lds {r['lds_active']:.3f} "
          f"conflict {r['lds_conflict']} wait_inst {r['wait_inst']} wait_any {r['wait_any']} clock {r['effective_clock_ghz']} GHz")
