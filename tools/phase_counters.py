#!/usr/bin/env python3
"""Sums up tools/phase_counters.sh: gpurun_out/phase_<tag>_<variant>.txt -> profiles/<tag>_C2_phase_counters.json
(per-dispatch means of the SQ counters for the product kernel, the two ablations and the seven phase cuts) and prints
the instruction counts per phase (the difference of two neighbouring cuts, per tile with candidates).
usage: python3 tools/phase_counters.py <tag> [tiles_with_candidates]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
hit_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 4914
out = {}
for v in ("full", "abl4", "abl1", "cut8", "cut9", "cut10", "cut11", "cut12", "cut13", "cut14"):
    p = os.path.join(ROOT, "gpurun_out", f"phase_{tag}_{v}.txt")
    if not os.path.exists(p):
        continue
    d = {}
    for ln in open(p):
        m = re.match(r"\s+(\w+)\s+per-dispatch mean\s+([0-9.e+-]+)", ln)
        if m:
            d[m.group(1)] = float(m.group(2))
    out[v] = d
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_C2_phase_counters.json"), "w"), indent=1)
names = {"cut8": "prologue, lane's primitive, ray set-up", "cut9": "closest hits", "cut10": "hit records", "cut11": "shadow bundle",
         "cut12": "shadow candidates", "cut13": "shadow rays", "cut14": "shading", "full": "combine, sky for misses, stores"}
order = ["cut8", "cut9", "cut10", "cut11", "cut12", "cut13", "cut14", "full"]
sky = out.get("abl1", {})
print("every tile sky: VALU %.0f per tile" % (sky.get("SQ_INSTS_VALU", 0) / 16200))
prev = None
for v in order:
    if v not in out:
        continue
    if prev is not None:
        dv = (out[v]["SQ_INSTS_VALU"] - out[prev]["SQ_INSTS_VALU"]) / hit_tiles
        ds = (out[v]["SQ_INSTS_SALU"] - out[prev]["SQ_INSTS_SALU"]) / hit_tiles
        print("%-42s VALU %+7.0f  SALU %+7.0f per tile with candidates" % (names[v], dv, ds))
    prev = v
