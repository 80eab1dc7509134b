#!/usr/bin/env python3
"""Merge rocprofv3 --pmc counter_collection.csv files (one per pass) into profiles/<name>.json: per kernel the mean of
every counter per dispatch and the derived figures bench.py quotes.  usage: pmc_to_json.py out.json pass1.csv pass2.csv ...
Units (MI355X_MICROARCH.md): SQ_*_CYCLES / ACTIVE / WAIT counters are quad-cycles summed over the chip, SQ_BUSY_CYCLES over
32 shader engines, SQ_LDS_* in LDS cycles summed over 256 CUs; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled (gfx950
tallies 128-byte read requests at 64 bytes)."""
import collections, csv, json, re, sys
out, paths = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in paths:
    with open(p) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"^void ", "", row.get("Kernel_Name", ""))
            name = re.sub(r"^miba::", "", name).split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
import os
res = {"source": os.environ.get("PMC_SOURCE", "rocprofv3 --pmc passes over tools/time_k3.py final-13682 (default tiles)") +
                 ", means per dispatch; merged by tools/pmc_to_json.py (units in its docstring)"}
for name, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = dict(m)
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        d["traffic_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    if "SQ_BUSY_CYCLES" in m:
        kc = m["SQ_BUSY_CYCLES"] / 32.0
        d["kernel_cycles"] = kc
        if "SQ_ACTIVE_INST_VALU" in m: d["valu_busy_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (kc * 1024.0)
        if "SQ_LDS_IDX_ACTIVE" in m: d["lds_busy_frac"] = m["SQ_LDS_IDX_ACTIVE"] / (kc * 256.0)
        if "SQ_WAVE_CYCLES" in m:
            d["waves_resident_per_simd"] = 4.0 * m["SQ_WAVE_CYCLES"] / (kc * 1024.0)
            if "SQ_WAIT_ANY" in m: d["wait_any_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
    res[name] = d
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items():
    if isinstance(v, dict):
        print(k[:50], {a: round(b, 3) for a, b in v.items() if a.endswith("frac") or a.startswith("waves") or a.startswith("traffic")})
