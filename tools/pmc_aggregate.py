"""Aggregate the rocprofv3 --pmc passes written by tools/collect_profiles.sh into
<dir>/<tag>_pmc_per_kernel.csv (per-kernel means over launches) and <dir>/<tag>_pmc_render.json (what bench.py
reports as roofline.traffic).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md "HBM"; WRITE_SIZE taken as reported."""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from enerf_amd.build import source_digest      # the digest of the kernel sources these counters were collected on

d, tag = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pmc*.csv"))):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void enerf::", "").replace("enerf::", "")
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
ctrs = sorted({c for k in vals for c in vals[k]})
rows = []
for k in sorted(vals):
    m = {c: (sum(v) / len(v) if v else 0.0) for c, v in vals[k].items()}
    row = {"kernel": k, "launches": max(len(v) for v in vals[k].values())}
    row.update({c: round(m.get(c, 0.0), 3) for c in ctrs})
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    row["mfma_busy_frac"] = round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / 8.0), 4) if gui else ""
    for c in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
        row[c + "_frac"] = round(m.get(c, 0.0) / wc, 4) if wc else ""
    row["hbm_bytes_corrected"] = round((2.0 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024.0, 1)
    rows.append(row)
out = os.path.join(d, f"{tag}_pmc_per_kernel.csv")
with open(out, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
# the dominant render kernel: the full-resolution level (R = 3 channels per lane group) when both levels render
renders = [r for r in rows if r["kernel"].startswith("k_render_rays")]
renders.sort(key=lambda r: 0 if r["kernel"].startswith("k_render_rays<3") else 1)
for r in renders[:1]:
    if True:
        js = {"kernel": r["kernel"], "source": f"rocprofv3 --pmc passes (tools/collect_profiles.sh), profiles/{tag}_pmc_per_kernel.csv",
              "fetch_size_kb": r.get("FETCH_SIZE"), "write_size_kb": r.get("WRITE_SIZE"),
              "hbm_bytes_per_launch": r["hbm_bytes_corrected"],
              "correction": "gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md HBM section), WRITE_SIZE as reported",
              "mfma_busy_frac": r["mfma_busy_frac"], "wait_inst_any_frac": r.get("SQ_WAIT_INST_ANY_frac"),
              "source_digest": source_digest()}
        json.dump(js, open(os.path.join(d, f"{tag}_pmc_render.json"), "w"), indent=1)
        print(js)
print("wrote", out)
