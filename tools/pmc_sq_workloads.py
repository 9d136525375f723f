"""Matrix-pipe busy fraction and wait shares of whole workloads from the SQ counter passes of tools/collect_profiles.sh:

    python3 tools/pmc_sq_workloads.py <dir with sq_<tag>_counter_collection.csv> <out.json>

Per workload, summed over all its dispatches: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CYCLES x CUs-per-SE-normalisation) is
not portable across counter definitions, so the fraction is formed from quantities with the same unit: SQ_VALU_MFMA_BUSY_CYCLES counts cycles
per SIMD with the matrix pipe busy, SQ_WAVE_CYCLES counts quad-cycles of resident waves (MI355X_MICROARCH.md: x4 = cycles).  For kernels that
keep ONE wave per SIMD resident for their whole life (the conv kernels) busy / (4 x wave cycles) is the pipe's busy share of the kernel; for
workloads with many waves per SIMD it is a lower bound.  The per-kernel rows carry both numbers and the wait shares, the workload row the
total over the conv kernels (one wave per SIMD), which is what the roofline fractions are about."""
import csv, glob, json, os, sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
doc = {"formula": "mfma_busy_frac = waves_per_SIMD * SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES) over the conv kernels of the workload (one wave per SIMD; two for conv_f9h_kernel's two-workgroups-per-CU builds) "
                  "(SQ_WAVE_CYCLES counts quad-cycles); wait_any / wait_inst = SQ_WAIT_ANY / SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; "
                  "lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE", "workloads": {}}
ONE_WAVE = ("conv_fwd9_kernel", "conv_wgrad3_kernel", "conv_f9h_kernel", "conv_wgrad_h_kernel", "conv_fwd8_kernel")
for f in sorted(glob.glob(os.path.join(d, "sq_*_counter_collection.csv"))):
    tag = os.path.basename(f)[3:-len("_counter_collection.csv")]
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for r in csv.DictReader(open(f)):
        raw = r["Kernel_Name"]
        name = raw.split("(")[0].split("<")[0].replace("void ", "").strip().split("::")[-1]
        if "conv_f9h_kernel" in raw:             # its 256-voxel builds keep TWO workgroups per CU = two waves per SIMD (Cfg<..., NIMG, 2>)
            import re
            m = re.search(r"Cfg<([0-9, ]+)>", raw)
            a = [int(v) for v in m.group(1).split(",")] if m else []
            name += "<two workgroups per CU>" if len(a) >= 8 and a[7] == 2 else "<one workgroup per CU>"
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[name] += 1
    kernels, tb, tw = {}, 0.0, 0.0
    for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        w = c.get("SQ_WAVE_CYCLES", 0.0)
        if w <= 0:
            continue
        occ = 2.0 if name.endswith("<two workgroups per CU>") else 1.0      # waves per SIMD: the pipe's busy share of the SIMD's time
        row = dict(launches=n[name], mfma_busy_frac=round(occ * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * w), 4),
                   wait_any=round(c.get("SQ_WAIT_ANY", 0.0) / w, 4), wait_inst=round(c.get("SQ_WAIT_INST_ANY", 0.0) / w, 4),
                   lds_conflict_frac=round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4),
                   wave_quad_cycles=int(w))
        if len(kernels) < 12:
            kernels[name] = row
        if any(k in name for k in ONE_WAVE):
            tb += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); tw += w / occ
    doc["workloads"][tag] = dict(mfma_busy_frac=round(tb / (4.0 * tw), 4) if tw else None, kernels=kernels)
    print(tag, doc["workloads"][tag]["mfma_busy_frac"], list(kernels.items())[:3])
json.dump(doc, open(out, "w"), indent=1)
