"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, MI355X_MICROARCH.md §HBM):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --mode sample --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timer
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ... (same)
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE are KB per dispatch; FETCH_SIZE is doubled for gfx950 (16-byte-per-lane reads are tallied at half)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(root, counter):
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no *counter_collection.csv under {root}"
    tot, cnt = defaultdict(float), defaultdict(int)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                # base name: template arguments dropped, so that the variants of one kernel (conv_fwd9_kernel<F9Cfg<...>>) are averaged
                # together, the way bench.py's per-kernel timer tags them
                name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").strip()
                tot[name] += float(r["Counter_Value"])
                cnt[name] += 1
    return tot, cnt


def main(fetch_dir, write_dir, out, cmd=None):
    ft, fc = collect(fetch_dir, "FETCH_SIZE")
    wt, wc = collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for name in sorted(ft, key=lambda k: -(ft[k] * 2 + wt.get(k, 0.0))):
        n = fc[name]
        fkb, wkb = ft[name] / n, wt.get(name, 0.0) / max(wc.get(name, 0), 1)
        kernels[name] = dict(launches=n, fetch_size_raw_kb_per_launch=round(fkb, 2), write_size_kb_per_launch=round(wkb, 2),
                             hbm_bytes_per_launch=int(round((2.0 * fkb + wkb) * 1024)))
    doc = dict(command="rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (two separate passes) -- " +
                       (cmd or "python3 bench.py --mode sample --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-kernel-timer"),
               units="FETCH_SIZE / WRITE_SIZE are KB per dispatch (rocprofv3 derived metrics); FETCH_SIZE is doubled for gfx950 "
                     "(16-byte-per-lane reads are tallied at half, MI355X_MICROARCH.md §HBM)",
               kernels=dict(list(kernels.items())[:24]))
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for k, v in list(kernels.items())[:8]:
        print(f"{v['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch x {v['launches']:5d}  {k[-70:]}")


if __name__ == "__main__":
    main(*sys.argv[1:5])
