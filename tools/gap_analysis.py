"""GPU idle time inside a sampler step, from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python3 bench.py --mode sample --steps 10 --warmup 3 \
        --no-cpu-baseline --no-family-b --no-kernel-timer
    python3 tools/gap_analysis.py gpurun_out/gap

A DDPM sampler step ends with one `ddpm_step_kernel`; for every interval between two consecutive ones the script prints
wall time, the sum of kernel durations, the idle share and the number of launches, then the kernels that precede the largest gaps.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no *kernel_trace.csv under {root}"
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "ddpm_step_kernel" in r[2]]
    print(f"{len(rows)} kernels, {len(marks)} sampler steps")
    gaps_after = defaultdict(lambda: [0, 0])
    tot_wall = tot_busy = 0
    for a, b in zip(marks[:-1], marks[1:]):
        seg = rows[a + 1:b + 1]
        wall = seg[-1][1] - rows[a][1]
        busy = sum(e - s for s, e, _ in seg)
        prev_end, prev_name = rows[a][1], rows[a][2]
        for s, e, n in seg:
            gap = max(0, s - prev_end)
            key = prev_name.split("(")[0][-60:]
            gaps_after[key][0] += gap
            gaps_after[key][1] += 1
            prev_end, prev_name = max(prev_end, e), n
        tot_wall += wall
        tot_busy += busy
        print(f"step: wall {wall / 1e6:8.3f} ms  busy {busy / 1e6:8.3f} ms  idle {100 * (1 - busy / wall):5.1f} %  launches {len(seg)}")
    if tot_wall:
        print(f"TOTAL idle {100 * (1 - tot_busy / tot_wall):.1f} % of {tot_wall / 1e6:.2f} ms over {len(marks) - 1} steps")
    print("largest cumulative gaps by preceding kernel (ms, count, avg us):")
    for k, (g, n) in sorted(gaps_after.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  {g / 1e6:8.3f}  {n:6d}  {g / n / 1e3:7.2f}  {k}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gap")
