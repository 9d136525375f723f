"""Print the top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/kstats.py <dir> [N] [copy_to]"""
import csv, glob, os, shutil, sys
d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))
assert f, f"no *kernel_stats.csv under {d}"
rows = list(csv.DictReader(open(f[-1])))
for r in rows[:n]:
    print(f"{float(r['Percentage']):6.2f}%  {int(r['Calls']):6d} x {float(r['AverageNs']) / 1e3:10.2f} us  {r['Name'][:110]}")
if len(sys.argv) > 3:
    shutil.copy(f[-1], sys.argv[3])
