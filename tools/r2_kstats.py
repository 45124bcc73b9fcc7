"""Per-kernel totals per bench step from a rocprofv3 --kernel-trace --stats run.  usage: r2_kstats.py <dir> <steps in the run>"""
import csv, glob, os, sys, re
d, steps = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(d + '/**/*_kernel_stats.csv', recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'rsp::dev' in r['Name']]
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    name = re.sub(r'rsp::dev::', '', r['Name'])[:70]
    calls, ns = int(r['Calls']), float(r['TotalDurationNs'])
    if 'solve' in name:
        tot += ns
    print(f"{name:70s} calls/step {calls/steps:8.1f} us/step {ns/1e3/steps:9.1f} avg_us {ns/1e3/calls:8.2f}")
print(f"solve kernels per step: {tot/1e3/steps:.1f} us")
