"""Sum a rocprofv3 --pmc counter over the solve kernels of the LAST bench step (csv output).
usage: traffic.py <dir> <counter> <launches per step>"""
import os, csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
name, n = sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == name and ('solve_level' in r['Kernel_Name'] or 'solve_clusters' in r['Kernel_Name'])]
rows.sort(key=lambda r: int(r['Dispatch_Id']))
last = rows[-n:]
by = collections.defaultdict(float)
for r in last:
    k = r['Kernel_Name']
    k = 'clusters' if 'clusters' in k else 'split8' if 'split<8' in k else 'split4' if 'split<4' in k else 'blocked'
    by[k] += float(r['Counter_Value'])
print(name, 'per step:', {k: round(v) for k, v in by.items()}, 'total', round(sum(by.values())))
