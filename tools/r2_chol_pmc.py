"""Per-dispatch counters of the chain-level chol_panel_factor launches (grid of 1-8 workgroups) from a rocprofv3 --pmc run."""
import csv, glob, os, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'chol_panel_factor' in r['Kernel_Name']]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    by[int(r['Dispatch_Id'])]['grid'] = int(r['Grid_Size'])
small = [v for k, v in sorted(by.items()) if v['grid'] <= 256 * 2]
big = [v for k, v in sorted(by.items()) if 256 * 2 < v['grid'] <= 256 * 16]
for name, grp in (('<=2 workgroups', small), ('3..16 workgroups', big)):
    if not grp: continue
    keys = [k for k in grp[0] if k != 'grid']
    print(name, len(grp), {k: round(sum(g[k] for g in grp) / len(grp), 1) for k in keys})
