"""Launch durations of the LAST single-RHS solve in a kernel trace: histogram per kernel."""
import os, csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'rsp::dev' in r['Kernel_Name'] and 'chol_' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'colmajor_to_rowmajor' in r['Kernel_Name']]
rows = rows[idx[-1]:]
by = collections.defaultdict(list)
for r in rows:
    nm = r['Kernel_Name']
    k = 'clusters' if 'clusters' in nm else 'small' if 'small' in nm else 'pipe' if ', 1>' in nm else 'lean' if 'solve_level<' in nm else nm.split('(')[0][-24:]
    by[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
print('span us', round(span), 'launches', len(rows))
for k, v in by.items():
    v.sort()
    print(k, 'n', len(v), 'sum', round(sum(v)), 'min', round(v[0], 1), 'median', round(v[len(v) // 2], 1), 'max', round(v[-1], 1))
