"""Per-launch durations of the last bench step from a rocprofv3 kernel trace (csv)."""
import os, csv, sys, glob
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
n = int(sys.argv[2])  # launches per step
rows = [r for r in csv.DictReader(open(f)) if 'solve_level' in r['Kernel_Name'] or 'solve_clusters' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-n:]
tot = {'B': 0.0, 'C': 0.0}
out = []
for i, r in enumerate(last):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    nm = r['Kernel_Name']
    k = 'C' if 'clusters' in nm else 'B'
    tot[k] += d
    out.append(f"{i}{k}:{int(r['Grid_Size_X'])//64}w:{d:.0f}us")
print(' '.join(out))
print('sum us', tot, 'span us', (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3)
