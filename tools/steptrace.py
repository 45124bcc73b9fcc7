"""Per-launch durations of the LAST bench step from a rocprofv3 kernel trace (csv): a step starts at the second-to-last
solve_clusters launch.  usage: steptrace.py <dir> [-v]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'solve_level' in r['Kernel_Name'] or 'solve_clusters' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'clusters' in r['Kernel_Name']]
last = rows[idx[-4]:]
tot = collections.defaultdict(float); cnt = collections.Counter(); out = []
for i, r in enumerate(last):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    nm = r['Kernel_Name']
    k = 'C' if 'clusters' in nm else 's' if 'small' in nm else 'P' if 'true, true' in nm or 'false, true' in nm else 'L'
    tot[k] += d; cnt[k] += 1
    out.append(f"{i}{k}:{d:.0f}")
if len(sys.argv) > 2: print(' '.join(out))
print('launches', len(last), {k: (cnt[k], round(v)) for k, v in tot.items()}, 'sum', round(sum(tot.values())), 'span', round((int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3))
