"""Per-kernel totals and the slowest launches of the LAST factorisation in a rocprofv3 kernel trace (csv)."""
import os, csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'chol_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last factorisation = after the last chol_scatter_A
last = max(i for i, r in enumerate(rows) if 'scatter' in r['Kernel_Name'])
rows = rows[last:]
tot = collections.Counter(); cnt = collections.Counter()
seq = []
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r['Kernel_Name'].split('(')[0].split('::')[-1]
    tot[k] += d; cnt[k] += 1
    seq.append((d, k, int(r['Grid_Size_X']) // 64))
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3
print('span us', span, {k: (round(tot[k]), cnt[k]) for k in tot})
print('slowest', sorted(seq, reverse=True)[:12])
print('first 12', [(round(d), k[5:], w) for d, k, w in seq[:12]])
print('last 12', [(round(d), k[5:], w) for d, k, w in seq[-12:]])
