import os, csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'permute_vector' in r['Kernel_Name']]
rows = rows[idx[-2]:idx[-1] + 1]
print('launches', len(rows), 'span us', (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3)
seq = []
for r in rows:
    nm = r['Kernel_Name'].split('(')[0].split('::')[-1]
    wg = int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])
    seq.append((nm[:28], wg, round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1)))
hist = collections.Counter()
for nm, wg, d in seq:
    hist[(nm, 1 if wg <= 8 else 2 if wg <= 16 else 3 if wg <= 64 else 4)] += 1
print(sorted(hist.items()))
tot = collections.defaultdict(float)
for nm, wg, d in seq: tot[nm] += d
print(dict(tot))
print(seq)
