"""Join RSP_DUMP_LEVELS output with a rocprofv3 kernel trace of a NON-overlapped run (RSP_PAIR_OVERLAP=0):
per level: work (matrix-core stages, vector entries, cluster count), launches and their durations."""
import os, csv, glob, re, sys
dump, prof, nchunks = sys.argv[1], sys.argv[2], int(sys.argv[3])
levels = []
for line in open(dump):
    m = re.match(r"L(\d+) tasks (\d+)/(\d+)/(\d+)/(\d+) blocks (\d+) rows (\d+) mfma_stages (\d+) max (\d+) vec_blocks (\d+) vec_entries (\d+) clusters (\d+)", line)
    if m: levels.append([int(x) for x in m.groups()])
f = sorted(glob.glob(prof + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'solve_level' in r['Kernel_Name'] or 'solve_clusters' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# expected launches of one step (lsolve levels then ltsolve levels)
exp = []
for lv in levels:
    l, c0, c1, c2, c3, blocks, nrows, stages, mx, vb, ve, cl = lv
    if cl: exp.append(('C', lv))
    small = c0 > 0 and c0 * nchunks >= 2048
    if small: exp.append(('S', lv))
    if c1 + c2 + c3 + (0 if small else c0) > 0: exp.append(('U', lv))
n = len(exp)
last = rows[-n:]
tot = {}
out = []
for (kind, lv), r in zip(exp, last):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    nm = r['Kernel_Name']
    ok = ('clusters' in nm) == (kind == 'C') and ('small' in nm) == (kind == 'S')
    l, c0, c1, c2, c3, blocks, nrows, stages, mx, vb, ve, cl = lv
    cls = 'chain' if kind == 'U' and stages * nchunks < 2000 else kind
    tot[cls] = tot.get(cls, 0) + d
    out.append((kind, l, d, c0, c1, c2, c3, stages, mx, ve, cl, ok))
print('launches', n, 'totals us', {k: round(v) for k, v in tot.items()})
for o in out:
    kind, l, d, c0, c1, c2, c3, stages, mx, ve, cl, ok = o
    if kind == 'U' and stages * nchunks >= 2000:
        # stage = 16 sources x 16 rows x 64 cols: bytes gathered 8 KiB + 2 KiB coefficients per (stage, chunk)
        gb = stages * nchunks * 10240 / 1e9
        print(f"L{l} U {d:7.1f}us tasks {c0}/{c1}/{c2}/{c3} stages {stages} max {mx}  -> {gb / (d * 1e-6):7.0f} GB/s gathered+coef {'' if ok else 'MISMATCH'}")
