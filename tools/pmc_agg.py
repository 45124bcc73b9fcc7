"""Sum rocprofv3 --pmc counters per kernel over the last bench step(s): usage pmc_agg.py <dir> [<dir> ...]"""
import os, csv, glob, sys, collections
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        nm = r['Kernel_Name']
        if 'rsp::dev::solve' not in nm: continue
        k = 'clusters' if 'clusters' in nm else 'small' if 'small' in nm else 'pipe' if ', 1>' in nm else 'lean' if 'solve_level<' in nm else nm[:30]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[(k, r['Counter_Name'])] += 1
    for k in acc:
        print(d.split('/')[-1], k, {c: f"{v:.4g}" for c, v in acc[k].items()}, 'dispatches', max(n[(k, c)] for c in acc[k]))
