"""Condense rocprofv3 outputs under gpurun_out/ into the small, tracked summaries under profiles/.
usage: collect_profiles.py <tag> <stats_dir> <bench.json of that run> [<fetch_dir> <write_dir>]
The LAST bench step is the span from the (2 x PARTS)-th-to-last solve_clusters launch to the last one: a step solves PARTS
column parts (pair pipeline, engine.hpp kPairParts), each with one cluster launch in lsolve (its first) and one in ltsolve
(its last)."""
import os, csv, glob, json, os, sys, collections
sys.path.insert(0, '.')
from bench import kernels_sha

PARTS = int(os.environ.get("RSP_PAIR_PARTS", "2"))

tag, stats_dir = sys.argv[1], sys.argv[2]
bench = json.load(open(sys.argv[3]))
ours = lambda name: 'solve_level' in name or 'solve_clusters' in name or 'solve_items' in name


def kernel_class(nm):
    return ('solve_items' if 'solve_items' in nm else 'solve_clusters' if 'clusters' in nm else 'solve_level_small' if 'small' in nm else
            'solve_level<deep ring, 16 columns>' if 'true, 1>' in nm else 'solve_level<deep ring, 64 columns>' if 'true, 4>' in nm else 'solve_level<lean>')


def last_step(rows, key):
    rows = sorted(rows, key=key)
    idx = [i for i, r in enumerate(rows) if 'clusters' in r['Kernel_Name']]
    return rows[idx[-2 * PARTS]:]


out = {"tag": tag, "kernels_sha": kernels_sha(), "config": bench["config"], "ms_per_step_under_profiler": bench["ms_per_step"],
       "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras"}
f = sorted(glob.glob(stats_dir + '/**/*_kernel_stats.csv', recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'rsp::dev' in r['Name']]
with open(f'profiles/{tag}_kernel_stats.csv', 'w', newline='') as g:
    w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
f = sorted(glob.glob(stats_dir + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
last = last_step([r for r in csv.DictReader(open(f)) if ours(r['Kernel_Name'])], lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    nm = r['Kernel_Name']
    k = kernel_class(nm)
    by[k][0] += 1
    by[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
out["step"] = {"launches": len(last),
               "kernel_us": {k: {"launches": v[0], "total_us": round(v[1], 1), "avg_us": round(v[1] / v[0], 2)} for k, v in by.items()},
               "sum_us": round(sum(v[1] for v in by.values()), 1),
               "span_us": round((int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3, 1)}
if len(sys.argv) > 5:
    def total(d, name):
        f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
        rs = last_step([r for r in csv.DictReader(open(f)) if r['Counter_Name'] == name and ours(r['Kernel_Name'])], lambda r: int(r['Dispatch_Id']))
        return sum(float(r['Counter_Value']) for r in rs), len(rs)
    def per_kernel(d, name):
        f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
        rs = last_step([r for r in csv.DictReader(open(f)) if r['Counter_Name'] == name and ours(r['Kernel_Name'])], lambda r: int(r['Dispatch_Id']))
        acc = collections.Counter()
        for r in rs:
            nm = r['Kernel_Name']
            acc[kernel_class(nm)] += float(r['Counter_Value'])
        return acc
    (fetch_kb, n1), (write_kb, n2) = total(sys.argv[4], 'FETCH_SIZE'), total(sys.argv[5], 'WRITE_SIZE')
    pf, pw = per_kernel(sys.argv[4], 'FETCH_SIZE'), per_kernel(sys.argv[5], 'WRITE_SIZE')
    out["traffic"] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "dispatches": [n1, n2],
                      "hbm_bytes_per_step": int((2 * fetch_kb + write_kb) * 1024),
                      "per_kernel_GB": {k: {"read_corrected": round(2 * pf[k] * 1024 / 1e9, 2), "write": round(pw[k] * 1024 / 1e9, 2)} for k in pf},
                      "note": "separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), summed over the launches of one step; FETCH_SIZE "
                              "doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted"}
json.dump(out, open(f'profiles/{tag}_summary.json', 'w'), indent=1)
print(json.dumps(out["step"]), out.get("traffic"))
