"""Condense rocprofv3 outputs under gpurun_out/ into the small, tracked summaries under profiles/.
usage: collect_profiles.py <tag> <stats_dir> <launches_per_step> <bench.json of that run> [<fetch_dir> <write_dir>]"""
import csv, glob, json, sys, collections

tag, stats_dir, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
bench = json.load(open(sys.argv[4]))
out = {"tag": tag, "config": bench["config"], "ms_per_step_under_profiler": bench["ms_per_step"], "command": "rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-rhs 0 --skip-extras"}
# 1. kernel stats (whole process) -> csv copy restricted to our kernels
f = glob.glob(stats_dir + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'rsp::dev' in r['Name']]
with open(f'profiles/{tag}_kernel_stats.csv', 'w', newline='') as g:
    w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
# 2. the last bench step: per-kernel-class time and launch count
f = glob.glob(stats_dir + '/**/*_kernel_trace.csv', recursive=True)[0]
tr = [r for r in csv.DictReader(open(f)) if 'solve_level' in r['Kernel_Name'] or 'solve_clusters' in r['Kernel_Name']]
tr.sort(key=lambda r: int(r['Start_Timestamp']))
last = tr[-n:]
by = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    k = 'solve_clusters' if 'clusters' in r['Kernel_Name'] else 'solve_level'
    by[k][0] += 1
    by[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
out["step"] = {"launches": n, "kernel_us": {k: {"launches": v[0], "total_us": round(v[1], 1), "avg_us": round(v[1] / v[0], 2)} for k, v in by.items()},
               "sum_us": round(sum(v[1] for v in by.values()), 1),
               "span_us": round((int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3, 1)}
# 3. HBM traffic of one step from separate --pmc passes (KB units; gfx950: FETCH_SIZE reads 1/2 of wide coalesced streams)
if len(sys.argv) > 6:
    def total(d, name):
        f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
        rs = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == name and ('solve_level' in r['Kernel_Name'] or 'solve_clusters' in r['Kernel_Name'])]
        rs.sort(key=lambda r: int(r['Dispatch_Id']))
        return sum(float(r['Counter_Value']) for r in rs[-n:])
    fetch_kb, write_kb = total(sys.argv[5], 'FETCH_SIZE'), total(sys.argv[6], 'WRITE_SIZE')
    out["traffic"] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
                      "hbm_bytes_per_step": int((2 * fetch_kb + write_kb) * 1024),
                      "note": "separate --pmc passes; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM section); Infinity-Cache hits are counted"}
json.dump(out, open(f'profiles/{tag}_summary.json', 'w'), indent=1)
print(json.dumps(out["step"]), out.get("traffic"))
