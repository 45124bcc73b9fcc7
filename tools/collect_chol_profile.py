"""Condense rocprofv3 runs of tools/chol_prof.py into profiles/<tag>_summary.json.
usage: collect_chol_profile.py <tag> <trace_dir> <fetch_dir> <write_dir>
The LAST factorisation of the run = everything from the last chol_scatter_A launch on."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, '.')
from bench import chol_sha

tag, trace_dir, fetch_dir, write_dir = sys.argv[1:5]
short = lambda nm: nm.split('(')[0].split('::')[-1]


def last_fact(rows, key):
    rows = sorted(rows, key=key)
    start = max(i for i, r in enumerate(rows) if 'chol_scatter_A' in r['Kernel_Name'])
    return rows[start:]


f = sorted(glob.glob(trace_dir + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = last_fact([r for r in csv.DictReader(open(f)) if 'chol_' in r['Kernel_Name']], lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    by[short(r['Kernel_Name'])][0] += 1
    by[short(r['Kernel_Name'])][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
out = {"tag": tag, "chol_sha": chol_sha(), "command": "rocprofv3 --kernel-trace | --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python tools/chol_prof.py 1024",
       "launches": len(rows), "span_us": round((int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3, 1),
       "kernel_us": {k: {"launches": v[0], "total_us": round(v[1], 1), "avg_us": round(v[1] / v[0], 2)} for k, v in by.items()}}


def counter(d, name):
    f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)[-1]
    rs = last_fact([r for r in csv.DictReader(open(f)) if r['Counter_Name'] == name and 'chol_' in r['Kernel_Name']], lambda r: int(r['Dispatch_Id']))
    per = collections.Counter()
    for r in rs:
        per[short(r['Kernel_Name'])] += float(r['Counter_Value'])
    return sum(per.values()), per


fetch_kb, pf = counter(fetch_dir, 'FETCH_SIZE')
write_kb, pw = counter(write_dir, 'WRITE_SIZE')
out["traffic"] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "hbm_bytes_per_factorisation": int((2 * fetch_kb + write_kb) * 1024),
                  "per_kernel_GB": {k: {"read_corrected": round(2 * pf[k] * 1024 / 1e9, 3), "write": round(pw[k] * 1024 / 1e9, 3)} for k in pf},
                  "note": "separate rocprofv3 --pmc passes; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md HBM "
                          "section); Infinity-Cache hits are counted"}
json.dump(out, open(f'profiles/{tag}_summary.json', 'w'), indent=1)
print(json.dumps(out)[:900])
