"""Per-launch durations of the LAST bench step from a rocprofv3 --kernel-trace run.  usage: r2_trace.py <dir> <launches per step>"""
import csv, glob, os, sys, re
d = sys.argv[1]
f = sorted(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'rsp::dev' in r['Kernel_Name'] and 'solve' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
per = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-per:]
t0 = int(rows[0]['Start_Timestamp'])
tot = 0
for i, r in enumerate(rows):
    name = re.sub(r'rsp::dev::', '', r['Kernel_Name'])
    name = re.sub(r'\(.*', '', name)[:40]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    print(f"{i:4d} {name:40s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):7d} wg {r['Workgroup_Size_X']:>4s} vgpr {r.get('VGPR_Count','?'):>4s} lds {r.get('LDS_Block_Size','?'):>6s} stream {r.get('Stream_Id','?'):>3s} start_us {(s-t0)/1e3:9.1f} dur_us {(e-s)/1e3:8.1f}")
print(f"sum of durations {tot/1e3:.1f} us, span {(int(rows[-1]['End_Timestamp'])-t0)/1e3:.1f} us")
