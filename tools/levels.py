"""Per-launch durations of the last bench step from a rocprofv3 kernel trace (csv)."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2])  # launches per step
rows = [r for r in csv.DictReader(open(f)) if sys.argv[3] in r['Kernel_Name']]
last = rows[-n:]
tot = 0
out = []
for i, r in enumerate(last):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    out.append(f"{i}:{int(r['Grid_Size_X'])//64}w:{r['Kernel_Name'].split('<')[1][0]}:{d:.0f}us")
print(' '.join(out))
print('sum us', tot, 'span us', (int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3)
