"""Launch timeline of the last bench step from a rocprofv3 kernel trace: per launch duration_us/workgroups + kernel tag."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'rsp::dev::solve' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'solve_clusters' in r['Kernel_Name']]
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
step = rows[idx[-2 * parts]:idx[-1] + 1]
def tag(n):
    if 'solve_items' in n: return 'I' + n.split('solve_items<')[1].split('>')[0].replace(' ', '')[-1]
    if 'clusters' in n: return 'c'
    if 'small' in n: return 's'
    if 'false, 4>' in n: return 'L'
    return 'd'
tot = {}
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[tag(r['Kernel_Name'])] = tot.get(tag(r['Kernel_Name']), 0) + d
print('launches', len(step), 'span_us', (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3, {k: round(v) for k, v in tot.items()}, 'sum', round(sum(tot.values())))
print(' '.join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}/{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}{tag(r['Kernel_Name'])}" for r in step))
