import os, csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
tot = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    nm = r['Kernel_Name']
    if 'refill' in nm or 'chain_' in nm:
        k = nm.split('(')[0].split('::')[-1]
        tot[k][0] += 1; tot[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print({k: (v[0], round(v[1] / v[0], 1)) for k, v in tot.items()})
