# one training step's kernels in launch order from a rocprofv3 kernel trace: start (us), duration (us), name
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'pack_kernel' in r['Kernel_Name']][-1]
t0 = int(rows[idx]['Start_Timestamp'])
brief = len(sys.argv) > 2
tot = {}
for r in rows[idx:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'].replace('ntx_train::', '').replace('void ', '')[:40]
    tot.setdefault(n, []).append(d)
    if not brief: print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f} {n}")
end = (int(rows[-1]['End_Timestamp']) - t0) / 1e3
for n, v in sorted(tot.items(), key=lambda kv: -sum(kv[1])): print(f"{sum(v):9.1f} us  {len(v):3d} x  {n}   [{' '.join(f'{x:.0f}' for x in v[:14])}]")
print(f"step: {end:.1f} us")
