"""One step of the DIB-R bench as a timeline from a rocprofv3 kernel trace (tools/round3/r03_trace.sh):
kernel, start offset, duration, gap to the previous kernel's end on the device (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# steps start at pv_forward_kernel; take the last complete step
starts = [i for i, n in enumerate(names) if 'pv_forward_kernel' in n]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {(s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:80]}")
    prev_end = max(prev_end, e)
    tot += e - s
print('step span', (int(rows[b]['Start_Timestamp']) - t0) / 1e3, 'us; kernel time sum', tot / 1e3)
