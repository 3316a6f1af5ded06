"""Condenses a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats file into a short table
(kept under profiles/).  usage: python tools/summarize_prof.py <kernel_stats.csv> <out.txt> [title]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else src
rows = list(csv.DictReader(open(src)))
with open(dst, 'w') as fh:
    fh.write(f'# {title}\n# source: rocprofv3 --kernel-trace --stats --output-format csv ; columns: calls, avg_us, total_ms, pct, kernel\n')
    for r in rows[:40]:
        name = r['Name']
        for junk in ('void ', '(anonymous namespace)::', 'at::native::'):
            name = name.replace(junk, '')
        fh.write(f"{int(r['Calls']):6d} {float(r['AverageNs'])/1e3:12.2f} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['Percentage']):6.2f}  {name[:110]}\n")
print(open(dst).read()[:1500])
