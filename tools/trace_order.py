"""Kernel launches of the LAST step in a rocprofv3 --kernel-trace csv, in start order: name, duration, gap to the previous kernel's end.
  python tools/trace_order.py <kernel_trace.csv> <anchor kernel name substring> [n_last]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
anchor = sys.argv[2]
idx = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 2
a, b = idx[-n_last], idx[-n_last + 1] if n_last > 1 else len(rows)
prev_end = None
tot = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    tot += (e - s)
    print('%8.2f us  gap %7.2f  %s' % ((e - s) / 1e3, gap, r['Kernel_Name'][:110]))
    prev_end = max(prev_end or 0, e)
print('launches %d  kernel time %.1f us  span %.1f us' % (b - a, tot / 1e3, (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3))
