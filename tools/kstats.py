"""Print the top rows of a rocprofv3 kernel_stats.csv compactly: tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:n]:
    name = r['Name'].replace('(anonymous namespace)::', '').split('(')[0][-70:]
    print('%-70s calls %5s avg %9.1f us  %s%%' % (name, r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
