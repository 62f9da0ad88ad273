"""Summarise a rocprofv3 kernel_trace.csv: busy time, idle gaps, per-kernel gap before it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
wall = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('kernels', len(rows), 'busy us', busy / 1e3, 'wall us', wall / 1e3, 'idle frac', 1 - busy / wall)
# the update loop alone: gaps above 200 us are the host between loops (prior sampling, resets, allocation, start-up)
loop_gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows, rows[1:])]
small = sum(g for g in loop_gaps if 0 < g <= 200_000)
print('inside the loops (gaps <= 200 us only): idle us', small / 1e3, 'idle frac', small / (busy + small))
gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gap[(a['Kernel_Name'][:28], b['Kernel_Name'][:28])].append(int(b['Start_Timestamp']) - int(a['End_Timestamp']))
tot = sorted(gap.items(), key=lambda kv: -sum(kv[1]))
for (a, b), g in tot[:25]:
    gs = sorted(g)
    print('%-30s -> %-30s n=%4d  mean gap %7.2f us  median %7.2f  total %8.1f us' % (a, b, len(g), sum(g) / len(g) / 1e3,
                                                                               gs[len(gs) // 2] / 1e3, sum(g) / 1e3))
