"""Merge a rocprofv3 kernel trace and HIP API trace into one timeline around the n-th launch of a kernel.
usage: timeline.py kernel_trace.csv hip_api_trace.csv <kernel substring> <which occurrence> [us before] [us after]"""
import csv, sys
kt, ht, pat, which = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
before = float(sys.argv[5]) if len(sys.argv) > 5 else 150.0
after = float(sys.argv[6]) if len(sys.argv) > 6 else 700.0
ks = sorted(csv.DictReader(open(kt)), key=lambda r: int(r['Start_Timestamp']))
hs = sorted(csv.DictReader(open(ht)), key=lambda r: int(r['Start_Timestamp']))
hits = [r for r in ks if pat in r['Kernel_Name']]
print('launches of', pat, ':', len(hits))
t0 = int(hits[which]['Start_Timestamp'])
ev = []
for r in ks:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 - before * 1e3 <= s <= t0 + after * 1e3:
        ev.append((s, 'GPU ', r['Kernel_Name'][:60], (e - s) / 1e3))
for r in hs:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 - before * 1e3 <= s <= t0 + after * 1e3:
        ev.append((s, 'host', r['Function'], (e - s) / 1e3))
ev.sort()
for s, who, name, dur in ev:
    print('%10.2f us  %s  %-62s %8.2f us' % ((s - t0) / 1e3, who, name, dur))
