#!/usr/bin/env python3
"""Launch by launch durations of the last decoder forward + VJP in a trace_cmd.sh run: vjp_calls.py <run dir>."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/trace/runc/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
idx = [i for i, r in enumerate(rows) if 'fc_stack_kernel' in r['Kernel_Name'] or 'fc_stack_batch_kernel' in r['Kernel_Name']][-1]
tot = 0
for r in rows[idx:]:
    n = r['Kernel_Name'].replace('sdfr::(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    dt = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += dt
    print(n.ljust(44), '%8.1f' % dt)
print('sum', round(tot, 1))
