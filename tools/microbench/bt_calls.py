#!/usr/bin/env python3
"""Durations of the transposed-resize launches of the last decoder VJP in a trace_cmd.sh run: bt_calls.py <run dir>."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/trace/runc/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'resize3_backward' in r['Kernel_Name']][-3:]
print(sys.argv[1], ['%.1f us grid %s x %s lds %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
      int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Workgroup_Size_X'], r['LDS_Block_Size']) for r in rows])
