for v in 1 2 4 8 7; do SDFR_LIB=$PWD/build/variants/libsdfr_skip$v.so bash tools/trace_cmd.sh r3y_skip$v tools/profile_decoder_vjp.py > /dev/null 2>&1; python - <<PY
import csv,glob
f=glob.glob('gpurun_out/r3y_skip$v/trace/runc/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'resize3_backward' in r['Kernel_Name']][-3:]
print('skip$v', ['%.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows])
PY
done
