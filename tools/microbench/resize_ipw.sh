# (needs the SDFR_RESIZE_IPW experiment hook in decoder.hip's resize launch: see git history)
for c in 1 2 4 8 16; do
  SDFR_RESIZE_IPW=$c bash tools/trace_cmd.sh r04k/ipw$c tools/profile_decoder.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r04k/ipw$c/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "resize3_tiled" in n: acc[n[40:70]+" grid "+r.get("Grid_Size_X","?")+"x"+r.get("Grid_Size_Y","?")].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in acc.items(): print("ipw cap $c", k, len(v), round(sum(v)/len(v)/1e3,1), round(min(v)/1e3,1))
PY
done
