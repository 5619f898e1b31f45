cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; R=$PWD; O=gpurun_out/prof_wgrad; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/a -o w -- python3 $R/scripts/perf_wgrad.py > $R/$O/w.log 2>&1 )
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/a/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# sequence of (name, dur, grid)
seq=[(r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y")) for r in rows if "wgrad" in r["Kernel_Name"]]
agg=collections.OrderedDict()
for n,d,gx,gy in seq:
    k=(n,gx,gy); agg.setdefault(k,[]).append(d)
for k,v in agg.items(): print(k, len(v), round(sum(v)/len(v),1))
PY
rm -rf $O/a
