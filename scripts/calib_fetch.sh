#!/bin/bash
# usage (GPU box): scripts/calib_fetch.sh TAG -> gpurun_out/TAG_fetch_calibration.txt
# FETCH_SIZE / WRITE_SIZE of kernels that move a known number of bytes (scripts/calib_fetch.hip), one counter per run.
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/${tag}_calib_$c
  timeout -s KILL 120 rocprofv3 --pmc $c -d $out/${tag}_calib_$c -o p --output-format csv -- $root/scripts/calib_fetch > $out/${tag}_calib_$c.log 2>&1 || { echo "calib $c failed"; tail -3 $out/${tag}_calib_$c.log; exit 1; }
done
python3 - "$out" "$tag" <<'PY'
import sys,csv,glob,collections,re
out,tag=sys.argv[1],sys.argv[2]
G=1<<30
known={"read_x4":("FETCH_SIZE",G),"read_x2":("FETCH_SIZE",G),"read_x1":("FETCH_SIZE",G),"read_b1":("FETCH_SIZE",G//4),
       "gather_x1":("FETCH_SIZE",G//64*64),"gather_local_x1":("FETCH_SIZE",G),
       "write_x4":("WRITE_SIZE",G),"write_x2":("WRITE_SIZE",G),"write_x1":("WRITE_SIZE",G),"scatter_x2":("WRITE_SIZE",G)}
lines=["# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KB) of kernels moving a known byte count (scripts/calib_fetch.hip), MI355X",
       "# kernel counter reported_bytes_per_launch known_bytes reported/known"]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{out}/{tag}_calib_{c}/**/*counter_collection.csv",recursive=True)
    agg=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name']!=c: continue
        k=re.sub(r"\(.*","",r['Kernel_Name'])
        agg[k]+=float(r['Counter_Value']); n[k]+=1
    for k in agg:
        if k in known and known[k][0]==c:
            rep=agg[k]/n[k]*1024
            lines.append("%s %s %d %d %.3f"%(k,c,rep,known[k][1],rep/known[k][1]))
        elif k in known:
            lines.append("%s %s %d (other direction)"%(k,c,agg[k]/n[k]*1024))
open(f"{out}/{tag}_fetch_calibration.txt","w").write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
