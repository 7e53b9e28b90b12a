#!/bin/bash
# usage (GPU box): scripts/sq_counters.sh TAG [BLOBS] -> gpurun_out/TAG_sq_counters.csv
# Per-kernel SQ / TA counters of one bench step (kernels serialized), one rocprofv3 --pmc run per counter group
# (at most 8 SQ and 2 TA counters per run).  Values are per launch, summed over the chip's shader engines;
# SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, cycle constants).
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
tag=$1
blobs=${2:-1250}
cd /tmp && export TMPDIR=/tmp
groups=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
 "TA_BUSY_avr GRBM_GUI_ACTIVE"
)
# a heartbeat on stdout while a counter run is silent (a run without output for 3 minutes is taken to be hung)
( while sleep 30; do echo "[sq_counters] $(date +%T) still running"; done ) &
hb=$!
trap 'kill $hb 2>/dev/null' EXIT
g=${SQ_FIRST_GROUP:-0}
failed=0
for grp in "${groups[@]:$g}"; do
  echo "[sq_counters] group $g: $grp"
  rm -rf $out/${tag}_sq_$g
  timeout -s KILL 300 rocprofv3 --pmc $grp -d $out/${tag}_sq_$g -o p --output-format csv -- python3 $root/bench.py --blobs $blobs --steps 1 --warmup 1 --overlap 1 --stage-steps 1 --no-cpu --no-stream --no-latency --pipe-steps 0 > $out/${tag}_sq_$g.log 2>&1
  rc=$?
  # (after a kill nothing more is started on the GPU in this call; the summary below is host work)
  if [ $rc -ne 0 ]; then echo "group $g ($grp) failed rc=$rc"; tail -3 $out/${tag}_sq_$g.log; if [ $rc -ge 124 ]; then failed=1; break; fi; fi
  g=$((g+1))
done
python3 - "$out" "$tag" "$blobs" <<'PY'
import sys,csv,glob,collections,re
out,tag,blobs=sys.argv[1:4]
rows=collections.defaultdict(list); names=[]
for f in sorted(glob.glob(f"{out}/{tag}_sq_*/**/*counter_collection.csv",recursive=True)):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(.*","",r['Kernel_Name']).replace("void ","")
        c=r['Counter_Name']
        if c not in names: names.append(c)
        rows[(k,c)].append((int(r['Grid_Size']),float(r['Counter_Value'])))
kernels=sorted({k for k,_ in rows if k.startswith("lzss::")})
with open(f"{out}/{tag}_sq_counters.csv","w") as g:
    g.write(f"# blobs={blobs}; per launch of the whole batch (the launches with the kernel's largest grid; the dictionary's and other one-segment launches are left out); rocprofv3 --pmc, one run per group\n")
    g.write("kernel,batch_launches,"+",".join(names)+"\n")
    for k in kernels:
        vals=[]; nl=0
        for c in names:
            v=rows.get((k,c),[])
            if not v: vals.append(""); continue
            gm=max(x for x,_ in v); sel=[y for x,y in v if x==gm]; nl=len(sel)
            vals.append("%.6g"%(sum(sel)/len(sel)))
        g.write(k+","+str(nl)+","+",".join(vals)+"\n")
print(open(f"{out}/{tag}_sq_counters.csv").read())
PY
exit $failed
