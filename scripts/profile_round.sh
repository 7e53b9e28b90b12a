#!/bin/bash
# usage (GPU box): scripts/profile_round.sh TAG   -> gpurun_out/TAG_{bench.json,kernel_stats.csv,kernel_stats_batch.csv,pmc_hbm_traffic.csv}
# One bench run for the JSON line, one rocprofv3 --kernel-trace --stats run, two --pmc runs (FETCH_SIZE, WRITE_SIZE).
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $root/bench.py --steps 5 --warmup 1 > $out/${tag}_bench.log 2>&1 || { echo "bench failed"; tail -5 $out/${tag}_bench.log; exit 1; }
tail -1 $out/${tag}_bench.log > $out/${tag}_bench.json
rm -rf $out/${tag}_trace
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --overlap 1 --stage-steps 1 --no-cpu --no-stream --no-latency --pipe-steps 0 > $out/${tag}_trace.log 2>&1 || { echo "trace failed"; exit 1; }
cp $(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
# the same statistics over the launches of the whole batch only (rocprofv3's own summary also averages in the
# one-segment launches that index the dictionary at context creation)
python3 - "$(find $out/${tag}_trace -name "*kernel_trace.csv" | head -1)" "$out/${tag}_kernel_stats_batch.csv" <<'PY'
import sys,csv,collections,re
rows=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")
    if not k.startswith("lzss::"): continue
    g=int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])*int(r["Grid_Size_Z"])
    rows[k].append((g,int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
with open(sys.argv[2],"w") as f:
    f.write("# launches with the kernel's largest grid (= the whole batch), from rocprofv3 --kernel-trace\n")
    f.write("kernel,batch_launches,avg_ns,min_ns,max_ns,other_launches\n")
    tot={k:sum(d for g,d in v if g==max(x for x,_ in v)) for k,v in rows.items()}
    for k in sorted(rows,key=lambda k:-tot[k]):
        gm=max(g for g,_ in rows[k]); d=[t for g,t in rows[k] if g==gm]
        f.write("%s,%d,%d,%d,%d,%d\n"%(k,len(d),sum(d)/len(d),min(d),max(d),len(rows[k])-len(d)))
PY
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/${tag}_pmc_$c
  timeout -s KILL 200 rocprofv3 --pmc $c -d $out/${tag}_pmc_$c -o p --output-format csv -- python3 $root/bench.py --steps 1 --warmup 0 --overlap 1 --stage-steps 1 --no-cpu --no-stream --no-latency --pipe-steps 0 > $out/${tag}_pmc_$c.log 2>&1 || { echo "pmc $c failed"; exit 1; }
done
python3 - "$out" "$tag" <<'PY'
import sys,csv,glob,collections,re
out,tag=sys.argv[1],sys.argv[2]
rows=[]
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"{out}/{tag}_pmc_{c}/**/*counter_collection.csv",recursive=True)
    agg=collections.defaultdict(float); disp=collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])):
        if r['Counter_Name']!=c: continue
        k=re.sub(r"\(.*","",r['Kernel_Name'])
        agg[k]+=float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
    for k in sorted(agg): rows.append((k,c,int(agg[k]),len(disp[k])))
with open(f"{out}/{tag}_pmc_hbm_traffic.csv","w") as g:
    g.write("# blobs=%s\n" % (sys.argv[3] if len(sys.argv) > 3 else "10000"))
    g.write("# batch_launches=2\n")   # --steps 1 --warmup 0 --stage-steps 1 --overlap 1: two launches of the whole batch
    g.write("kernel,counter,sum_value_KB,dispatches\n")
    for r in rows: g.write("%s,%s,%d,%d\n"%r)
print("ok", len(rows))
PY
