#!/bin/bash
# usage (GPU box): scripts/pmc.sh NAME "COUNTER COUNTER ..."   -> gpurun_out/pmc_NAME/ + per-kernel summary on stdout
cd /tmp && export TMPDIR=/tmp
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$name
rm -rf $out
timeout -s KILL 150 rocprofv3 --pmc $1 -d $out -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --overlap 1 --stage-steps 1 --no-cpu --no-stream --no-latency --pipe-steps 0 > $out.log 2>&1
python3 - "$out" <<'PY'
import sys,csv,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
if not f: print('no csv'); sys.exit(0)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r['Kernel_Name'][:40]; agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    cnt[(k,r['Counter_Name'])]+=1
for k in agg:
    print(k, {c: '%.4g'%(v/cnt[(k,c)]) for c,v in agg[k].items()})
PY
