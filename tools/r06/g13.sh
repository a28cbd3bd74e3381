cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06n; mkdir -p $O
FS="--force-shard --single-layout --no-fullsort --no-cpu-baseline --no-config-legs --no-e2e --no-ingest --steps 20 --warmup 5"
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py $FS --shard row --no-map > /dev/null 2> $O/trace.err; echo rc=$?
ls $O/trace/*
python - <<'P'
import csv,glob,os
O=os.environ.get('GRAFT_REPO_ROOT')+'/gpurun_out/r06n/trace/'
f=glob.glob(O+'*memory_copy_trace.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    print(len(rows), rows[0].keys())
    import collections
    by=collections.Counter(); tot=collections.Counter()
    for r in rows:
        k=(r.get('Direction'), int(r.get('Bytes',0))//1024)
        by[k]+=1; tot[k]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    for k,c in sorted(by.items(), key=lambda x:-tot[x[0]])[:25]: print(k, c, round(tot[k]/1e3,1),'us total')
P
find $O/trace -name "*kernel_trace.csv" -delete; find $O/trace -name "*.db" -delete
