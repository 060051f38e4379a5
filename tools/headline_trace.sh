cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/htrace
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/htrace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/htrace'
rows=[]
for f in glob.glob(root+'/**/*kernel_trace.csv', recursive=True):
    rows+=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]
prev_end=None
for r in rows:
    n=r['Kernel_Name']
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    tag='E' if 'estep_mfma4' in n else ('M' if 'mstats_wide' in n else None)
    if tag: out.append('%s %.3f gap %.1f'%(tag,(e-s)/1e6,((s-prev_end)/1e3 if prev_end else 0)))
    prev_end=e
open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/htrace_seq.txt','w').write('\n'.join(out)+'\n')
print('\n'.join(out))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/htrace
