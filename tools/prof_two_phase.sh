cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof2p -- python3 bench.py --two-phase --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof2p.json 2> gpurun_out/prof2p.err
python3 - <<'PY'
import csv, glob, re
f=glob.glob('gpurun_out/prof2p/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if r['Kernel_Name'].startswith('sgd_momentum')]
# pick a replayed step in the middle
a,b=idx[8]+1, idx[9]+1
sub=rows[a:b+4]
t0=int(sub[0]['Start_Timestamp'])
prev_end=None
print("kernels in step", len(sub))
for r in sub:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    gap = (s-prev_end) if prev_end is not None else 0
    n=re.sub(r'\(.*','',r['Kernel_Name'].replace('void ',''))[:60]
    if gap>3000 or 'nccl' in n.lower() or 'rccl' in n.lower() or (e-s)>150000:
        print("%9.1f us  gap %7.1f  dur %7.1f  q=%s  %s" % ((s-t0)/1e3, gap/1e3, (e-s)/1e3, r['Queue_Id'], n))
    prev_end=max(prev_end or 0, e)
print("step span ms", (int(sub[-5]['End_Timestamp'])-t0)/1e6)
PY
find gpurun_out/prof2p -name "*.csv" -size +5M -delete
