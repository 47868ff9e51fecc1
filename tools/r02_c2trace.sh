set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_c2trace; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/la8 -o c2 -- python3 bench.py --steps 16 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 --lookahead 8 > $O/la8.json 2> $O/la8.err
rocprofv3 --kernel-trace --output-format csv -d $O/la1 -o c2 -- python3 bench.py --steps 16 --warmup 8 --no-ttt --no-cpu-baseline --no-probe --p 100 --rows 10000 --lookahead 1 > $O/la1.json 2> $O/la1.err
python3 - <<'PY'
import csv,re
for tag in ('la8','la1'):
    rows=list(csv.DictReader(open(f'gpurun_out/r02_c2trace/{tag}/c2_kernel_trace.csv')))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    def short(n): return re.sub(r'\(.*','',n).replace('void ','').replace('lsspa::','')[:26]
    idx=[i for i,r in enumerate(rows) if 'small_p' in r['Kernel_Name']]
    # pass 1 region: find a window in the timed pass (first half of the small_p launches after warmup)
    k=idx[2] if tag=='la8' else idx[12]
    print('==',tag)
    for r0,r1 in zip(rows[k:k+22],rows[k+1:k+23]):
        print(f"  {short(r0['Kernel_Name']):26s} dur {(int(r0['End_Timestamp'])-int(r0['Start_Timestamp']))/1e3:7.1f} us  gap_after {(int(r1['Start_Timestamp'])-int(r0['End_Timestamp']))/1e3:6.1f}")
PY
