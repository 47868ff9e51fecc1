# Board power while the isolated k-loop harness runs (fp64: HBM stream, Infinity Cache, L2 in turn, twice)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/power_h; mkdir -p $O; rm -f $O/s_*.txt
(tools/bin/mfma_bench6 > $O/bench6.log 2>&1) &
BP=$!
i=0
while kill -0 $BP 2>/dev/null; do
  i=$((i+1))
  rocm-smi --showpower --showclocks > $O/s_$i.txt 2>&1
  sleep 0.15
done
wait $BP
python3 - <<'PY'
import glob,re
rows=[]
for f in sorted(glob.glob('gpurun_out/power_h/s_*.txt'), key=lambda x:int(re.findall(r's_(\d+)',x)[0])):
    t=open(f).read()
    p=re.findall(r'Power \(W\): ([\d.]+)',t); s=re.findall(r'sclk clock level: \S+ \((\d+)Mhz\)',t)
    if p and s: rows.append((float(p[0]), int(s[0])))
print('samples in time order (W, sclk MHz):')
print(' '.join(f'{int(p)}/{s}' for p,s in rows))
PY
grep "base " $O/bench6.log | head -6
