# Board power / clocks while a long run of C3 steps is in flight (developer diagnosis: is the step loop at the power cap?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/power; mkdir -p $O; rm -f $O/s_*.txt
rocm-smi --showmaxpower 2>&1 | grep -i "power" | head -3
(python3 bench.py --steps ${1:-1500} --warmup 20 --no-ttt --no-cpu-baseline --no-probe > $O/bench.json 2> $O/bench.err) &
BP=$!
i=0
while kill -0 $BP 2>/dev/null; do
  i=$((i+1))
  rocm-smi --showpower --showclocks > $O/s_$i.txt 2>&1
  sleep 0.2
done
wait $BP
python3 - <<'PY'
import glob,re,os
rows=[]
for f in sorted(glob.glob('gpurun_out/power/s_*.txt'), key=lambda x:int(re.findall(r's_(\d+)',x)[0])):
    t=open(f).read()
    p=re.findall(r'Power \(W\): ([\d.]+)',t); s=re.findall(r'sclk clock level: \S+ \((\d+)Mhz\)',t)
    if p and s: rows.append((float(p[0]), int(s[0])))
rows.sort(reverse=True)
print('samples', len(rows), 'top by power (W, sclk MHz):', rows[:12])
PY
python3 -c "import json;d=json.load(open('gpurun_out/power/bench.json'));print(d['ms_per_step'], d['value'])"
