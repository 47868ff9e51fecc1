#!/bin/bash
P="--no-probe --no-ttt --no-cpu-baseline --no-sustained --no-full-pass --no-full-run"
for st in 20 80; do
python3 bench.py --steps $st --warmup 5 --dtype f32 --reg 0 $P 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('f32 steps', d['steps'], d['ms_per_step'])"
done
python3 bench.py --steps 20 --warmup 25 --dtype f32 --reg 0 $P 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('f32 warm 25', d['steps'], d['ms_per_step'])"
for la in "--lanes 1 --lookahead 8" "--lanes 2 --lookahead 8" "--lanes 1 --lookahead 16" "--lanes 2 --lookahead 16" "--lanes 2 --lookahead 4"; do
python3 bench.py --steps 256 --warmup 32 --p 100 --rows 10000 $la $P 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print('C2 $la', d['value'], d['ms_per_step'])"
done
