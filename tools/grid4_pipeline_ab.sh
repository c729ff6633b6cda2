# the reference-default 4x4 grid: one lock-step group of 16 against alternating groups (2 x 8, 2 x 16)
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2: %.0f frames/s, %.3f s/video, gemm %.1f TFLOP/s' % (d['value'], c['sec_per_video'], r['achieved']))"; }
run() { python bench.py --grid 4 --lockstep $1 --pipeline $2 --steps $3 --warmup 1 --no-cpu-baseline --no-grid4 --no-verify > /tmp/g4.json 2>/tmp/g4.err || tail -5 /tmp/g4.err; show /tmp/g4.json "lockstep $1 x $2 groups, $3 videos"; }
run 16 1 16
run 8 2 16
run 16 2 32
run 31 2 62
run 16 1 16
