show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2: %.0f frames/s, conv %.1f TFLOP/s (share %.2f), %.1f ms/video' % (d['value'], r['achieved'], r['time_share_of_step'], d['ms_per_step']))"; }
run() { python bench.py --heuristic yolo --lockstep $1 --pipeline $2 --steps $3 --yolo-max-batch ${4:-76} --no-cpu-baseline --no-grid4 --no-verify > /tmp/y.json 2>/tmp/y.err || tail -5 /tmp/y.err; show /tmp/y.json "lockstep $1 x $2 groups, $3 videos, chunk ${4:-76}"; }
run 24 2 48
run 16 2 64
run 31 2 62
run 16 3 48
run 12 4 48
run 24 1 48
run 24 2 48
