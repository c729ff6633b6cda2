show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2: %.0f frames/s, %.4f s/video, gemm %.1f TFLOP/s, solo %.3f s' % (d['value'], c['sec_per_video'], r['achieved'], c['single_search_alone_latency_sec']))"; }
run() { python bench.py --lockstep $1 --pipeline $2 --steps $3 --no-cpu-baseline --no-grid4 --no-verify > /tmp/g4.json 2>/tmp/g4.err || tail -5 /tmp/g4.err; show /tmp/g4.json "lockstep $1 x $2 groups, $3 videos"; }
run 4 2 8
run 4 2 16
run 8 2 16
run 4 3 12
run 4 4 16
run 4 2 8
run 2 4 8
