show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2: %.0f frames/s, %.4f s/video, gemm %.1f alg (frac %.3f), attention %.1f, verify/video %.1f' % (d['value'], c['sec_per_video'], r['achieved_algorithmic'], r['frac'], r['attention_kernel']['achieved_algorithmic'], c['verify_calls_per_video']))"; }
run() { python bench.py --lockstep $1 --pipeline $2 --steps $3 --max-batch $4 --no-cpu-baseline --no-grid4 --no-verify --no-other-configs > /tmp/g4.json 2>/tmp/g4.err || tail -5 /tmp/g4.err; show /tmp/g4.json "lockstep $1 x $2 groups, $3 videos, max-batch $4"; }
run 8 2 16 256
run 8 2 16 384
run 8 2 16 512
run 12 2 24 512
run 16 2 32 768
run 8 2 16 256
run 5 2 20 256
run 10 2 20 512
