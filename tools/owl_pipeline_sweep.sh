# lock-step group size x alternating groups on the headline workload (same box, back to back); round 5: re-run for the f32x3 mode
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2: %.0f frames/s, %.4f s/video, gemm %.1f TFLOP/s algorithmic, attention %.1f, solo %.3f s, host cpu %.3f s/video' % (d['value'], c['sec_per_video'], r['achieved_algorithmic'], r.get('attention_kernel', {}).get('achieved_algorithmic', 0), c['single_search_alone_latency_sec'], c['host_cpu_sec_per_video']))"; }
run() { python bench.py --lockstep $1 --pipeline $2 --steps $3 --no-cpu-baseline --no-grid4 --no-verify --no-other-configs > /tmp/g4.json 2>/tmp/g4.err || tail -5 /tmp/g4.err; show /tmp/g4.json "lockstep $1 x $2 groups, $3 videos"; }
run 4 2 16
run 8 2 16
run 6 2 12
run 4 3 12
run 4 4 16
run 8 3 24
run 4 2 16
run 2 4 16
