for i in 1 2; do
for cfg in "16 32" "31 62" "24 48"; do set -- $cfg
python bench.py --heuristic yolo --steps $2 --warmup 1 --lockstep $1 --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lockstep $1 steps $2: %.0f frames/s, conv %.1f TFLOP/s = %.3f (share %.2f), %.1f ms/video' % (d['value'], r['achieved'], r['frac'], r['time_share_of_step'], d['ms_per_step']))"
done; done
