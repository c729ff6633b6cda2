for L in ${LOCKSTEPS:-8 16 31}; do for MB in ${BATCHES:-32 38 76}; do
python bench.py --heuristic yolo --steps 32 --warmup 1 --lockstep $L --yolo-max-batch $MB --no-cpu-baseline --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lockstep $L max_batch $MB: %.0f frames/s, conv %.1f TFLOP/s (share %.2f), %.1f ms/video' % (d['value'], r['achieved'], r['time_share_of_step'], d['ms_per_step']))"
done; done
