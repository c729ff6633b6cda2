for L in 4 8 16; do for MB in 192 256 384; do
python bench.py --steps 16 --warmup 1 --lockstep $L --max-batch $MB --no-cpu-baseline --no-verify --no-grid4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('lockstep $L max_batch $MB: %.0f frames/s, gemm %.1f TFLOP/s (share %.2f), attn %.1f, %.1f ms/video' % (d['value'], r['achieved'], r['time_share_of_step'], r['attention_f32_kernel']['achieved'], d['ms_per_step']))"
done; done
