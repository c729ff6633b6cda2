set -u
python -m pytest tests/test_gpu_searcher.py -m gpu -q -x -k "lockstep or alternating" 2>&1 | tail -5
python -m pytest tests/test_gpu_errors.py tests/test_gpu_detector.py -m gpu -q -x 2>&1 | tail -3
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$1', round(d['value']), round(r['achieved'],1), round(d['ms_per_step'],1), d['config'].get('keyframes_verified'), d['config']['keyframes_rank0_step0'])"; }
for P in 1 2 1 2; do
python bench.py --pipeline $P --no-cpu-baseline --no-grid4 > gpurun_out/ab_owl_p$P.json 2> gpurun_out/ab_owl_p$P.err; show gpurun_out/ab_owl_p$P.json
done
for P in 1 2; do
python bench.py --pipeline $P --heuristic yolo --steps 48 --no-cpu-baseline --no-grid4 > gpurun_out/ab_yolo_p$P.json 2> gpurun_out/ab_yolo_p$P.err; show gpurun_out/ab_yolo_p$P.json
python bench.py --pipeline $P --steps 8 --warmup 1 --weights bf16 --nframes 14400 --grid 15 --search-nframes 32 --no-cpu-baseline --no-grid4 > gpurun_out/ab_c5_p$P.json 2> gpurun_out/ab_c5_p$P.err; show gpurun_out/ab_c5_p$P.json
done
