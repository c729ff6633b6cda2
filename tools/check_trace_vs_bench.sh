export TMPDIR=/tmp; R=$PWD; cd /tmp
python -m pytest $R/tests/test_gpu_searcher.py $R/tests/test_gpu_sharding.py $R/tests/test_gpu_yolo.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error" | tail -3
rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 8 --warmup 1 --no-cpu-baseline > $R/gpurun_out/chk_bench.json 2> $R/gpurun_out/chk_bench.err
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --timed-region $R/gpurun_out/chk_bench.json --check | tail -6
python $R/tools/rocpd_gaps.py $DB --timed-region $R/gpurun_out/chk_bench.json | head -12
rm -rf /tmp/kt; rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --heuristic yolo --steps 48 --warmup 1 --no-cpu-baseline --no-grid4 > $R/gpurun_out/chk_bench_yolo.json 2> $R/gpurun_out/chk_bench_yolo.err
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB --timed-region $R/gpurun_out/chk_bench_yolo.json --check | tail -6
python -c "
import json
for f in ['chk_bench.json','chk_bench_yolo.json']:
    d=json.loads(open('$R/gpurun_out/'+f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['roofline']['achieved'],1), d['config']['keyframes_verified'])"
