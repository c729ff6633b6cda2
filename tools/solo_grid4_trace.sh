# One 4x4 search ALONE under the kernel tracer: per-kernel table and idle gaps inside the search (round 4: where the
# reference-default solo latency goes once the FITPACK fit is native).  Writes gpurun_out/solo_grid4_*.{md,txt,log}
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
cd /tmp
python $ROOT/tools/solo_latency_probe.py 4 > $ROOT/gpurun_out/solo_grid4_probe.log 2>&1
# A/Bs, same box: the speculative forward BEHIND the verification batch (round 5's placement); without it; the plain sequential loop of
# rounds 1-4; the native-f32 mode
( echo "# TSTAR_SPECULATE_BEHIND=1"; TSTAR_SPECULATE_BEHIND=1 python $ROOT/tools/solo_latency_probe.py 4 2>&1 | head -3
  echo "# TSTAR_NO_SPECULATION=1"; TSTAR_NO_SPECULATION=1 python $ROOT/tools/solo_latency_probe.py 4 2>&1 | head -3
  echo "# TSTAR_SOLO_SEQUENTIAL=1"; TSTAR_SOLO_SEQUENTIAL=1 python $ROOT/tools/solo_latency_probe.py 4 2>&1 | head -3
  echo "# weights mode f32"; python $ROOT/tools/solo_latency_probe.py 4 0 f32 2>&1 | head -3 ) >> $ROOT/gpurun_out/solo_grid4_probe.log
rm -rf /tmp/sg4
rocprofv3 --kernel-trace -d /tmp/sg4 -o tr -- python $ROOT/tools/solo_latency_probe.py 4 > /dev/null 2> /tmp/sg4.err || tail -3 /tmp/sg4.err
DB=$(find /tmp/sg4 -name '*.db' | head -1)
python $ROOT/tools/rocpd_stats.py "$DB" --timed-region > $ROOT/gpurun_out/solo_grid4_kernel_stats.md 2>&1
python $ROOT/tools/rocpd_gaps.py "$DB" --timed-region > $ROOT/gpurun_out/solo_grid4_gaps.txt 2>&1
python $ROOT/tools/rocpd_streams.py "$DB" --timed-region > $ROOT/gpurun_out/solo_grid4_streams.txt 2>&1
head -40 $ROOT/gpurun_out/solo_grid4_kernel_stats.md
