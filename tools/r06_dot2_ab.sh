# Round 6 lab: the exact three-term split with v_dot2c_f32_bf16 remainders against the subtract form -- bit equality over all binades and
# NOTE: attn_lab_d1 / x3_lab_d1 were built from sources carrying a TSTAR_SPLIT_DOT2 switch that was removed again once the result was in
# (git show 3bef2c0^:tstar_amd/csrc/attention_x3.h has it); the script is kept as the record of how profiles/r06_dot2_split_lab.log was made.
# VALU rate (dot2_split_lab), then the two kernels that split in their loops, built both ways: attention_x3 (VALU-issue-bound) and the
# f32x3 GEMM main loop (split in MFMA shadows).  Writes gpurun_out/r06_dot2/*.log
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06_dot2
mkdir -p $OUT
cd $ROOT/tools/lab
./dot2_split_lab > $OUT/dot2_split_lab.log 2>&1
for v in 0 1; do
    ./attn_lab_d$v 256 2>&1 | tail -8 > $OUT/attn_lab_d$v.log
    timeout 300 ./x3_lab_d$v 2>&1 | tail -40 > $OUT/x3_lab_d$v.log
done
tail -4 $OUT/dot2_split_lab.log
for v in 0 1; do echo "== attention split form $v"; grep "in-kernel" $OUT/attn_lab_d$v.log | tail -3; done
for v in 0 1; do echo "== x3 GEMM split form $v"; grep -i "v2" $OUT/x3_lab_d$v.log | head -12; done
