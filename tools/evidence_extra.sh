# Further evidence for the measurement section, in one GPU call (run from the repo root on the GPU box):
#   bash tools/evidence_extra.sh <tag>
# leaves under gpurun_out/: <tag>_per_layer_c3_bf16.txt (per-GEMM-launch times and TFLOP/s of one step, event
# brackets on one stream), <tag>_kernel_stats_{c2,c4,cli}.csv (rocprofv3 --stats of the other configurations),
# <tag>_soak_long.txt (30000 bf16 steps from the seed-3 initialisation).
TAG=${1:-round}
R=$PWD
O=$R/gpurun_out
timeout -k 10 300 python bench.py --no-cpu-baseline --per-layer > $O/${TAG}_per_layer.json 2> $O/${TAG}_per_layer_c3_bf16.txt || exit 1
cd /tmp && export TMPDIR=/tmp
for c in c2 c4 cli; do
  rm -rf $O/prof_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$c -o p --output-format csv -- python $R/bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $O/prof_$c.log 2>&1 || exit 1
  cp $O/prof_$c/p_kernel_stats.csv $O/${TAG}_kernel_stats_$c.csv
done
cd $R
timeout -k 10 600 python tools/soak_train.py 30000 3e-5 bf16 3 > $O/${TAG}_soak_long.txt 2>&1 || exit 1
tail -3 $O/${TAG}_soak_long.txt
