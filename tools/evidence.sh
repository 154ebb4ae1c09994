# Bench lines and logs of every configuration, in one GPU call (run from the repo root on the GPU box):
#   bash tools/evidence.sh <tag>
# leaves under gpurun_out/: <tag>_bench_<config>.json for c3 (the default line, CPU baseline included), c2, c4,
# c4d and c3 in f32; <tag>_infer_c5.txt (eval forward at 128 x 128, batch 1024); <tag>_soak.txt (6000 bf16 steps from the
# seed-3 initialisation, which leaves the mean-image plateau, then 3000 from seed 0, which does not).
TAG=${1:-round}
O=gpurun_out
timeout -k 10 400 python bench.py > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err || exit 1
for c in c2 c4 c4d; do
  timeout -k 10 200 python bench.py --config $c --no-cpu-baseline > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err || exit 1
done
timeout -k 10 200 python bench.py --dtype f32 --no-cpu-baseline > $O/${TAG}_bench_c3_f32.json 2> $O/${TAG}_bench_c3_f32.err || exit 1
timeout -k 10 200 python tools/infer_bench.py 128 100 1024 10 > $O/${TAG}_infer_c5.txt 2>&1 || exit 1
timeout -k 10 300 python tools/soak_train.py 6000 3e-5 bf16 3 > $O/${TAG}_soak.txt 2>&1 || exit 1
timeout -k 10 300 python tools/soak_train.py 3000 3e-5 bf16 0 >> $O/${TAG}_soak.txt 2>&1 || exit 1
for f in $O/${TAG}_bench_*.json; do cut -c1-200 $f; done; cat $O/${TAG}_infer_c5.txt; grep "^ok" $O/${TAG}_soak.txt
