# Bench lines of every configuration plus the data-parallel rehearsals, in one GPU call (from the repo root):
#   bash tools/evidence_round3.sh <tag>
# leaves under gpurun_out/: <tag>_bench_{c2,c4,c4d,c5,c3_f32,cli}.json, <tag>_bench_c3_forced_ddp.json (four graphs
# with the exchange hooks between them, one rank, no process group), <tag>_bench_c3_one_rank_rccl.json (the same
# with a one-rank RCCL group doing the all-reduces), <tag>_bench_c3_two_ranks_gloo_one_gpu.json (two processes
# sharing this GPU, host-side reduction), <tag>_soak.txt.
TAG=${1:-round}
O=gpurun_out
for c in c2 c4 c4d c5 cli; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > $O/${TAG}_bench_$c.json 2> $O/${TAG}_bench_$c.err || exit 1
done
timeout -k 10 200 python bench.py --dtype f32 --no-cpu-baseline > $O/${TAG}_bench_c3_f32.json 2> $O/${TAG}_bench_c3_f32.err || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/${TAG}_bench_c3_same_box.json 2> $O/${TAG}_bench_c3_same_box.err || exit 1
CVG_BENCH_FORCE_DDP=1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/${TAG}_bench_c3_forced_ddp.json 2> $O/${TAG}_bench_c3_forced_ddp.err || exit 1
CVG_BENCH_FORCE_DDP=nccl timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/${TAG}_bench_c3_one_rank_rccl.json 2> $O/${TAG}_bench_c3_one_rank_rccl.err || exit 1
CVG_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --no-cpu-baseline --steps 10 --warmup 3 > $O/${TAG}_bench_c3_two_ranks_gloo_one_gpu.json 2> $O/${TAG}_bench_c3_two_ranks_gloo_one_gpu.err || exit 1
timeout -k 10 300 python tools/soak_train.py 6000 3e-5 bf16 3 > $O/${TAG}_soak.txt 2>&1 || exit 1
for f in $O/${TAG}_bench_*.json; do echo $f; tail -1 $f | cut -c1-170; done; grep "^ok" $O/${TAG}_soak.txt
