# Step time for several values of one environment variable, interleaved, same box: bash tools/ab_sites.sh VAR v1 v2 ...
VAR=$1; shift
for rep in 1 2; do
  for v in "$@"; do
    t=$(env $VAR=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "$VAR=$v  $t ms"
  done
done
