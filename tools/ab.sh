# Same-box A/B of the step time: bash tools/ab.sh "<env assignments for B>" [bench args]
# runs A (default tree) and B (with the given environment) alternately, three times each, and prints ms/step.
ENVB="$1"; shift
ARGS="--steps 30 --warmup 5 --no-cpu-baseline $*"
for i in 1 2 3; do
  a=$(timeout -k 10 200 python bench.py $ARGS | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
  b=$(env $ENVB timeout -k 10 200 python bench.py $ARGS | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
  echo "A(default) $a ms   B($ENVB) $b ms"
done
