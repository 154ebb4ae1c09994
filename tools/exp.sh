for c in 32 64 128 256; do
timeout -k 10 100 python tools/layer_bench.py 256 64 $c 64 3 1 fwd 30 || exit 1
done
timeout -k 10 100 python tools/layer_bench.py 256 32 128 128 3 1 fwd 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 16 256 256 3 1 fwd 30 || exit 1
