for e in 0 1; do
if [ $e = 1 ]; then export CVG_WGRAD_4W=1; else unset CVG_WGRAD_4W; fi
echo "4W=$e"
timeout -k 10 100 python tools/layer_bench.py 256 8 512 512 3 1 wgrad 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 16 256 256 3 1 wgrad 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 32 128 128 3 1 wgrad 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 32 256 128 3 0 wgrad 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 64 64 64 3 1 wgrad 30 || exit 1
done
