for e in 0 1; do
if [ $e = 1 ]; then export CVG_PATCH_PERSIST=1; else unset CVG_PATCH_PERSIST; fi
echo "PERSIST=$e"
timeout -k 10 100 python tools/layer_bench.py 256 64 32 64 3 1 fwd 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 64 64 64 3 1 fwd 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 64 128 64 3 0 fwd 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 32 128 128 3 1 fwd 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 16 256 256 3 1 fwd 30 || exit 1
timeout -k 10 100 python tools/layer_bench.py 256 66 64 128 3 1 fwd 30 || exit 1
done
