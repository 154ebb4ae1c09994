# Scratch A/B timing of a few layers.
for w in 0 2 1; do
export CVG_PATCH_WIDE=$w
echo "WIDE=$w"
for a in "256 32 128 128 3 1" "256 16 256 256 3 1" "256 8 512 512 3 1" "256 66 64 128 3 1" "256 34 128 256 3 1"; do
timeout -k 10 100 python tools/layer_bench.py $a fwd 30 || exit 1
done
done
