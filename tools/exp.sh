# Scratch: phase timing of patch_fwd from the instrumented experiment build, then the product build.
for lib in build_exp/lib_T.so; do
if [ -n "$lib" ]; then export CVG_LIB=$PWD/$lib; else unset CVG_LIB; fi
for a in "256 32 128 128 3 1" "256 16 256 256 3 1" "256 64 64 64 3 1" "256 66 64 128 3 1"; do
timeout -k 10 100 python tools/layer_bench.py $a fwd 20 || exit 1
done
done
