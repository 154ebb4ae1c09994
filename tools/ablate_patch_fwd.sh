# Timing-only ablations of patch_fwd's main loop (results are wrong on purpose): builds libraries without the weight
# reloads (NOW), without the LDS reads of the patch (NOX), without the patch prefetch (NOP), without the chunk barrier
# (NOBAR) and without all three operand streams (ALL) next to the package's library; run them on the GPU box with
#   for v in NOW NOX NOP NOBAR ALL; do CVG_LIB=cvae_gan_amd/lib/lib_exp_$v.so python tools/layer_bench.py 256 8 512 512 3 1 fwd; done
# and remove cvae_gan_amd/lib/lib_exp_*.so afterwards.   bash tools/ablate_patch_fwd.sh   (here, on the CPU)
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Iinclude"
S=cvae_gan_amd/csrc
mkdir -p build_exp
for v in NOW NOX NOP NOBAR ALL; do
  D="-DEXP_$v"; [ $v = ALL ] && D="-DEXP_NOW -DEXP_NOX -DEXP_NOP"
  /opt/rocm/bin/hipcc $F $D -c $S/conv_patch.hip -o build_exp/cp_$v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o cvae_gan_amd/lib/lib_exp_$v.so $S/api.o $S/igemm.o build_exp/cp_$v.o $S/elementwise.o || exit 1
done
