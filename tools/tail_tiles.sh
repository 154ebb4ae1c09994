# Launch time of the tail's forward shape and the stem's image gradient for a list of tile overrides (th,tw,tb):
#   bash tools/tail_tiles.sh 64,1,4 24,5,2 ...
for t in "$@"; do
  echo "tile $t"
  CVG_PATCH_TILE=$t timeout -k 10 100 python tools/tail_timing.py || exit 1
done
