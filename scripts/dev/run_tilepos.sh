# dev (GPU): long sequences cut into tiles (scripts/dev/perf_long2.py) over the tile size (BSK_TILE_POS; 0 = the planner's choice) and the batch size
for cfg in "2e9 400" "2e8 40" "2e7 4"; do
set -- $cfg; export TOTAL=$1 NSEQ=$2
for tp in ${TPS:-0 96 256 512 1024 2048}; do
  if [ $tp = 0 ]; then unset BSK_TILE_POS; else export BSK_TILE_POS=$tp; fi
  echo "== total $TOTAL tile_pos $tp: $(python scripts/dev/perf_long2.py 2>&1 | grep minimizer | cut -c20-200)"
done
done
