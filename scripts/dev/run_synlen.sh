# dev (GPU): syncmers k=31 s=11 over the read length, 3 10^9 bases per point (kernel ms = bsk_sketch_timed's events around everything a call launches)
for rl in ${RLS:-380 392 400 420 448 500 1000 3000}; do
  n=$((3000000000 / rl))
  echo "== syn $rl $(python scripts/perf_quick.py $n syn 31 11 4 $rl 2>&1 | grep -E "Gbases|plan:" | tr '\n' ' ' | cut -c1-200)"
done
