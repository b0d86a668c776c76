for rl in 200 300 400 1000; do
  n=$((3000000000 / rl))
  echo "== $rl $(python scripts/perf_quick.py $n min 21 11 6 $rl 2>&1 | grep -E "Gbases|plan:|checksum" | tr '\n' ' ' | cut -c1-260)"
done
echo "== w=8 400 $(BSK_NO_RING=1 python scripts/perf_quick.py 7500000 min 21 8 6 400 2>&1 | grep -E "Gbases|plan:" | tr '\n' ' ')"
echo "== w=13 400 $(BSK_NO_RING=1 python scripts/perf_quick.py 7500000 min 21 13 6 400 2>&1 | grep -E "Gbases|plan:" | tr '\n' ' ')"
