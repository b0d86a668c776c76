for rep in 1 2; do
for v in base dround1 dnoflush; do
  if [ $v = base ]; then unset BSK_LIB; else export BSK_LIB=$PWD/scripts/variants/libbsk_$v.so; fi
  for rl in 400; do
    echo "== $v $rl $(python scripts/perf_quick.py 7500000 min 21 11 8 $rl 2>&1 | grep -E "Gbases|plan:" | tr '\n' ' ')"
  done
done
done
