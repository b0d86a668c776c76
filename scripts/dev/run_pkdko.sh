# dev (GPU): k_minimizer_pkd variants (scripts/build_variant.sh <tag> "-D..." k_minimizer_pkd) against the shipped library, one session
for rep in 1 2; do
for v in base ${VARIANTS:-dnopipe}; do
  if [ $v = base ]; then unset BSK_LIB; else export BSK_LIB=$PWD/scripts/variants/libbsk_$v.so; fi
  for rl in ${RLS:-400}; do
    echo "== $v $rl $(BSK_NO_RING=1 python scripts/perf_quick.py $((3000000000 / rl)) min 21 ${W:-11} 8 $rl 2>&1 | grep -E "Gbases|plan:" | tr '\n' ' ')"
  done
done
done
