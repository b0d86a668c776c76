for t in r38 r47 r59 r44n32; do
  echo "== $t"
  BSK_LIB=$PWD/scripts/variants/libbsk_$t.so python scripts/dev/perf_syn_long.py 31,11,225 31,11,250 31,11,275 31,11,300 31,11,325 31,11,350 31,11,400 2>&1 | sed 's/ck[0-9]* tie[0-9]*//g'
done
