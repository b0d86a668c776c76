# dev (GPU): rates off the tuned parameter points -- looking for plans that fall off a cliff.  CFGS="kind k x;..." RLS="..."
IFS=';' read -ra CF <<< "${CFGS:-syn 21 11;syn 25 15;syn 15 9;min 31 15;min 21 20;min 21 28}"
for cfg in "${CF[@]}"; do
  set -- $cfg
  for rl in ${RLS:-150 250 400 700 1500 4000}; do
    n=$((2000000000 / rl))
    echo "== $1 k=$2 x=$3 $rl bp: $(python scripts/perf_quick.py $n $1 $2 $3 3 $rl 2>&1 | grep -E "Gbases|plan:" | tr '\n' ' ' | cut -c1-160)"
  done
done
