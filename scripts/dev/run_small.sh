# dev (GPU): small batches (a pipeline chunk, a class plan's part): kernel time against the batch size
for n in ${NS:-20000 100000 300000 1000000 3000000}; do
  for cfg in "min 21 11 150" "min 21 11 400" "syn 31 11 150"; do
    set -- $cfg
    echo "== n=$n $1 k=$2 x=$3 $4 bp: $(python scripts/perf_quick.py $n $1 $2 $3 6 $4 2>&1 | grep -E "kernel ms|Gbases|plan:" | tr '\n' ' ' | cut -c1-220)"
  done
done
