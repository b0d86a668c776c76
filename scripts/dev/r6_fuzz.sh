#!/bin/bash
# round 6: the fuzz scripts on the round's library (GPU box) -> gpurun_out/r6_fuzz.txt
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_fuzz.txt
: > $O
run() { echo "-- $*" >> $O; ( time timeout 1500 python "$@" 2>&1 | tail -4 ) >> $O 2>&1; }
run scripts/fuzz_campaign.py 21000000 ${N1:-12000}
run scripts/dev/fuzz_syn_long.py 810000 ${N2:-500}
run scripts/fuzz_large.py 41000 40
run scripts/fuzz_gather.py 510000 1500
run scripts/fuzz_class.py 530000 300
BSK_NO_SYN_PF=1 run scripts/dev/fuzz_syn_long.py 820000 150
echo "-- fuzz_campaign 22000000 3000 with BSK_NO_TILE_DEFER=1" >> $O; ( BSK_NO_TILE_DEFER=1 timeout 900 python scripts/fuzz_campaign.py 22000000 3000 2>&1 | tail -2 ) >> $O
cat $O
