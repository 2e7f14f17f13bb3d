#!/bin/bash
# reservation-window cost with the wider peel search / the dW_qkv + dW_o pairing (tools/comm_interference.py, one box)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r05_call20; mkdir -p $O
for rep in 1 2; do
for sw in "VLR_PEEL_MAX=3 VLR_PAIR_QKVO=0" "VLR_PEEL_MAX=3" "VLR_PAIR_QKVO=0" "-"; do
  if [ "$sw" = "-" ]; then pre=""; else pre="$sw"; fi
  echo "== [$sw] $rep" >> $O/out.txt
  env $pre timeout 200 python tools/comm_interference.py --steps 6 --wgs 0,16 --cus 0,16 --scopes backward 2>&1 | grep "^probe" >> $O/out.txt
done
done
cat $O/out.txt
