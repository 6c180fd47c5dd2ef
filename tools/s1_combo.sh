#!/bin/bash
# S1 screen: the token-stationary form with / without the token-phase gate, the balanced token map and the two-chain main
# loop, and the K-outer one-round form.  bash tools/s1_combo.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
for combo in "0 0 0 0" "0 0 0 1" "0 1 1 0" "0 1 1 1" "0 1 0 1" "0 0 1 1" "5 0 0 0"; do
  set -- $combo
  echo "== variant $1 gate $2 balance $3 dual $4"
  SN_ASSIGN_VARIANT=$1 SN_ASSIGN_GATE=$2 SN_ASSIGN_BALANCE=$3 SN_ASSIGN_DUAL=$4 timeout -k 10 120 python tools/time_assign.py 30 2>&1 | grep -E "^variant"
done
