#!/bin/bash
# round 6: is the row-tile anomaly the compiler's fused-multiply-add choice?  tail_bf16.hip with the SLP vectoriser ON and -ffp-contract fast (default) / on / off
for v in slp_on slp_on_contract_on slp_on_contract_off; do
  export GEM_HIP_LIB=$PWD/build/ab/libgem_$v.so
  echo "=== $v"
  python -m pytest tests/test_hip_determinism.py tests/test_hip_parity.py -q -m gpu -s -k "row_tile or instantiations" 2>&1 | grep -o 'OBSERVATION {"bitwise": [a-z]*\|OBSERVATION {"against_row_tiles": 2, "bitwise": [a-z]*\|passed.*\|failed.*'
done
