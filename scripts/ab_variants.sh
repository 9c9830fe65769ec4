#!/bin/bash
# Same-box A/B of library variants on the default bench (development aid):
#   scripts/ab_variants.sh [bench args] -- prints us/transition per variant in scripts/variants/
for lib in "" scripts/variants/*.so ""; do
  if [ -n "$lib" ]; then export BINF_LIB_OVERRIDE=$PWD/$lib; else unset BINF_LIB_OVERRIDE; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %.2f us  %.3e' % ('${lib:-default}', d['roofline']['avg_transition_us'], d['value']))" || exit 1
done
