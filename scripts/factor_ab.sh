#!/bin/bash
# A/B the factor kernel variants (0 block VALU, 2 wave MFMA) on full iterations, interleaved in one session
for rep in 1 2; do for v in 4 2 3; do echo -n "variant $v: "; GS_FACTOR_VARIANT=$v python scripts/probe.py ${1:-cfg4} 2>&1 | grep per-iter; done; done
