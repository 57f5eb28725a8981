#!/bin/bash
# A/B the lanes-per-pose knob of the ELL linearisation kernel (back-to-back passes, HIP events)
for T in 2 4 8; do echo "GS_ELL_LANES=$T"; GS_ELL_LANES=$T python scripts/lin_loop.py cfg4 200; done
for T in 2 4; do echo "cfg3 GS_ELL_LANES=$T"; GS_ELL_LANES=$T python scripts/lin_loop.py cfg3 200; done
