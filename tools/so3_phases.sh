#!/bin/bash
# time of the descriptor kernel up to the end of each phase (tools build of the library: make -C gpr_calculator_amd/csrc ablation)
# usage (GPU box): bash tools/so3_phases.sh
export GPR_HIP_LIB=$PWD/gpr_calculator_amd/libgprhip_ablation.so
for nb in ${SO3_NBS:-64}; do echo "--- $nb structures"; for d in 1 2 3 4 0; do SO3_NB=$nb GPR_SO3_DBG=$d python3 tools/so3_only.py 2>&1 | grep "^so3"; done; done
