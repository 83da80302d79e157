#!/bin/bash
# replica lanes (streams) on mid-size launches
for n in 1 2 3 4; do echo "ISINGMC_STREAMS=$n"; ISINGMC_STREAMS=$n python3 tools/lanes_ab.py 2>&1 | grep -v amdgpu | head -3; done
