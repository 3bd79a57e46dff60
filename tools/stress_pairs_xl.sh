#!/bin/bash
# The long version of tools/stress_pairs.sh (about 18 minutes of GPU time): more seeds, more shifts and scales.
set -e
o=gpurun_out
first=${1:-300000}
export STRESS_LISTS=1
timeout -k 10 280 python -u tools/stress_terrain.py $first 4500                      > $o/stress_r03q_terrain.log 2>&1
timeout -k 10 280 python -u tools/stress_terrain.py $((first + 10000)) 1200 0 1 hard  > $o/stress_r03q_terrain_hard.log 2>&1
timeout -k 10 120 python -u tools/stress_terrain.py $((first + 20000)) 1500 1e5       > $o/stress_r03q_terrain_1e5.log 2>&1
timeout -k 10 200 python -u tools/stress_terrain.py $((first + 22000)) 1200 1e6       > $o/stress_r03q_terrain_1e6.log 2>&1
timeout -k 10 200 python -u tools/stress_terrain.py $((first + 24000)) 1500 0 1e3     > $o/stress_r03q_terrain_big.log 2>&1
timeout -k 10 200 python -u tools/stress_terrain.py $((first + 26000)) 600 3e3 1e-2 hard > $o/stress_r03q_terrain_small_hard.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 200 python -u tools/stress_parity.py $((first + 30000)) 6000 huge        > $o/stress_r03q_huge_team4.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 150 python -u tools/stress_parity.py $((first + 40000)) 3000 huge 0 1 wild > $o/stress_r03q_huge_team4_wild.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 150 python -u tools/stress_parity.py $((first + 50000)) 3000 huge 1e6      > $o/stress_r03q_huge_team4_1e6.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 150 python -u tools/stress_parity.py $((first + 60000)) 8000 both 0 1 wild > $o/stress_r03q_both_team4_wild.log 2>&1
echo ALL PAIRS XL STRESS STEPS DONE
