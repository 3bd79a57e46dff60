#!/bin/bash
# Stress record for the pair lists of the team kernel (mrt_cull.hip, PAIRS): terrains and large random scenes under teams
# dealing leaves, against the brute-force kernel bit for bit (run through gpurun; logs under gpurun_out/stress_r03p_*.log).
set -e
o=gpurun_out
first=${1:-200000}
export STRESS_LISTS=1
timeout -k 10 280 python -u tools/stress_terrain.py $first 1500                    > $o/stress_r03p_terrain.log 2>&1
timeout -k 10 280 python -u tools/stress_terrain.py $((first + 10000)) 400 0 1 hard > $o/stress_r03p_terrain_hard.log 2>&1
timeout -k 10 200 python -u tools/stress_terrain.py $((first + 20000)) 400 3e4      > $o/stress_r03p_terrain_3e4.log 2>&1
timeout -k 10 200 python -u tools/stress_terrain.py $((first + 25000)) 300 0 1e-3   > $o/stress_r03p_terrain_small.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 280 python -u tools/stress_parity.py $((first + 30000)) 1200 huge        > $o/stress_r03p_huge_team4.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 200 python -u tools/stress_parity.py $((first + 40000)) 600 huge 0 1 wild > $o/stress_r03p_huge_team4_wild.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 200 python -u tools/stress_parity.py $((first + 50000)) 600 huge 1e5      > $o/stress_r03p_huge_team4_1e5.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 280 python -u tools/stress_parity.py $((first + 60000)) 3000 both         > $o/stress_r03p_both_team4.log 2>&1
echo ALL PAIRS STRESS STEPS DONE
