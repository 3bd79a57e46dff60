#!/bin/bash
# The round's stress record on the final kernels (run through gpurun; logs under gpurun_out/stress_r03_*.log).
# Every step under its own time limit, chained: a failure or a limit ends the script.
set -e
o=gpurun_out
export STRESS_LISTS=1
timeout -k 10 280 python -u tools/stress_parity.py 20000 2500 huge                > $o/stress_r03_huge.log 2>&1
timeout -k 10 200 python -u tools/stress_parity.py 30000 600 huge 1e5             > $o/stress_r03_huge_1e5.log 2>&1
timeout -k 10 200 python -u tools/stress_parity.py 31000 600 huge 0 1 wild        > $o/stress_r03_huge_wild.log 2>&1
MRT_CULL_TEAM=4 timeout -k 10 200 python -u tools/stress_parity.py 32000 800 huge > $o/stress_r03_huge_team4.log 2>&1
timeout -k 10 280 python -u tools/stress_parity.py 40000 6000 both                > $o/stress_r03_both.log 2>&1
timeout -k 10 280 python -u tools/stress_parity.py 50000 4000 both 0 1 wild       > $o/stress_r03_both_wild.log 2>&1
timeout -k 10 200 python -u tools/stress_parity.py 60000 2000 both 1e4            > $o/stress_r03_both_1e4.log 2>&1
unset STRESS_LISTS
timeout -k 10 280 python -u tools/stress_bounce.py 70000 3000                     > $o/stress_r03_bounce.log 2>&1
timeout -k 10 280 python -u tools/stress_bounce.py 80000 3000 --wild              > $o/stress_r03_bounce_wild.log 2>&1
timeout -k 10 200 python -u tools/stress_bounce.py 90000 1000 --shift 1e4         > $o/stress_r03_bounce_1e4.log 2>&1
timeout -k 10 280 python -u tools/stress_terrain.py 100000 1500                   > $o/stress_r03_terrain.log 2>&1
timeout -k 10 280 python -u tools/stress_terrain.py 110000 400 0 1 hard           > $o/stress_r03_terrain_hard.log 2>&1
timeout -k 10 200 python -u tools/stress_terrain.py 120000 400 3e4                > $o/stress_r03_terrain_3e4.log 2>&1
echo ALL STRESS STEPS DONE
