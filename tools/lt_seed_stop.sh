# where the seeding kernel's time goes: reads end after their seeding (1), after the chain (2), or run to the end (0)
for a in 1 2 0; do echo "== KMAHIP_LT_STOP=$a"; KMAHIP_LT_STOP=$a KMAHIP_DEBUG_TIMING=1 python3 tools/mt1_time.py 100000 10000 5000000 0 2>&1 | grep -E "39588\+39588: seed done"; done
python3 tools/mt1_time.py 100000 10000 5000000 1000 2>&1 | tail -3
