# lane kernels: 0 = as is, 1 = no matrix stores, 2 = no walk, 3 = neither, 6 = neither sweep nor walk (staging only)
for a in 0 1 2 3 6; do echo "== KMAHIP_LT_ABLATE=$a"; KMAHIP_LT_ABLATE=$a KMAHIP_DEBUG_TIMING=1 python3 tools/mt1_time.py 39588 10000 5000000 0 2>&1 | grep -E "lane class done" | tail -9 | awk '{print $10}' | tr '\n' ' '; echo; done
