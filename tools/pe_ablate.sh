#!/bin/bash
# GPU box, diagnostic build: kernel times of the paired-end path per ablation of the align kernel (1 no DP, 2 no seeding, 4 no chaining,
# 64 no cooperative DP, 128 nothing that misses the DP queues).  gpurun -- bash tools/pe_ablate.sh
for a in 0 1 2 4 64 128; do
  echo "ablate $a: $(KMAHIP_LIB=kma_amd/libkmahip_diag.so KMAHIP_ABLATE_ALIGN=$a timeout -k 10 200 python tools/pe_time.py 1000000 2>&1 | grep 'kernel times' | sed 's/.*seed_tasks/seed_tasks/')"
done
