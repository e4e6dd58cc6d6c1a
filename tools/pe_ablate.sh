for a in 0 1 128 16; do
  echo "ablate $a: $(KMAHIP_LIB=kma_amd/libkmahip_diag.so KMAHIP_ABLATE_ALIGN=$a timeout -k 10 200 python tools/pe_time.py 1000000 2>&1 | grep 'kernel times' | sed 's/.*seed_tasks/seed_tasks/')"
done
