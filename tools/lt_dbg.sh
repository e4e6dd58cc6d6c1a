for d in 0; do
echo "=== dbg $d" >> gpurun_out/lt_dbg.log
KMAHIP_LT_DBG=$d KMAHIP_DEBUG_TIMING=1 timeout -k 5 40 python -m pytest tests/test_mt1_gpu.py -x -q -s 2>&1 | grep -E "=== dbg|done|does not|phase hist|  wg" >> gpurun_out/lt_dbg.log
done
cat gpurun_out/lt_dbg.log
