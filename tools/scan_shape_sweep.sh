#!/bin/bash
# Run ON THE GPU BOX: scan_se_kernel compiled for workgroups of 64 / 128 / 256 threads (4 / 8 / 16 strand items) at 7 or 8
# wavefronts per SIMD; step time and scan kernel time of each.
set -e
cd kma_amd/csrc
for shape in "64 7" "64 8" "128 7" "128 8" "256 7"; do
	set -- $shape
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DKMAHIP_STHREADS=$1 -DKMAHIP_SCAN_WAVES=$2 -c -o scan.o scan.hip 2>/dev/null
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkmahip.so db.o scan.o chain.o align.o longtrace.o conclave.o assemble.o ingest.o fragout.o index.o pipeline.o session.o comm.o api.o -lz
	(cd ../.. && timeout -k 10 300 python -m pytest tests/test_scan_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -1)
	(cd ../.. && timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('threads $1 waves $2:', round(d['ms_per_step'],3), 'scan', round(d['roofline']['scan']['kernel_ms'],3))")
done
