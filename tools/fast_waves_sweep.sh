#!/bin/bash
# Run ON THE GPU BOX: stage 3a's register-only kernel compiled for 4 / 5 / 6 wavefronts per SIMD (KMAHIP_FAST_WAVES), step time of each.
set -e
cd kma_amd/csrc
for w in 4 5 6; do
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DKMAHIP_FAST_WAVES=$w -c -o align.o align.hip 2>/dev/null
	/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkmahip.so db.o scan.o chain.o align.o longtrace.o conclave.o assemble.o ingest.o fragout.o index.o pipeline.o session.o comm.o api.o -lz
	(cd ../.. && python bench.py --steps 5 --warmup 1 --no-cpu "$@" | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('waves $w:', d['ms_per_step'], d['value'])")
done
