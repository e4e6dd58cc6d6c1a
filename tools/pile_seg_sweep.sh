# the pile-up's unit length against depth 2 000x (1 M reads of 10 kb vs 5 Mb): KMAHIP_PILE_SEG_COLS
for s in 512 1024 2048 4096; do echo "== KMAHIP_PILE_SEG_COLS=$s"; KMAHIP_PILE_SEG_COLS=$s python3 tools/c4_time.py ${1:-1000000} 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['stage_ms'], round(d['reads_per_s']))"; done
