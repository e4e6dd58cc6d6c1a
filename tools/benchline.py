import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["value"]/1e6,1), round(d["ms_per_step"],2), {k:round(d["roofline"][k]["kernel_ms"],2) for k in ("prefilter","scan","seed","align")})
