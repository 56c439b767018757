import json,sys
d=json.load(open(sys.argv[1]))
print("aggregate GB/s", d["value"])
for r in d["cases"]: print(r["quantizer"], r["shape"], r["us"], r["frac_of_8TBs"])
