import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "value %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), " | ".join("%s %.4f (%.3f)" % (k["kernel"][:22], k["ms"], k["frac"]) for k in d.get("kernels",[])))
