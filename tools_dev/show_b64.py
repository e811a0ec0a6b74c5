import json, sys
for n in sys.argv[1:]:
    d = json.load(open(n)); print(n.split("/")[-1], round(d["ms_per_step"], 4), round(d["value"]), d["config"]["launch"][:70])
