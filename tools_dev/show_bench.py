import json, sys
d = json.load(open(sys.argv[1]))
print("headline", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"])
print(json.dumps(d.get("batch_64"), indent=0)[:1200])
for k, v in d.get("configs", {}).items():
    print(k, round(v["ms_per_step"], 4), round(v["roofline"]["frac"], 3))
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["batch_64"]["value"])
