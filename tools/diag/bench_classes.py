import json,sys
j=json.loads(sys.stdin.read())
tag=sys.argv[1]
print(tag, "c2", j["value"], j["ms_per_step"])
for k,v in j["kernel_classes"].items(): print("  ", tag, "c2", k, v["ms_per_step"], v["launches_per_step"])
ns=j["north_star_shape"]; print(tag, "c3", ns["ms_per_step"])
for k,v in ns["kernel_classes"].items(): print("  ", tag, "c3", k, v["ms_per_step"], v["launches_per_step"])
