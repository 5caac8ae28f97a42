import sys, json
for l in sys.stdin:
    try: j = json.loads(l)
    except Exception: continue
    print(j["config"]["workload"], round(j["ms_per_step"], 5), "ms")
