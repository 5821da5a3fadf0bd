import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j["roofline"]
        print("strict qps",j["value"],"fast",j["strict_ties"]["fast_mode_queries_per_s"],"main_ms",r["kernel_ms"],"ties",r["tie_replayed_queries"],"frac",r["frac"], "parity", j.get("parity_vs_oracle"))
