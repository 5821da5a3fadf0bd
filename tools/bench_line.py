import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); r=j["roofline"]
        print("strict qps",j["value"],"fast",j["strict_ties"]["fast_mode_queries_per_s"],"two callers",j.get("two_caller_threads_queries_per_s"),"simd8",j.get("simd_order_queries_per_s"),"main_ms",r["kernel_ms"],"literal",r.get("queries_resolved_with_literal_heaps"),"equal",r.get("queries_that_met_equal_distances"),"frac",r["frac"], "parity", j.get("parity_vs_oracle"))
