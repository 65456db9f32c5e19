"""Print the few numbers of a bench.py JSON line one looks at while tuning: tools/bench_brief.py <log>"""
import json
import sys

for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    r = d.get("roofline", {})
    print("value %.1f %s  ms/step %.4f  kernel_ms %.4f  frac %.4f  frac_issued %.4f" % (
        d["value"], d["unit"], d["ms_per_step"], r.get("kernel_ms_avg", 0), r.get("frac", 0), r.get("frac_issued", 0)))
    print("work", json.dumps(d.get("work")))
    v = d.get("variants") or {}
    print("variants", json.dumps({k: ((round(x["value"], 1), round(x["ms_per_step"], 4)) if isinstance(x, dict) and "value" in x else x)
                                   for k, x in v.items()}))
    k = d.get("knn")
    if k:
        print("knn %.3e q/s %.3f ms; k1 %.3e %.3f ms" % (k["value"], k["ms_per_step"], k["nearest_k1"]["value"], k["nearest_k1"]["ms_per_step"]))
        if "presorted" in k:
            print("  presorted", json.dumps({a: (b if not isinstance(b, dict) else round(b["ms_per_step"], 3)) for a, b in k["presorted"].items() if a != "points"}))
        if "roofline" in k:
            print("  knn roofline frac %.4f kernel_ms %.3f" % (k["roofline"]["frac"], k["roofline"]["kernel_ms_avg"]))
    if "cpu_baseline" in d:
        print("cpu", json.dumps(d["cpu_baseline"])[:300])
