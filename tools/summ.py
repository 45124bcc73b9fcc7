import json,sys
d=json.load(open(sys.argv[1]))
c=d["config"]
print("value GB/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "pct", round(d["pct_hbm_roofline"],2), "launches", c["lsolve_launches"], c["ltsolve_launches"], "tasks", c["lsolve_tasks"], "nnzL", c["nnz_L"])
print("setup", {k:round(v,2) for k,v in d["setup"].items()}, "check", d["check"])
print("extras", {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["extras"].items()})
if "cpu_baseline" in d: print("cpu", d["cpu_baseline"])
