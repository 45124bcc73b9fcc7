"""Round 3: turn the config runs of tools/r3_configs.sh (gpurun_out/r3_*.json) into the committed summaries under profiles/."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def load(name):
    with open(os.path.join(OUT, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def dump(name, obj):
    with open(os.path.join(PROF, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")


widths = {w: load(f"r3_c3_w{w}.json") for w in (256, 128, 64, 32)}
dump("r03_config_c3.json", {"source": "tools/r3_configs.sh: python bench.py --config c3 --nrhs 256 --steps 3 --warmup 1 (one MI355X)", "line": widths[256]})
dump("r03_config_c5.json", {"source": "tools/r3_configs.sh: python bench.py --config c5 --steps 2 --warmup 1 (one MI355X, host arrays: PCIe inclusive)", "line": load("r3_c5.json")})
dump("r03_config_c4.json", {"source": "tools/r3_configs.sh: python tools/configs.py c4 (one MI355X)", "line": load("r3_c4.json")})

# C3 is one system with 256 right-hand sides: the only way N GPUs share it is by columns (rsp_set_devices: each device solves
# nrhs / N columns with its own replica of the factor's programmes).  The per-device time at N GPUs is therefore the
# single-GPU time at width 256 / N, measured here; the curve below is that PROJECTION, not a multi-GPU measurement.
t = {w: widths[w]["ms_per_step"] for w in widths}
curve = []
for n_gpus in (1, 2, 4, 8):
    w = 256 // n_gpus
    curve.append({"n_gpus": n_gpus, "columns_per_gpu": w, "ms_per_step_measured_on_one_gpu_at_this_width": round(t[w], 3),
                  "projected_speedup_vs_1gpu": round(t[256] / t[w], 3), "projected_efficiency": round(t[256] / t[w] / n_gpus, 3)})
dump("r03_c3_strong_widths.json", {
    "what": "C3 (Laplacian 80^3, lsolve+ltsolve, 256 RHS in total) split by right-hand-side columns over N GPUs: PROJECTED from "
            "single-GPU runs at the per-GPU width; unmeasured at N>1 (the round-end driver holds the only multi-GPU node)",
    "why_it_flattens": "the factor (4.3 GB of coefficients) is read once per solve whatever the width: below ~64 columns the step is "
                       "bound by streaming L, not by the products, so narrower shares stop getting faster",
    "excluded": "the one-time device-to-device copy of the solve programmes to each replica (rsp_set_devices; hipMemcpyPeer)",
    "kernels_sha": widths[256]["roofline"].get("kernels_sha"),
    "curve": curve,
})
print(json.dumps(curve, indent=1))
