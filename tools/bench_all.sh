#!/bin/bash
# The bench lines of a round (through gpurun, from the repo root):  TAG=r05 bash tools/bench_all.sh
# -> gpurun_out/${TAG}_bench_<workload>.json; copy what is to be judged to profiles/.
TAG=${TAG:-r05}
python bench.py > gpurun_out/${TAG}_bench_triaxial256.json 2> gpurun_out/${TAG}_bench.err
for w in marine128 salt384 uniform256; do python bench.py --workload $w --no-survey --no-256 > gpurun_out/${TAG}_bench_$w.json 2>/dev/null; done
TAG=$TAG python - <<'PY'
import json, os
tag = os.environ['TAG']
for w in ("triaxial256", "marine128", "salt384", "uniform256"):
    d = json.load(open(f"gpurun_out/{tag}_bench_{w}.json"))
    t = d.get("time_to_tol", {})
    print(w, round(d["ms_per_step"], 2), round(d["value"]), round(d["roofline"]["frac"], 4), t.get("gpu", {}).get("cycles"),
          round(t.get("gpu", {}).get("seconds", 0), 3), t.get("reduced_copy"))
d = json.load(open(f"gpurun_out/{tag}_bench_triaxial256.json"))
for k, v in d["smoothers_256"]["smoothers"].items():
    print(k, round(v["ms_per_call"], 3), round(v["frac"], 4), round(v["frac_delivered"], 4))
for blk in ("survey_8_sources", "survey_config5"):
    print(blk, {k: (round(v["ms_per_source"], 1), v["cycles"]) for k, v in d.get(blk, {}).items() if isinstance(v, dict) and "ms_per_source" in v})
PY
