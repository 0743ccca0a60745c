python bench.py > gpurun_out/r03_bench_triaxial256.json 2> gpurun_out/r03_bench.err
for w in marine128 salt384 uniform256; do python bench.py --workload $w --no-survey --no-256 > gpurun_out/r03_bench_$w.json 2>/dev/null; done
python - <<PY
import json
for w in ("triaxial256","marine128","salt384","uniform256"):
    d=json.load(open(f"gpurun_out/r03_bench_{w}.json"))
    t=d.get("time_to_tol",{})
    print(w, round(d["ms_per_step"],2), round(d["value"]), round(d["roofline"]["frac"],4), t.get("gpu",{}).get("cycles"), round(t.get("gpu",{}).get("seconds",0),3), t.get("reduced_copy"))
d=json.load(open("gpurun_out/r03_bench_triaxial256.json"))
for k,v in d["smoothers_256"]["smoothers"].items(): print(k, round(v["ms_per_call"],3), round(v["frac"],4), round(v["frac_delivered"],4))
PY
