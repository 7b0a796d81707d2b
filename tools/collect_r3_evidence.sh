#!/bin/bash
# After `gpurun -- 'bash tools/gpu_r3_final.sh'`: condense the rocprofv3 runs and copy the evidence files into profiles/
# (run locally; prints the numbers to check).  Then: gpurun -- 'python bench.py > gpurun_out/r3_bench.json' -> profiles/r3_bench.json
set -e
python tools/condense_profile.py r3 > /dev/null
python tools/condense_profile.py r3_cube6 > /dev/null
python - <<'PY'
import json,shutil
for t in ("r3","r3_cube6"):
    d=json.load(open(f"profiles/{t}_pmc.json")); print(t, d["_meta"])
O="gpurun_out/r3final/"
it={"iteration_4views":json.load(open(O+"iteration_4views.json")),"iteration_1view":json.load(open(O+"iteration_1view.json")),
    "probe_views":[json.loads(l) for l in open(O+"probe_views.txt") if l.startswith("{")]}
b=json.load(open(O+"bench_16M.json")); it["bench_16M_rays"]=b
json.dump(it,open("profiles/r3_iteration.json","w"),indent=1)
shutil.copy(O+"configs.json","profiles/r3_configs.json"); shutil.copy(O+"sweeps.txt","profiles/r3_sweeps.txt"); shutil.copy(O+"bench_gloo2.json","profiles/r3_bench_gloo2_rehearsal.json")
print("iteration 4 views / 1 view ms:", it["iteration_4views"]["iteration_ms_unsynchronised"], it["iteration_1view"]["iteration_ms_unsynchronised"])
print("16M rays:", b["value"], b["ms_per_step"])
b=json.load(open(O+"bench.json"))
print("bench:", b["ms_per_step"], b["value"], b["phase_ms"]["trace"], b["phase_ms"]["backtrace"], b["parity_check"]["ok"], b["lib_version"])
for k,v in b["variants"].items():
    if isinstance(v,dict): print(" ", k, v["ms_per_step"], v["trace"], v["backtrace"])
PY
