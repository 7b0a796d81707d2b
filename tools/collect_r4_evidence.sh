#!/bin/bash
# After `gpurun -- 'bash tools/gpu_r4_final.sh'`: condense the rocprofv3 runs and copy the evidence files into profiles/
# (run locally; prints the numbers to check).
set -e
python tools/condense_profile.py r4 > /dev/null
python tools/condense_profile.py r4_cube6 > /dev/null
python - <<'PY'
import json,shutil
for t in ("r4","r4_cube6"):
    d=json.load(open(f"profiles/{t}_pmc.json")); print(t, d["_meta"], {k:round(v.get("effective_clock_ghz",0),3) for k,v in d.items() if k!="_meta" and "effective_clock_ghz" in v})
O="gpurun_out/r4final/"
it={"iteration_4views":json.load(open(O+"iteration_4views.json")),"iteration_1view":json.load(open(O+"iteration_1view.json"))}
json.dump(it,open("profiles/r4_iteration.json","w"),indent=1)
for a,b in (("configs.json","r4_configs.json"),("sweeps.txt","r4_sweeps.txt"),("bench_gloo2.json","r4_bench_gloo2_rehearsal.json"),
            ("bench_gloo2_overlap.json","r4_bench_gloo2_overlap_rehearsal.json"),("bench_gloo2_cube6.json","r4_bench_gloo2_cube6_rehearsal.json"),
            ("bench.json","r4_bench.json")):
    shutil.copy(O+a,"profiles/"+b)
ch={f"K{K}":json.load(open(O+f"chunk_G8_k3_K{K}.json")) for K in (0,4)}
json.dump({k:{"ms_per_step":v["ms_per_step"],"phase_ms":v["phase_ms"],"chunking":v.get("chunking"),"config":v["config"]["workload"]} for k,v in ch.items()},open("profiles/r4_chunking.json","w"),indent=1)
print("iteration 4 views / 1 view ms:", it["iteration_4views"]["iteration_ms_unsynchronised"], it["iteration_1view"]["iteration_ms_unsynchronised"])
b=json.load(open(O+"bench.json"))
print("bench:", b["ms_per_step"], b["value"], b["phase_ms"]["trace"], b["phase_ms"]["backtrace"], b["parity_check"]["ok"], b["lib_version"])
for k,v in b["variants"].items():
    if isinstance(v,dict): print(" ", k, v["ms_per_step"], v["trace"], v["backtrace"], v["adjoint_kernel"]["kernel"], v["adj_ns_ratio_to_headline"])
PY
