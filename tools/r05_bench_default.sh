#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
python bench.py 2>gpurun_out/r05/bench_default.err | tail -1 > gpurun_out/r05/bench_default.json
python - <<PY
import json
j=json.load(open("gpurun_out/r05/bench_default.json"))
print("value", j["value"], "ms/step", j["ms_per_step"], "single", j["single_frame"]["value"], "kernel", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], j["roofline"]["lane_slot_frac"])
for k,v in j["extra_workloads"].items():
    if "value" in v: print(k, round(v["value"],1), v.get("roofline_frac"), v.get("lane_slot_frac"), v.get("kernel_ms"))
print("boundary", [(r["spp"], round(r["ratio_to_device_resident"],3), round(r["pinned_film_ratio_to_device_resident"],3)) for r in j["boundary"]["rates"]], j["boundary"]["host_threads_adding"], j["boundary"]["cpus_granted"])
print("scaling", j["projected_scaling"]["kernel_efficiency"], j["projected_scaling"]["pipelined"]["efficiency"], j["projected_scaling"]["stress"]["kernel_efficiency"])
print("cpu", j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"], j.get("speedup_vs_cpu_baseline"), j["rmse_full_spp"]["value"])
print("jit", j["extra_workloads"]["run_time_instantiations"])
PY
tail -5 gpurun_out/r05/bench_default.err
