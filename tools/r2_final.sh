#!/bin/bash
# evidence run: rocprofv3 summaries of bench.py and of the stage tool, three driver-style bench
# runs, the configuration table
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r02 > gpurun_out/collect_r02.log 2>&1 || { tail -20 gpurun_out/collect_r02.log; exit 1; }
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_run$i.json 2>/dev/null || exit 1
done
bash tools/config_table.sh > gpurun_out/r02_config_table.txt || exit 1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import json
for i in (1, 2, 3):
    d = json.loads(open(f"gpurun_out/r02_bench_run{i}.json").read().strip().splitlines()[-1])
    dn = d["decode_nms"]
    print(i, "value %.4g ms/step %.4f frac(kernel) %.3f traffic %s | decode %.1f us (%.3f) single %.1f us (%.3f) | cpu %.3g" % (
        d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], dn["ms_per_batch"] * 1e3,
        dn["frac_of_hbm_peak"], dn["single_stream_ms_per_batch"] * 1e3, dn["single_stream_frac_of_hbm_peak"],
        d["cpu_baseline"]["value"]))
PY
cat gpurun_out/r02_config_table.txt | cut -c1-60
