#!/bin/bash
# three driver-style bench runs back to back; prints the headline fields of each
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 $RH_BENCH_EXTRA > gpurun_out/r02_bench_run$i.json 2>/dev/null || exit 1
done
python - <<'PY'
import json
for i in (1, 2, 3):
    d = json.loads(open(f"gpurun_out/r02_bench_run{i}.json").read().strip().splitlines()[-1])
    dn = d["decode_nms"]
    print(i, "value %.4g ms/step %.4f frac(kernel) %.3f traffic %s | decode %.1f us (%.3f) trials %s first %.1f | single %.1f us (%.3f) | class %.1f us | cpu %.3g" % (
        d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], dn["ms_per_batch"] * 1e3,
        dn["frac_of_hbm_peak"], [round(t * 1e3, 1) for t in dn["stream_set_trials_ms_per_batch"]],
        dn["uncounted_first_trial_ms_per_batch"] * 1e3,
        dn["single_stream_ms_per_batch"] * 1e3, dn["single_stream_frac_of_hbm_peak"],
        dn["pipeline_class_ms_per_batch"] * 1e3, d["cpu_baseline"]["value"]))
PY
