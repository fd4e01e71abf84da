#!/bin/bash
# GPU suite, then the benchmark three times the way the driver calls it, then once without prefetch
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -u -m pytest tests -x -q --timeout 120 --timeout-method thread -m gpu > gpurun_out/r2_tests.log 2>&1 || { tail -40 gpurun_out/r2_tests.log; exit 1; }
tail -2 gpurun_out/r2_tests.log
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_$i.json 2> gpurun_out/r2_bench_$i.err || { tail -20 gpurun_out/r2_bench_$i.err; exit 1; }
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-prefetch --no-cpu-baseline > gpurun_out/r2_bench_noprefetch.json 2> gpurun_out/r2_bench_np.err || { tail -20 gpurun_out/r2_bench_np.err; exit 1; }
python - <<'PY'
import json
for f in ("r2_bench_1","r2_bench_2","r2_bench_3","r2_bench_noprefetch"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    dn=d["decode_nms"]
    print(f, "value %.4g ms/step %.4f p10/p90 %s  parity %s | stream %.4f frac %.3f | decode %.1f us (%.3f) p10/p90 %s single %.1f us (%.3f)" % (
        d["value"], d["ms_per_step"], [round(x,4) for x in d["ms_per_step_p10_p90"]], d["parity_ok"], d["roofline"]["kernel_ms_mean"], d["roofline"]["frac"],
        dn["ms_per_batch"]*1e3, dn["frac_of_hbm_peak"], [round(x*1e3,1) for x in dn["ms_per_batch_p10_p90"]], dn["single_stream_ms_per_batch"]*1e3, dn["single_stream_frac_of_hbm_peak"]))
PY
