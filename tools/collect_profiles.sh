#!/bin/bash
# usage (on the GPU box): tools/collect_profiles.sh <tag>
# rocprofv3 --kernel-trace --stats of the default bench.py command and of the per-stage
# timing tool; the summaries land in gpurun_out/profiles_<tag>/ (copy them to profiles/).
set -e
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 $GRAFT_REPO_ROOT/bench.py > $out/bench_stdout.log 2>&1
grep "^{\"metric\"" $out/bench_stdout.log > $out/bench_line_under_rocprof.json || true
cp $(ls $out/bench/*/*kernel_stats.csv | head -1) $out/${tag}_bench_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train -- python3 $GRAFT_REPO_ROOT/tools/bench_stages.py --cfg 2 > $out/train_stdout.log 2>&1
cp $(ls $out/train/*/*kernel_stats.csv | head -1) $out/${tag}_loss_stages_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/infer -- python3 $GRAFT_REPO_ROOT/tools/bench_stages.py --cfg 2 --infer > $out/infer_stdout.log 2>&1
cp $(ls $out/infer/*/*kernel_stats.csv | head -1) $out/${tag}_postprocess_stages_kernel_stats.csv
rm -rf $out/bench $out/train $out/infer
ls -la $out
