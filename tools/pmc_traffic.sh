#!/bin/bash
# usage (on the GPU box): tools/pmc_traffic.sh
# HBM traffic of the loss step's stream kernel from two separate rocprofv3 --pmc passes
# (FETCH_SIZE, then WRITE_SIZE; kernel trace only) over bench.py itself, turned into
# profiles-style JSON (gpurun_out/traffic_focal_stream.json) that names the kernel sources it was
# measured on.  Copy the result to profiles/traffic_focal_stream.json.
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_traffic
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 170 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline > $out/$ctr.log 2>&1 || { tail -5 $out/$ctr.log; exit 1; }
done
python3 - "$out" "$GRAFT_REPO_ROOT" <<'PY'
import collections, csv, glob, json, os, sys
out, root = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import bench
vals = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{ctr}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr and "focal_stream_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    vals[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
    with open(f"{out}/r03_pmc_{ctr.lower()}.txt", "w") as fh:
        for k, (m, n) in sorted(vals[ctr].items()):
            fh.write(f"{k}: {ctr} mean {m:.2f} KiB over {n} launches\n")
algo = bench.BATCH * 69354 * (8 * bench.CLASSES + 48)
rec = {"config": "B=16, N=69354, C=80 (bench.py workload)",
       "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --repeats 3 --no-cpu-baseline ; "
                  "the same with --pmc WRITE_SIZE (separate passes; tools/pmc_traffic.sh)",
       "correction": "gfx950 FETCH_SIZE reports half of a wide coalesced read stream: read bytes = 2 * FETCH_SIZE * 1024; "
                     "WRITE_SIZE is exact for 16-byte stores",
       "how": "rocprofv3 PMC passes (FETCH_SIZE x2 correction + WRITE_SIZE), committed summary matched by source hash",
       "source_sha256": bench.kernel_source_hash(),
       "algorithmic_bytes_per_launch": algo, "kernels": {}}
for k in vals["FETCH_SIZE"]:
    if k not in vals["WRITE_SIZE"]:
        continue
    rd = 2 * vals["FETCH_SIZE"][k][0] * 1024
    wr = vals["WRITE_SIZE"][k][0] * 1024
    rec["kernels"][k] = {"FETCH_SIZE_KiB_mean": vals["FETCH_SIZE"][k][0], "WRITE_SIZE_KiB_mean": vals["WRITE_SIZE"][k][0],
                         "launches": [vals["FETCH_SIZE"][k][1], vals["WRITE_SIZE"][k][1]],
                         "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr),
                         "hbm_bytes_per_launch": int(rd + wr)}
json.dump(rec, open(os.path.join(root, "gpurun_out", "traffic_focal_stream.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
