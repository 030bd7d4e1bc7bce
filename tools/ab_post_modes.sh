cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6p
timeout 300 python -m pytest tests/test_gpu_headline.py tests/test_gpu_layer.py tests/test_gpu_aggregation.py -q -x 2>&1 | tail -2
export P3R_LIB_PATH=plonky3_recursion_amd/knobs/libp3r_hip.so
for r in 1 2 3; do for m in 0 1 2; do
  P3R_POST_MODE=$m python bench.py --steps 20 --no-cpu-baseline --no-config2 --no-quintic --detail-out gpurun_out/r6p/detail_m${m}_$r.json > /dev/null 2>gpurun_out/r6p/err.txt
done; done
python - <<'EOF'
import json
for m in (0,1,2):
    for r in (1,2,3):
        d=json.load(open("gpurun_out/r6p/detail_m%d_%d.json"%(m,r)))
        print("mode",m,"run",r, round(d["ms_per_step"],3), d["proof_sha256"][:8], {k:round(v["ms_per_step"],3) for k,v in d["small_layers"].items()}, round(d.get("small_layer_throughput",{}).get("proofs_per_s",0),1))
EOF
unset P3R_LIB_PATH
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6p/trace -- python3 bench.py --no-cpu-baseline --no-config2 --no-small-layers --no-quintic --steps 4 --warmup 1 > gpurun_out/r6p/run.log 2>&1
python tools/host_gaps.py "gpurun_out/r6p/trace/**/*_kernel_trace.csv" > gpurun_out/r6p/gaps_after.txt; cat gpurun_out/r6p/gaps_after.txt; find gpurun_out/r6p -name "*kernel_trace.csv" -delete
