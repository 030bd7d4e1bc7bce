cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 900 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench_err.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python3 bench.py --no-cpu-baseline --no-config2 > gpurun_out/final/stats_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-config2 --steps 1 --warmup 0 > gpurun_out/final/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_write -- python3 bench.py --no-cpu-baseline --no-config2 --steps 1 --warmup 0 > gpurun_out/final/pmc_write.log 2>&1
find gpurun_out/final -name "*.csv" | head -20
# keep the merge small: drop the per-dispatch traces except counter collection + stats
find gpurun_out/final/stats -name "*kernel_trace.csv" -delete
ls -la gpurun_out/final/*
