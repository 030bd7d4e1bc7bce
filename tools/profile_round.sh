# usage (on the GPU box, via gpurun): bash tools/profile_round.sh [tag]
# Raw output goes to gpurun_out/final/; tools/collect_profiles.py <round> turns it into profiles/<round>/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
(cd tools/microbench && ./int_rates > ../../gpurun_out/final/int_rates.txt 2>&1; ./perm_f64 > ../../gpurun_out/final/perm_f64.txt 2>&1; ./valu_classes > ../../gpurun_out/final/valu_classes.txt 2>&1)
# instructions per permutation of the dominant kernel: N launches over a matrix of known shape (tools/pmc_hash_rows.py)
for F in koala-bear baby-bear; do
  for C in SQ_INSTS_VALU SQ_WAVES; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/final/hashrows_${F}_$C -- python3 tools/pmc_hash_rows.py $F bench > gpurun_out/final/hashrows_${F}_$C.log 2>&1
  done
done
# the same for the arity-4 leaf kernel over the width-32 permutation, both instances (round-4 review item 5)
for W in builtin general; do
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/final/hashrows_w32_$W -- python3 tools/pmc_hash_rows.py koala-bear w32 $W > gpurun_out/final/hashrows_w32_$W.log 2>&1
done
# the bench line reads the instruction count just measured (profiles/<round>/pmc_hash_rows.json)
python3 tools/collect_profiles.py ${1:-r06} --hash-rows-only
# the driver's own command first (default flags): its ONE contract line, and how long the whole run takes
( time timeout 600 python bench.py --detail-out gpurun_out/final/bench_default_detail.json > gpurun_out/final/bench_default_line.json 2> gpurun_out/final/bench_default_err.log ) 2> gpurun_out/final/bench_default_time.txt
# then the profile round's form: every secondary leg, the detail dict is what profiles/<round>/bench_line_final.json keeps
timeout 1200 python bench.py --full --detail-out gpurun_out/final/bench_line.json > gpurun_out/final/bench_contract_line.json 2> gpurun_out/final/bench_err.log
ARGS="bench.py --no-cpu-baseline --no-config2 --no-small-layers --no-quintic"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python3 $ARGS > gpurun_out/final/stats_run.log 2>&1
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/final/pmc_$C -- python3 $ARGS --steps 1 --warmup 0 > gpurun_out/final/pmc_$C.log 2>&1
done
# keep the merge small: drop the per-dispatch traces except counter collection + stats
find gpurun_out/final -name "*kernel_trace.csv" -delete
find gpurun_out/final -name "*agent_info.csv" -delete
du -sh gpurun_out/final
# BASELINE config 5 (BabyBear, 2^22 rows) and config 4 (aggregation tree on one GPU): builder-run lines
timeout 900 python bench.py --field baby-bear --log-height 22 --steps 3 --no-cpu-baseline --no-config2 --no-small-layers --no-quintic --detail-out gpurun_out/final/bench_line_babybear_2p22.json > gpurun_out/final/bench_line_babybear_2p22.line 2> gpurun_out/final/bench_babybear_err.log
timeout 600 python bench.py --tree --steps 3 --warmup 1 --detail-out gpurun_out/final/bench_line_tree_1gpu.json > gpurun_out/final/bench_line_tree_1gpu.line 2> gpurun_out/final/bench_tree_err.log
timeout 600 python bench.py --tree --tree-workers 4 --steps 3 --warmup 1 --detail-out gpurun_out/final/bench_line_tree_1gpu_4workers.json > gpurun_out/final/bench_line_tree_1gpu_4workers.line 2>> gpurun_out/final/bench_tree_err.log
timeout 600 python bench.py --tree --tree-workers 4 --trees 4 --steps 3 --warmup 1 --detail-out gpurun_out/final/bench_line_forest_1gpu_4trees.json > gpurun_out/final/bench_line_forest_1gpu_4trees.line 2>> gpurun_out/final/bench_tree_err.log
# the plain multi-rank entry (the parent spawns the ranks; two ranks share the box's one GPU over gloo)
P3R_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --no-cpu-baseline --no-config2 --no-small-layers --no-quintic --detail-out gpurun_out/final/bench_line_2ranks_gloo.json > gpurun_out/final/bench_line_2ranks_gloo.line 2> gpurun_out/final/bench_2ranks_err.log
P3R_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --tree --trees 0 --tree-workers 2 --steps 3 --detail-out gpurun_out/final/bench_line_forest_2ranks_gloo.json > gpurun_out/final/bench_line_forest_2ranks_gloo.line 2>> gpurun_out/final/bench_2ranks_err.log
timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-config2 --no-small-layers --no-quintic --spans > /dev/null 2> gpurun_out/final/spans.txt
