#!/bin/bash
# Round-6 evidence in one call (GPU box): kernel tables (rocprofv3 --kernel-trace --stats) of the eager bench, the default
# bench, the one-at-a-time step with the deferred S1 finish, configs [3] / [4]; PMC passes (HBM bytes, instructions) of the
# eager bench; HBM bytes of the class branch on a pruned atlas, compacted and not.   bash tools/prof_r06.sh
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
mkdir -p $O
SN_BENCH_EAGER=1 SN_CLASS_BRANCH_FIRST=0 rocprofv3 --kernel-trace --stats -d $O/r06_eager -o e --output-format csv -- python3 $R/bench.py --steps 200 --warmup 10 --regions 1 --no-cpu-baseline --no-extra-legs > $O/r06_bench_eager_under_rocprof.json 2> $O/r06_eager.err && echo "eager done"
rocprofv3 --kernel-trace --stats -d $O/r06_default -o d --output-format csv -- python3 $R/bench.py --steps 200 --warmup 10 --regions 3 --no-cpu-baseline --no-extra-legs > $O/r06_bench_under_rocprof.json 2> $O/r06_default.err && echo "default done"
rocprofv3 --kernel-trace --stats -d $O/r06_predictor_step -o p --output-format csv -- python3 $R/tools/prof_shape.py c2 100 defer > $O/r06_predictor_step.txt 2>&1 && echo "predictor step done"
for s in c4 c5; do
  rocprofv3 --kernel-trace --stats -d $O/r06_$s -o p --output-format csv -- python3 $R/tools/prof_shape.py $s 10 > $O/r06_${s}_prof.txt 2>&1 && echo "$s done"
done
export SN_BENCH_EAGER=1 SN_BENCH_BATCHES=4
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $g -d $O/r06_pmc_p$i -o p -- python3 $R/bench.py --steps 5 --warmup 2 --regions 1 --no-cpu-baseline --no-extra-legs > $O/r06_pmc_p$i.log 2>&1 && echo "pmc pass $i done"
done
unset SN_BENCH_EAGER SN_BENCH_BATCHES
for c in 1 0; do
  for g in FETCH_SIZE WRITE_SIZE; do
    SN_ATLAS_COMPACT=$c rocprofv3 --kernel-trace --output-format csv --pmc $g -d $O/r06_pmc_pruned_c${c}_$g -o p -- python3 $R/tools/prof_pruned.py > $O/r06_pmc_pruned_c${c}_$g.log 2>&1 && echo "pruned compact=$c $g done"
  done
done
cd $R
python3 tools/pmc_to_json.py $O/r06_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, eager bench (--steps 5, 4 batches); per launch" $O/r06_pmc_p1 $O/r06_pmc_p2
python3 tools/pmc_to_json.py $O/r06_pmc_step_instructions.json "rocprofv3 --pmc SQ_* (two passes), eager bench; per launch" $O/r06_pmc_p3 $O/r06_pmc_p4
python3 tools/pmc_to_json.py $O/r06_pmc_pruned_atlas_compacted.json "pruned atlas (70 % of the class vertices under the threshold), compacted class branch: FETCH_SIZE / WRITE_SIZE per launch, eager steps" $O/r06_pmc_pruned_c1_FETCH_SIZE $O/r06_pmc_pruned_c1_WRITE_SIZE
python3 tools/pmc_to_json.py $O/r06_pmc_pruned_atlas_uncompacted.json "the same atlas through the plain fused route (SN_ATLAS_COMPACT=0)" $O/r06_pmc_pruned_c0_FETCH_SIZE $O/r06_pmc_pruned_c0_WRITE_SIZE
cd $R
python3 tools/trace_split.py $O/r06_eager > $O/r06_bench_eager_split_by_grid.txt
echo all done
python3 tools/pmc_by_grid.py $O/r06_pmc_p1 $O/r06_pmc_p2 > $O/r06_pmc_hbm_traffic_by_grid.txt 2>/dev/null
bash tools/prof_train_r06.sh
echo profiles done
