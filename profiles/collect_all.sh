#!/bin/bash
# Round-end evidence on ONE box: bench lines of every BASELINE configuration (+ the gradient of cfg2 / cfg3 / cfg4) and the
# rocprofv3 kernel-stats + PMC summaries of the metric kernels, copied where profiles/ keeps them.
#   bash profiles/collect_all.sh <tag>      (from the repo root, on the GPU box)
TAG=${1:-r3z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
for c in cfg1 cfg3 cfg4 cfg5; do python bench.py --config $c --steps 40 --no-cpu-baseline > $OUT/${TAG}_bench_$c.json 2>/dev/null; done
for c in cfg2 cfg3 cfg4 nv20; do python bench.py --config $c --mode grad --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_${c}_grad.json 2>/dev/null; done
for c in cfg2 cfg2p cfg3 cfg4 cfg5; do
  bash profiles/collect.sh $TAG $c > $OUT/collect_$c.log 2>&1
  cp gpurun_out/prof_${TAG}_$c/trace/*/*_kernel_stats.csv $OUT/${TAG}_${c}_kernel_stats.csv 2>/dev/null
  cp gpurun_out/prof_${TAG}_$c/pmc_summary.txt $OUT/${TAG}_${c}_pmc_summary.txt 2>/dev/null
  cp gpurun_out/prof_${TAG}_$c/kernel_stats_steady.json $OUT/${TAG}_${c}_kernel_stats_steady.json 2>/dev/null
done
bash profiles/collect.sh ${TAG}g cfg4 --mode grad > $OUT/collect_cfg4_grad.log 2>&1
cp gpurun_out/prof_${TAG}g_cfg4/trace/*/*_kernel_stats.csv $OUT/${TAG}_cfg4_grad_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_${TAG}g_cfg4/pmc_summary.txt $OUT/${TAG}_cfg4_grad_pmc_summary.txt 2>/dev/null
python profiles/percall_boundaryA.py > $OUT/${TAG}_percall_boundaryA.json 2>/dev/null
python profiles/tile_split_timing.py > $OUT/${TAG}_tile_split.json 2>/dev/null
python profiles/small_batch_latency.py > $OUT/${TAG}_small_batch_latency.json 2>/dev/null
python profiles/reference_suite_timing.py > $OUT/${TAG}_reference_suite.json 2>/dev/null
python profiles/coopx_timing.py > $OUT/${TAG}_coopx_timing.json 2>/dev/null
python profiles/jvp_twin_timing.py 2>/dev/null | tail -1 > $OUT/${TAG}_jvp_twin_timing.json
python profiles/probes_wide_timing.py 2>/dev/null | tail -1 > $OUT/${TAG}_probes_wide_timing.json
python profiles/dc_per_cu_ab.py 2>/dev/null | tail -1 > $OUT/${TAG}_dc_per_cu_ab.json
python profiles/adaptive_ckpt_ab.py 2>/dev/null | tail -1 > $OUT/${TAG}_adaptive_ckpt_ab.json
# PMC of the narrow gradient kernels (one probe, four probes) and of the default architecture's gradient
bash profiles/pmc_target.sh ${TAG}_cfg2_grad profiles/grad_profile_target.py CFG=cfg2 REPS=4 > $OUT/${TAG}_cfg2_grad_pmc.txt 2>&1
bash profiles/pmc_target.sh ${TAG}_cfg3_grad profiles/grad_profile_target.py CFG=cfg3 REPS=3 > $OUT/${TAG}_cfg3_grad_pmc.txt 2>&1
bash profiles/pmc_target.sh ${TAG}_nv20_grad profiles/default_net_grad_profile_target.py NV=20 > $OUT/${TAG}_nv20_grad_pmc.txt 2>&1
bash profiles/pmc_target.sh ${TAG}_cfg4_grad profiles/grad_profile_target.py CFG=cfg4 REPS=3 > $OUT/${TAG}_cfg4_grad_pmc.txt 2>&1
bash profiles/kstats.sh ${TAG}_nv20_grad profiles/default_net_grad_profile_target.py NV=20 > /dev/null 2>&1
cp gpurun_out/${TAG}_nv20_grad_kernel_stats.csv $OUT/ 2>/dev/null
python profiles/grad3_check.py > $OUT/${TAG}_grad3_check.log 2>&1
# HBM traffic per step of every workload of the default line
bash profiles/traffic_all.sh ${TAG}t > $OUT/traffic.log 2>&1
cp $ROOT/gpurun_out/${TAG}t/traffic.json $OUT/${TAG}_traffic.json 2>/dev/null
cd $ROOT
python -m pytest tests -q -m gpu > $OUT/${TAG}_pytest_gpu.log 2>&1
# gpurun copies at most 64 MiB back: the raw rocprofv3 directories (kernel traces of several hundred launches, one per PMC pass) have
# been summarised above - only the summaries travel
rm -rf $ROOT/gpurun_out/prof_${TAG}_* $ROOT/gpurun_out/prof_${TAG}g_* $ROOT/gpurun_out/pmc_${TAG}_* $ROOT/gpurun_out/${TAG}t $ROOT/gpurun_out/${TAG}_nv20_grad_*
python profiles/brief.py $OUT/${TAG}_bench_*.json
