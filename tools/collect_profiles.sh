#!/bin/bash
# Everything profiles/ holds for one round, in one GPU-box call.  usage: tools/collect_profiles.sh r03
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; cd $R; mkdir -p gpurun_out/$TAG
python3 bench.py > gpurun_out/$TAG/bench_default_run.json 2> gpurun_out/$TAG/bench_default_run.err
python3 bench.py --math bf16 --batch 48 --no-cpu-baseline > gpurun_out/$TAG/bench_bf16_b48.json 2>/dev/null
tools/profile_step.sh $TAG/step_bf16 24 "bf16 activation mode (BASELINE config 4), batch 48" --math bf16 --batch 48 --steps 20 --warmup 3 --no-cpu-baseline > /dev/null
tools/profile_step.sh $TAG/step_f32 24 "fp32 (BASELINE config 2), batch 24" --steps 20 --warmup 3 --no-cpu-baseline --no-side-leg > /dev/null
# dominant kernels alone: fp32 gather reads whole lines (FETCH_SIZE x 2), the bf16 kernels' pieces are counted exactly (x 1)
tools/profile_roofline.sh ${TAG}_bf16_b48 bf16 48 conv conv_bf16_v2_kernel 1 > gpurun_out/$TAG/roofline_bf16.log 2>&1
tools/profile_roofline.sh ${TAG}_bf16_wgrad_b48 bf16 48 wgrad igemm_wgrad_b16_kernel 1 > gpurun_out/$TAG/roofline_bf16_wgrad.log 2>&1
tools/profile_roofline.sh ${TAG}_f32_b24 f32 24 conv igemm_fwd_kernel 2 > gpurun_out/$TAG/roofline_f32.log 2>&1
python3 tools/conv16_bench.py 48 > gpurun_out/$TAG/conv16_layers.txt 2>/dev/null
python3 tools/layer_bench.py 24 > gpurun_out/$TAG/layers_f32_b24.txt 2>/dev/null
python3 tools/timeline.py gpurun_out/$TAG/step_bf16/p_kernel_trace.csv 4 > gpurun_out/$TAG/timeline_bf16.txt 2>&1
python3 tools/timeline.py gpurun_out/$TAG/step_f32/p_kernel_trace.csv 4 > gpurun_out/$TAG/timeline_f32.txt 2>&1
python3 tools/replay_regimes.py 24 f32 all 2>/dev/null | grep "ms/step" > gpurun_out/$TAG/replay_regimes_f32.txt
python3 tools/replay_pieces.py 24 2>/dev/null | grep "ms/step" > gpurun_out/$TAG/replay_pieces_f32.txt
python3 tools/replay_pieces.py 48 bf16 2>/dev/null | grep "ms/step" > gpurun_out/$TAG/replay_pieces_bf16.txt
python3 tools/host_bound.py 24 2>/dev/null | tail -1 > gpurun_out/$TAG/host_bound.txt
python3 tools/host_bound.py 48 bf16 2>/dev/null | tail -1 >> gpurun_out/$TAG/host_bound.txt
python3 tools/replay_host_times.py 24 2>/dev/null | grep "step\|launches" > gpurun_out/$TAG/replay_host_times_f32.txt
python3 tools/image_layer_bench.py 24 2>/dev/null | grep px > gpurun_out/$TAG/image_layers.txt
python3 tools/image_layer_bench.py 48 bf16 2>/dev/null | grep px >> gpurun_out/$TAG/image_layers.txt
rm -f gpurun_out/$TAG/step_*/p_kernel_trace.csv
find gpurun_out/$TAG gpurun_out/roofline_${TAG}_* -name "*.csv" -size +2M -delete 2>/dev/null
tail -1 gpurun_out/$TAG/bench_default_run.json | cut -c1-300
