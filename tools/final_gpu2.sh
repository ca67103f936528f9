# round-2 closing GPU call (short budget): new-kernel parity on the MI355X, then the step A/B (fused passes vs their
# element-wise forms), then the reference-pinned step goldens, then the full default bench line.
set -x
mkdir -p gpurun_out/r02b
cd /root/repo
export TMPDIR=/tmp
(timeout 150 python -m pytest tests/test_kernels.py tests/test_functional.py tests/test_modules_vae.py -m gpu -q -x \
   -k "mel_loss or latent_head or group_norm_relu or deferred or bias_sink or plane_score or conformer_block" 2>&1 | tail -5) \
   > gpurun_out/r02b/pytest_new_kernels.log
timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r02b/bench_fused.json 2> gpurun_out/r02b/bench_fused.err
timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --legacy-paths > gpurun_out/r02b/bench_legacy.json 2> gpurun_out/r02b/bench_legacy.err
(timeout 300 python -m pytest tests/test_step_golden.py tests/test_task_step.py -m gpu -q 2>&1 | tail -12) > gpurun_out/r02b/pytest_step.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02b/bench_train_bf16x3.json 2> gpurun_out/r02b/bench_train_bf16x3.err
(timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6) > gpurun_out/r02b/pytest_gpu.log
tail -3 gpurun_out/r02b/*.log; grep -h "ms/step" gpurun_out/r02b/*.err; cat gpurun_out/r02b/bench_fused.json gpurun_out/r02b/bench_legacy.json | cut -c1-300
