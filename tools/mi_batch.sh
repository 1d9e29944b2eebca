# batch level path: parity tests, then the mixed-integer bench workload batched against MPC_NO_BATCH=1 (run on the GPU box)
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -25
timeout 900 python tools/mi_batch.py 2>&1 | tail -12
