timeout 600 python -m pytest tests/test_gpu_som_kernels.py -m gpu -x -q -k "online" 2>&1 | tail -3
python bench.py --steps 3 --warmup 1 2>/dev/null | grep -o '"online_train": {[^}]*}'
bash scripts/jobs/ubench5.sh
