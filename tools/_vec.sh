python -m pytest tests/test_gpu_envs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
python tools/vec_env_probe.py 2>&1 | tail -1
