timeout 900 python -m pytest tests/test_scene_gpu.py -q -x -k nat 2>&1 | tail -15
