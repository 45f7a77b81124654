# tests/test_bench_launch.py puts this directory on PYTHONPATH: bench.py's launcher branch must finish without importing torch
raise ImportError("torch must not be imported by bench.py's launcher")
