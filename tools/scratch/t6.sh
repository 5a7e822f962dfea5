cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tests/c/build/bench_streams 1 200
tests/c/build/bench_streams 2 200
tests/c/build/bench_streams 1 100 --scene-peak
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_multigpu.py -q -m gpu 2>&1 | tail -5
