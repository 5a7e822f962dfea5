cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
echo "== self-spawned, 2 ranks on device 0 (gloo)"
PL_BENCH_DEVICES=0,0 PL_BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 100 --warmup 20 2>/tmp/e1 | cut -c1-330; tail -2 /tmp/e1
echo "== torchrun form"
PL_BENCH_DEVICES=0,0 PL_BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 100 --warmup 20 2>/tmp/e2 | cut -c1-330; tail -2 /tmp/e2
echo "== nccl requested with two ranks on one device: must fall back, not die"
PL_BENCH_DEVICES=0,0 timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 2>/tmp/e3 | cut -c1-200; grep -c "using gloo" /tmp/e3; tail -2 /tmp/e3 | cut -c1-300
