cd $GRAFT_REPO_ROOT
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:34]: v for k, v in r['passes_us'].items()})"; }
for wl in nv12_1080p_to_4k_default_preset nv12_1080p_to_4k_ewa_dither10 high_quality_preset_1080p_to_4k mix_24_to_60_ewa_1080p_to_4k; do echo -n "$wl: "; one $wl; done
