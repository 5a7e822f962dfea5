#!/bin/bash
# one line per bench workload (ms/step, dominant kernel, per-pass times)
cd "$(dirname "$0")/.."
for w in ewa_lanczos_1080p_to_4k_dither10 bilinear_1080p_to_4k lanczos_1080p_to_4k_dither10 default_preset_1080p_to_4k nv12_1080p_to_4k_ewa_dither10 nv12_1080p_to_4k_default_preset hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap ewa_1080p_to_4k_hdr_tonemap; do
  python bench.py --workload $w --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['ms_per_step'], d['value'], r['kernel_us'], r['frac'], r['passes_us'])"
done
