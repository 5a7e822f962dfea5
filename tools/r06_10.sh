#!/bin/bash
# round 6, tenth GPU call: non-temporal stores for an INTERMEDIATE (the debanded 8K rgba16hf plane,
# 265 MB, read by the next pass): A = library with -DPLH_DEBAND_NT, B = the tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
BASE=debandnt NODRIVER=1 STEPS=100 bash tools/r05_ab.sh r06_10_deband_nt_ab ewa_8k_to_4k_deband_tonemap hdr10_4k_tonemap_high_quality high_quality_preset_1080p_to_4k
