#!/bin/bash
# k_pass_chain with two pixels per lane (CHAIN_NP=2) against one
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
BASE=r06c NODRIVER=1 STEPS=200 bash tools/r05_ab.sh r06_42_chain_np2_ab hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap 2>&1 | tail -12
