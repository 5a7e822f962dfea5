#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -o "^\s*Name\s*:\s*[A-Za-z0-9_]*\|Counter_Name[^,]*\|^[A-Z][A-Za-z0-9_]*\b" | sort -u | grep -i "^ *Name\|TA_\|TCP_\|TCC_\|SQ_\|GRBM\|TD_" | tr -d ' ' | sed 's/Name://' | sort -u | tr '\n' ' ' > $GRAFT_REPO_ROOT/gpurun_out/r06_33_counters.txt
wc -c $GRAFT_REPO_ROOT/gpurun_out/r06_33_counters.txt
rocprofv3 --list-avail 2>/dev/null | head -60
