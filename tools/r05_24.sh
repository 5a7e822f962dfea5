#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export PL_HIP_DBG_NO_LISTING=1; else unset PL_HIP_DBG_NO_LISTING; fi
  echo -n "no_listing=$v: "
  python tools/r05_22.py 2>&1 | tail -1
done
