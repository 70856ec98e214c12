#!/bin/bash
# ReaxFF GPU tests, the reach table, then same-box A/B (usage: tools/r05_reax_ab.sh <tag> "name:ENV=.." ...)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SCEMA_MD_TIMING=1 python bench.py --force-field reax --sims 2 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "bond order below" | sort -u
exec tools/r05_reax_check.sh "$@"
