#!/bin/bash
# same-box A/B: alternate the bench of this tree and of a base tree checked out under .ab_base (git worktree add .ab_base <commit>; build there);
# extra arguments go to both bench runs (e.g. --c5)
for i in 1 2 3; do
  (cd $GRAFT_REPO_ROOT/.ab_base && python bench.py --steps 300 --no-cpu --no-native --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base ', d['ms_per_step'])")
  (cd $GRAFT_REPO_ROOT && python bench.py --steps 300 --no-cpu --no-native --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head ', d['ms_per_step'])")
done
