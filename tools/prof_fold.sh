cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pf -o pf -- python $GRAFT_REPO_ROOT/tools/ln_fold_ab.py 4 2 > /tmp/pf.log 2>&1
tail -2 /tmp/pf.log
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pf -name "*.db" | head -1) 2>/dev/null | head -24
