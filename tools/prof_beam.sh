cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R && python -m pytest tests/test_hip_topk_logprob.py -m gpu -q -x 2>&1 | grep -E "assert|Error|error|^E " | head -12
cd /tmp
rm -rf /tmp/kb; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kb -o kb -- python $R/tools/beam_probe.py 5 1 32 > $O/r04_prof_beam.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kb/kb_results.db > $O/r04_beam_kernel_stats.md 2>&1
head -40 $O/r04_beam_kernel_stats.md | cut -c1-170
