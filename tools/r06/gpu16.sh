bash ${GRAFT_REPO_ROOT:-/root/repo}/tools/prof_round6.sh
