bash $GRAFT_REPO_ROOT/scripts/jobs/r3_baseline.sh
bash $GRAFT_REPO_ROOT/scripts/jobs/r3_pre_prof.sh
