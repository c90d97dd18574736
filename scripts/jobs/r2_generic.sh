bash scripts/jobs/r2_prof_cfg.sh 2>&1 | grep "== cfg\|update_prep\|exact_kernel\|cluster_sums_kernel\|filter_kernel"
