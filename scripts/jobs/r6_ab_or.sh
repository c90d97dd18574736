# same-box A/B of variant builds WITH the operating-range legs (rows in runs of 64 equal labels), + parity tests on the default build
R=$GRAFT_REPO_ROOT
cd $R; O=gpurun_out/r6_ab; mkdir -p $O
line() { python scripts/dev/ab_line.py ark_analysis_amd/variants/$1.so --no-cpu-baseline --no-online --no-pmc 2>$O/err_$1.log | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());o=d['operating_range'];print('$1', d['value'], d['phases_ms']['assign_filter_kernel'], d['phases_ms']['assign_and_mean_table'], 'runs of 64:', o['rows in runs of 64 equal labels']['labels_and_mean_table_one_pass_ms'], 'trained:', o['trained codebook']['labels_and_mean_table_one_pass_ms'], 'data rows:', o['codebook = data rows']['labels_and_mean_table_one_pass_ms'], 'near ties:', o['node pairs 1e-2 apart']['labels_and_mean_table_one_pass_ms'])"; }
for r in $(seq ${REPS:-3}); do for v in $VARIANTS; do line $v; done; done | tee $O/bench_or_$(echo $VARIANTS | tr ' ' '_').txt

