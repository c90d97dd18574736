C=ark_analysis_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -I$C scripts/ubench/step_phase_timing.hip $C/pxsom_api.hip $C/pxsom_assign_filter.hip $C/pxsom_assign_filter_acc.hip -ffinite-math-only -o /tmp/spt 2>&1 | grep -E "error" 
for g in 1 2; do echo "tpw $g"; /tmp/spt $g; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sptprof -o t -- /tmp/spt 2 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summarize.py /tmp/sptprof /tmp/sptprof/sum.txt > /dev/null; head -12 /tmp/sptprof/sum.txt | cut -c1-200
