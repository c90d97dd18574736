# Round 5: create_pixel_matrix (plane-parallel division, page-locked bounce upload, tables assembled side by side) + packed-K filter with the
# node block in the score bits (config 5 shape probe)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_pre3
timeout 1500 python -m pytest tests/test_pipeline_dropin.py tests/test_gpu_preprocessing.py tests/test_gpu_som_kernels.py -m gpu -x -q > gpurun_out/r5_pre3/pytest.log 2>&1; tail -3 gpurun_out/r5_pre3/pytest.log
for n in 10 30; do python scripts/debug/create_pixel_matrix_timeline.py --fovs $n 2>&1 | tail -3; done | tee gpurun_out/r5_pre3/timeline.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o pk -- python $GRAFT_REPO_ROOT/scripts/debug/packed_filter_probe.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; python scripts/prof_summarize.py /tmp/pk gpurun_out/r5_pre3/packed_probe_trace.txt > /dev/null; grep -i "packed\|exact" gpurun_out/r5_pre3/packed_probe_trace.txt | cut -c1-60,96-170 | head
PXSOM_FUZZ_DTYPE=f16 PXSOM_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -x -q -k "fuzz_assign" 2>&1 | tail -2
