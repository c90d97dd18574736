# where the screened exact kernel's time went on config 4's lists: ablation builds (hooks removed from the tree since; labels wrong, timing only) -> gpurun_out/r4_exact/ablate.txt
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4_exact
OUT=$R/gpurun_out/r4_exact/ablate.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for a in 0 1 3 7 4 2; do
  export PXSOM_HIPCC_EXTRA="-DPXSOM_SCREEN_ABLATE=$a"
  (cd $R && python -c "
from ark_analysis_amd import _build
_build.build()" > /dev/null 2>&1)
  rm -rf /tmp/pa
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -o t -- python $R/bench.py --config cfg4 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > /dev/null 2>&1
  python $R/scripts/prof_summarize.py /tmp/pa /tmp/pa/sum.txt bmu_exact > /dev/null
  echo "=== ablate $a (1: pairs not evaluated, 2: no node walk, 4: no reference distance)" >> $OUT
  grep -E "bmu_exact" /tmp/pa/sum.txt | head -4 | cut -c1-60,100-190 >> $OUT
done
cat $OUT
