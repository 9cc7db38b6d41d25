cd /tmp && export TMPDIR=/tmp
for u in 0 7; do
  rm -rf /tmp/ru_$u
  FD_SPCONV_TILES=-1 FD_V2_UNIFORM=$u rocprofv3 --kernel-trace --stats -d /tmp/ru_$u -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/tiles_bench.py --iters 20 > /dev/null 2>&1
  f=$(find /tmp/ru_$u -name "*kernel_stats.csv" | head -1)
  echo "uniform=$u: $(grep res16 $f | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}' | head -1) $(grep res16 $f | cut -d, -f2-4 | tail -1)"
  grep res16 $f | head -1 | rev | cut -d, -f1-8 | rev
done
