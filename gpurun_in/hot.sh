cd $GRAFT_REPO_ROOT
free -g | head -2
for fam in deep river; do for size in 2000 5000; do
  time python bench.py --only hotpath --size $size --family $fam 2> gpurun_out/r05_hot_${fam}_$size.err > gpurun_out/r05_hot_${fam}_$size.json
  tail -3 gpurun_out/r05_hot_${fam}_$size.err; cat gpurun_out/r05_hot_${fam}_$size.json | cut -c1-1800
done; done
