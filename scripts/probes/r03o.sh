set -x
O=gpurun_out/r03o; mkdir -p $O
S="1000000,1,50 1000000,64,50 100000,64,50 1000000,64,100 100000,64,100"
for sets in 0 1 2 4; do
  echo "SETS=$sets" >> $O/sets.log
  CRAG_PF_SETS=$sets timeout -k 10 300 python scripts/probes/search_bench.py $S >> $O/sets.log 2>&1
done
grep "SETS\|rows" $O/sets.log | cut -c1-260
