set -x
O=gpurun_out/r03i; mkdir -p $O
for nt in 1 2 4; do
  echo "NT=$nt" >> $O/skinny.log
  CRAG_SKINNY_NT=$nt timeout -k 10 200 python scripts/probes/skinny_bench.py >> $O/skinny.log 2>&1
done
grep -v amdgpu $O/skinny.log
timeout -k 10 400 python scripts/probes/encode_overlap.py > $O/overlap.log 2>&1; grep stream $O/overlap.log
