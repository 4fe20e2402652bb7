set -e
mkdir -p gpurun_out
timeout -k 10 600 python scripts/dev_parity.py > gpurun_out/parity.log 2>&1 || true
grep -E "ALL OK|FAILURES|ids_equal=False" gpurun_out/parity.log || true
tail -16 gpurun_out/parity.log
for cfg in "100000 32 50" "100000 64 100" "1000000 32 100" "1000000 64 100" "1000000 1 50" "100000 64 10" "100000 32 10"; do
  set -- $cfg
  ROWS=$1 NQ=$2 K=$3 python scripts/dev_time.py
  CRAG_UNPIPELINED=1 ROWS=$1 NQ=$2 K=$3 python scripts/dev_time.py
done
python scripts/hybrid_bench.py
