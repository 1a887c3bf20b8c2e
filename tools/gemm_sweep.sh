# Shapes of the K = N statistics contraction (evoamd_gemm_tn) as the configs launch it:
#   bash tools/gemm_sweep.sh      (c2, c3, c4 shard, c4/100k, c5)
cd /root/repo
for shape in "10000 512 128 384" "50000 256 64 -1" "12500 1280 512 768" "100000 1280 512 768" "25000 1024 256 -1"; do
  timeout -k 10 100 python tools/bench_gemm.py $shape
done
