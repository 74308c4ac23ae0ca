for t in 8 6; do
  echo "tile $t"
  export SCOT_GEMM_TILE=$t
  BK_COLD=1 python tools/bench_kernels.py gemm2 | grep "gemm" | grep -v "fc2 s"
  BK_COLD=1 python tools/bench_kernels.py gemm3 | grep "gemm" | grep -v "fc2 s"
done
