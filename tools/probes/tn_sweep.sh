for w in 256 512 768 1024 2048; do
  echo "TN wgs $w"
  export SCOT_GEMM_TN_WGS=$w
  BK_COLD=1 python tools/bench_kernels.py gemm0 | grep "TN"
  BK_COLD=1 python tools/bench_kernels.py gemm1 | grep "TN"
done
