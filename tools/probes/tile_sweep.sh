for t in 8; do
  echo "tile $t"
  export SCOT_GEMM_TILE=$t
  for g in gemm1 gemm2 gemm3; do BK_COLD=1 python tools/bench_kernels.py $g | grep "gemm" | grep -v "NT fc2 s"; done
done
unset SCOT_GEMM_TILE
echo TN0
export SCOT_GEMM_TILE_TN=0
for g in gemm1 gemm2 gemm3; do BK_COLD=1 python tools/bench_kernels.py $g | grep "gemm TN"; done
