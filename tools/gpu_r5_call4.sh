set -u
out=gpurun_out/r5d; mkdir -p $out
timeout 900 python tools/bench_deep_gemm.py --model L --json $out/deep_gemm_L.json > $out/deep_gemm_L.txt 2>&1; cat $out/deep_gemm_L.txt
timeout 900 python tools/bench_deep_gemm.py --model B256 --json $out/deep_gemm_B256.json > $out/deep_gemm_B256.txt 2>&1; cat $out/deep_gemm_B256.txt
