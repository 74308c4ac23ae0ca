set -u
out=gpurun_out/r5b; mkdir -p $out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wide" > $out/pytest_gemm.txt 2>&1; tail -3 $out/pytest_gemm.txt
timeout 900 python tools/bench_deep_gemm.py --json $out/deep_gemm_B.json > $out/deep_gemm_B.txt 2>&1; cat $out/deep_gemm_B.txt
