for w in 1280 512 384 320; do echo "JPK_AD_WARM=$w"; export JPK_AD_WARM=$w; bash tools/enc_stats.sh gpurun_out/r04i_$w 2>&1 | grep -E "k_adapt|k_pairs|^[0-9]"; done
JPK_AD_WARM=320 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_diff.py -x -q 2>&1 | tail -2
