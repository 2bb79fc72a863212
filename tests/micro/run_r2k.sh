timeout 1500 python -m pytest tests/test_fullsize_gpu.py tests/test_conv_gpu.py -m gpu -q -x -k "full_batch or forward_parity or poisoned or predict_end" 2>&1 | tail -3
python bench.py --layers --no-cpu-baseline 2> /tmp/layers.txt | cut -c1-200; grep "fused\b" /tmp/layers.txt | grep -v " 0.00 GFLOP" | head -8
