mkdir -p gpurun_out/r2j
timeout 1200 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "winograd or forward_parity or predict_end_to_end or poisoned" 2>&1 | tail -3
python bench.py --layers --no-cpu-baseline > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/layers.txt; cut -c1-330 gpurun_out/r2j/bench.json; grep "conv_heads" gpurun_out/r2j/layers.txt | head -3
python bench.py --backbone vgg16 --layers --no-cpu-baseline > gpurun_out/r2j/bench_vgg.json 2> gpurun_out/r2j/layers_vgg.txt; cut -c1-330 gpurun_out/r2j/bench_vgg.json; grep -c "wino" gpurun_out/r2j/layers_vgg.txt
