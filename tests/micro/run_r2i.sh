mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "winograd" 2>&1 | tail -2
python bench.py --layers --no-cpu-baseline > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/layers.txt; cut -c1-330 gpurun_out/r2i/bench.json; grep "conv_heads" gpurun_out/r2i/layers.txt
python bench.py --backbone vgg16 --layers --no-cpu-baseline > gpurun_out/r2i/bench_vgg.json 2> gpurun_out/r2i/layers_vgg.txt; cut -c1-330 gpurun_out/r2i/bench_vgg.json; grep "wino" gpurun_out/r2i/layers_vgg.txt
