python -m pytest tests/test_conv_gpu.py -q -x -k "mobilenet_v2_ssd_forward_parity or poisoned_arena" 2>&1 | tail -2
python tests/micro/band_ab.py 64 2>&1 | grep -v "dwproj" | tail -9
