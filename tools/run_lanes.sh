export SSD_HIP_LANE_CALIBRATE=0 GPU_MAX_HW_QUEUES=3 LANES=3
echo base; python tests/micro/lanes_now.py 2>&1 | tail -1
echo "heads split 1"; LANE_TABLE_EDIT="1_conv_heads=1,2_conv_heads=1" python tests/micro/lanes_now.py 2>&1 | tail -1
echo "heads split 2"; LANE_TABLE_EDIT="1_conv_heads=2,2_conv_heads=2" python tests/micro/lanes_now.py 2>&1 | tail -1
echo "image groups cap 1"; SSD_IMAGE_GROUPS_CAP=1 python tests/micro/lanes_now.py 2>&1 | tail -1
echo "image groups cap 2"; SSD_IMAGE_GROUPS_CAP=2 python tests/micro/lanes_now.py 2>&1 | tail -1
echo "cap 2 + heads split 2"; SSD_IMAGE_GROUPS_CAP=2 LANE_TABLE_EDIT="1_conv_heads=2,2_conv_heads=2" python tests/micro/lanes_now.py 2>&1 | tail -1
