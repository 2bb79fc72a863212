#!/bin/bash
# SQ counters of conv_mfma_kernel, production (ab0) vs no-global-loads (ab1) vs pure MFMA loop (ab11).
OUT=gpurun_out/convsq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32"
        "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum")
for v in 0 1 11; do
  export SSD_HIP_LIBRARY=$GRAFT_REPO_ROOT/tf-ssd_amd/csrc/build/ablate/libssd_hip_ab$v.so
  i=0
  for P in "${PASSES[@]}"; do
    rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/v${v}_p$i -o c -- python tests/micro/conv_ablate.py --child > $OUT/v${v}_p$i.log 2>&1 || echo "pass $i failed for v$v: $(tail -2 $OUT/v${v}_p$i.log | cut -c1-200)"
    i=$((i+1))
  done
done
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for v in (0, 1, 11):
    for f in glob.glob("gpurun_out/convsq/v%d_p*/**/*counter_collection.csv" % v, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_mfma" not in k: continue
            k = k.split("<")[1].split(">")[0]
            a = agg[(k, r["Counter_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
        for (k, c), a in agg.items():
            res[(k, c)][v] = a[1] / a[0]
kernels = sorted({k for k, c in res})
for k in kernels:
    print("kernel <%s>" % k)
    for (kk, c), d in sorted(res.items()):
        if kk != k: continue
        print("   %-36s full %14.0f   noloads %14.0f   pure %14.0f" % (c, d.get(0, -1), d.get(1, -1), d.get(11, -1)))
PY
