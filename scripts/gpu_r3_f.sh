#!/bin/bash
# GPU-box helper (round 3): compaction + the host-side suites, config-3 timing with a kernel timeline, packed-K1 ablation
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_io_cli.py tests/test_cpp_host.py tests/test_gpu_eigensnp.py tests/test_abi.py "tests/test_gpu_stream.py::test_config5_per_gpu_shard_streamed" -m gpu -x -q --deselect tests/test_gpu_parity.py::test_c4_per_gpu_shard_i8_and_2bit > gpurun_out/pytest_$tag.log 2>&1; rc=$?
tail -12 gpurun_out/pytest_$tag.log
if grep -q "Memory access fault" gpurun_out/pytest_$tag.log; then echo "GPU FAULT"; exit 1; fi
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python scripts/bench_config3.py > gpurun_out/config3_$tag.json 2> gpurun_out/config3_$tag.err; tail -3 gpurun_out/config3_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/config3_$tag.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, 'rsvd_ms', round(v['rsvd_ms'],3), v['gemm_launch_us'], v['stages_us_per_call'])"
CFG3_ONLY=i8/int8 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_c3_$tag -- python3 scripts/bench_config3.py > gpurun_out/trace_c3_$tag.log 2>&1
f=$(find gpurun_out/trace_c3_$tag -name "*kernel_trace.csv" | head -1)
python scripts/call_timeline.py $f 10 > gpurun_out/timeline_c3_$tag.md 2>&1; tail -60 gpurun_out/timeline_c3_$tag.md
ABL="16 17 18 20 24" MODE=real bash scripts/gpu_gq2_ablate.sh $tag || exit 1
ABL="16 19" MODE=random bash scripts/gpu_gq2_ablate.sh $tag || exit 1
