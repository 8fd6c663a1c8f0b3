#!/bin/bash
# GPU-box helper (round 3): parity after the orth / narrow-kernel changes, bench + timeline, configs[2] shape, then the 2-bit parity
# tests again with three digit planes as the packed default (GPCA_PACKED_PLANES=3).   usage: gpu_r3_b.sh <tag>
tag=$1
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_eigensnp.py -m gpu -x -q --deselect tests/test_gpu_parity.py::test_c4_per_gpu_shard_i8_and_2bit --deselect tests/test_gpu_stream.py::test_config5_per_gpu_shard_streamed > gpurun_out/pytest_$tag.log 2>&1
tail -15 gpurun_out/pytest_$tag.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err || tail -5 gpurun_out/bench_$tag.err
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$tag.json"))
print("ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["all_kernels_ms_per_step"])
for k in ("f32_mfma_path", "packed_2bit_residency", "packed_2bit_four_planes"):
    if k in d: print(k, d[k]["ms_per_step"])
PY
timeout -k 10 300 python scripts/bench_config3.py > gpurun_out/config3_$tag.json 2> gpurun_out/config3_$tag.err; tail -3 gpurun_out/config3_$tag.err; cat gpurun_out/config3_$tag.json
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$tag -- python bench.py --no-extras --no-cpu-baseline --no-second-path --steps 12 > gpurun_out/trace_$tag.log 2>&1
f=$(find gpurun_out/trace_$tag -name "*kernel_trace.csv" | head -1)
python scripts/call_timeline.py $f 8 > gpurun_out/timeline_$tag.md 2>&1; tail -70 gpurun_out/timeline_$tag.md
GPCA_PACKED_PLANES=3 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py -m gpu -q -k "2bit or packed" --deselect tests/test_gpu_parity.py::test_c4_per_gpu_shard_i8_and_2bit --deselect tests/test_gpu_stream.py::test_config5_per_gpu_shard_streamed > gpurun_out/pytest_3planes_$tag.log 2>&1
tail -30 gpurun_out/pytest_3planes_$tag.log
