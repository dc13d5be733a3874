mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
for v in b64_w1 b64_w2 b64_w3 b64_w4 b256_w1 b256_w2 b256_w3 b256_w4; do
  SAMSIM_HIP_LIB=samsim_amd/csrc/variants/libsamsim_hip_$v.so timeout -k 10 300 python bench.py --steps 4 --warmup 1 --substeps 10 --no-cpu-baseline > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$v.json"))
print("$v", "%.3e col-steps/s"%d["value"], "%.1f ms/launch"%d["roofline"]["mean_launch_ms"], "frac %.4f"%d["roofline"]["frac"])
PY
done
SAMSIM_HIP_LIB=samsim_amd/csrc/variants/libsamsim_hip_b64_w2.so timeout -k 10 300 python bench.py --workload tc1 --steps 4 --warmup 1 --substeps 20 --no-cpu-baseline > gpurun_out/bench_tc1.json 2>gpurun_out/bench_tc1.err; cat gpurun_out/bench_tc1.json | cut -c1-400
