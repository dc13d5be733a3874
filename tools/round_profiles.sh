# round_profiles.sh -- the one gpurun call behind the profiles/rN_* files of a round: default bench, rocprofv3 trace + PMC passes, region clocks
# and counters (profiling builds under samsim_amd/csrc/variants: -DSAMSIM_STAMPS=1 / =2, -DSAMSIM_PATH_MODE=1), secondary workloads, same-bits
# checks, the two probes, the 20-step bench.  Run as: gpurun --timeout 1200 -- "bash tools/round_profiles.sh"; then tools/collect_profiles.py.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x   # (collect_profiles.py takes this directory)
mkdir -p $O
V=samsim_amd/csrc/variants
md5sum samsim_amd/csrc/libsamsim_hip.so $V/*.so | tee $O/md5.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# 1. the default bench line (stage windows, cpu baseline with the reference binary)
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err || (tail -5 $O/bench_default.err; exit 1)
python - <<'PY'
import json; d=json.load(open("gpurun_out/r3x/bench_default.json")); print("default %.4e"%d["value"], d["ms_per_step"], round(d["roofline"]["frac"],4), d["failed_columns"]); print(json.dumps(d["cpu_baseline"])[:1800]); print({k:("%.3e"%v["column_timesteps_per_s"], v["failed_columns"]) for k,v in d["extra"]["stages"].items()})
PY
# 2. rocprofv3: kernel trace + stats, PMC passes, on the headline step (500 time steps per step)
timeout -k 10 700 bash tools/profile.sh $O/prof500 --substeps 500 --steps 3 --warmup 1
python tools/pmc_summary.py --dir $O/prof500 --substeps 500 --lib samsim_amd/csrc/libsamsim_hip.so --parts 2 > $O/pmc_summary_500.json
find $O -name "*.csv" -size +2M -delete
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3x/pmc_summary_500.json")); print({k:v for k,v in d.items() if not isinstance(v,(dict,list))}); print(d.get("instructions_per_layer_cell_of_a_wave")); print(d.get("kernel_trace"), d.get("kernel_stats"))
PY
# 3. regions and counters
SAMSIM_HIP_LIB=$V/libsamsim_hip_st1.so timeout -k 10 120 python tools/stamps.py > $O/stamps_winter.json
SAMSIM_HIP_LIB=$V/libsamsim_hip_st1.so timeout -k 10 200 python tools/stamps.py --fixture sheba_ensemble_80_day360.npz --launches 2 > $O/stamps_melt.json
SAMSIM_HIP_LIB=$V/libsamsim_hip_st2.so timeout -k 10 200 python tools/stamps.py > $O/counters_winter.json
SAMSIM_HIP_LIB=$V/libsamsim_hip_st2.so timeout -k 10 200 python tools/stamps.py --fixture sheba_ensemble_80_day360.npz --launches 2 > $O/counters_melt.json
# 4. secondary workloads
timeout -k 10 200 python bench.py --workload tc1 --no-cpu-baseline --substeps 100 --steps 5 --warmup 1 > $O/bench_tc1.json 2> $O/bench_tc1.err
timeout -k 10 300 python bench.py --workload cfg5 --ncol 262144 --no-cpu-baseline --substeps 20 --steps 5 --warmup 1 > $O/bench_cfg5.json 2> $O/bench_cfg5.err
timeout -k 10 200 python bench.py --nlayer 100 --no-cpu-baseline --no-extra --substeps 100 --steps 5 --warmup 1 > $O/bench_nlayer100.json 2> $O/bench_nlayer100.err
timeout -k 10 300 python bench.py --gpus 2 --device-map 0,0 --ncol 524288 --no-cpu-baseline --no-extra --steps 4 --warmup 1 > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two_ranks_one_gpu.err
python - <<'PY'
import json
for n in ("tc1","cfg5","nlayer100","two_ranks_one_gpu"):
    d=json.load(open(f"gpurun_out/r3x/bench_{n}.json")); print(n, "%.4e"%d["value"], round(d["roofline"]["frac"],4), d["failed_columns"], d["n_gpus"], d["config"].get("shared_devices"))
PY
# 5. same bits as the unfused order
for fx in sheba_ensemble_80.npz sheba_ensemble_80_day345.npz sheba_ensemble_80_day360.npz; do
  SAMSIM_HIP_LIB=$V/libsamsim_hip_pm1.so timeout -k 10 300 python tools/path_equiv.py --steps 2000 --fixture $fx --out /tmp/pm1.npz
  timeout -k 10 300 python tools/path_equiv.py --steps 2000 --fixture $fx --out /tmp/pm2.npz
  python tools/path_equiv.py --compare /tmp/pm1.npz /tmp/pm2.npz > $O/path_equiv_$fx.json || true
done
SAMSIM_HIP_LIB=$V/libsamsim_hip_pm1.so timeout -k 10 300 python tools/path_equiv.py --cfg5 --ncol 1024 --steps 400 --out /tmp/pm1_cfg5.npz
timeout -k 10 300 python tools/path_equiv.py --cfg5 --ncol 1024 --steps 400 --out /tmp/pm2_cfg5.npz
python tools/path_equiv.py --compare /tmp/pm1_cfg5.npz /tmp/pm2_cfg5.npz > $O/path_equiv_cfg5.json || true
grep -h verdict $O/path_equiv_*.json
tools/div_probe | tee $O/div_probe.txt
tools/layout_probe 8 24 | tee $O/layout_probe.txt
SAMSIM_HIP_LIB=$V/libsamsim_hip_st2.so timeout -k 10 200 python tools/stamps.py --fixture sheba_ensemble_80_day345.npz --launches 2 > $O/counters_melt345.json
timeout -k 10 300 python bench.py --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err
python - <<'PY'
import json; d=json.load(open("gpurun_out/r3x/bench_steps20.json")); print("steps20 %.4e"%d["value"], d["ms_per_step"], round(d["roofline"]["frac"],4), d["failed_columns"], d["roofline"].get("traffic"))
PY
