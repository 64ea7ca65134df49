mkdir -p gpurun_out/r3d
(timeout 1500 python -m pytest tests/test_gpu_parallel.py tests/test_cpp_api.py tests/test_bench_spawn.py -m gpu -x -q 2>&1 | tail -30) > gpurun_out/r3d/pytest_par.log
(timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parallel.py --deselect tests/test_bench_spawn.py --deselect tests/test_cpp_api.py 2>&1 | tail -8) > gpurun_out/r3d/pytest_rest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3d/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --sequential-particles --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3d/trace_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r3d/trace.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/r3d/trace/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: find the last k_rng_seed... take the last ~120 kernels
out = []
t0 = None
for r in rows[-140:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if t0 is None: t0 = s
    name = r['Kernel_Name'].split('(')[0].replace('void soil::', '').replace('soil::', '')[:60]
    out.append((round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1), name))
json.dump(out, open('gpurun_out/r3d/last_step_kernels.json', 'w'))
PY
rm -rf gpurun_out/r3d/trace
cat gpurun_out/r3d/pytest_par.log; tail -3 gpurun_out/r3d/pytest_rest.log
