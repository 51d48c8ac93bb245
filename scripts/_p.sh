bash scripts/profile.sh r06g > gpurun_out/prof_r06g.log 2>&1
bash scripts/profile_c5.sh r06g_k8 > gpurun_out/prof_r06g_k8.log 2>&1
python bench.py > gpurun_out/r06g_bench.json 2> gpurun_out/r06g_bench.err
python - <<'PY'
import json
d=json.loads([x for x in open('gpurun_out/r06g_bench.json') if x.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['mfma_pipe_busy_frac'])
print(d['float32_route']['self_dist_100k'])
print(d['float32_route']['knn2'], d['float32_route']['xcheck1'])
print(d['self_2nn']['kernel_ms_steady'], d['fresh_pair']['ms_per_image_pair'], d['expand_c3']['wall_s'], d['verified_vs_oracle'])
PY
python -m pytest tests -q -m gpu --durations=15 -x 2>&1 | tail -25 > gpurun_out/r06g_pytest_gpu.log; tail -4 gpurun_out/r06g_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
