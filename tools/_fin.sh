cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py 2>gpurun_out/fin_err.log | tail -1 > gpurun_out/r04_bench_default.json
python bench.py --workload config5 --steps 24 --warmup 8 2>>gpurun_out/fin_err.log | tail -1 > gpurun_out/r04_bench_config5.json
python bench.py --workload config2 --steps 24 --warmup 8 2>>gpurun_out/fin_err.log | tail -1 > gpurun_out/r04_bench_config2.json
python - <<PY
import json
for f in ("default","config5","config2"):
    d=json.load(open("gpurun_out/r04_bench_%s.json"%f))
    print(f, round(d["value"]/1e9,3), round(d["ms_per_step"],4), d.get("cpu_baseline",{}).get("value"), d.get("roofline",{}).get("frac"))
PY
