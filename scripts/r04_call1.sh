mkdir -p gpurun_out/r04; python bench.py > gpurun_out/r04/bench_side.json 2> gpurun_out/r04/bench_side.err; tail -c 600 gpurun_out/r04/bench_side.err; python -c "
import json
d=json.load(open('gpurun_out/r04/bench_side.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['median_samples_per_s'])
for k,v in d.get('side_configs',{}).items(): print(k, json.dumps(v)[:1500])
"
