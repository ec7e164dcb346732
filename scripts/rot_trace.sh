export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rot -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rule-n > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
python scripts/kstats.py gpurun_out/prof_rot/r_results.db 8
python -c "
import json
d=json.load(open('gpurun_out/bench_prof.json')); print(d['ms_per_step'], d['stages_ms'])"
