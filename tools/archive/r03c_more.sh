mkdir -p gpurun_out/r03c
python bench.py > gpurun_out/r03c/bench_64spp.json 2>gpurun_out/r03c/e1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > gpurun_out/r03c/bench_bathroom.json 2>gpurun_out/r03c/e2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload cornell-glass --width 1024 --height 1024 > gpurun_out/r03c/bench_cornell.json 2>gpurun_out/r03c/e3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --via-loader > gpurun_out/r03c/bench_loader.json 2>gpurun_out/r03c/e4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tracer-param PathSemantics=1 > gpurun_out/r03c/bench_wavefront_rules.json 2>gpurun_out/r03c/e5
python tools/plugin_compare.py > gpurun_out/r03c/plugin_compare.txt 2>&1
for f in 64spp bathroom cornell loader wavefront_rules; do echo $f; python tools/bench_brief.py < gpurun_out/r03c/bench_$f.json | cut -c1-150; tail -2 gpurun_out/r03c/e?; done 2>&1 | grep -v "^==>" | grep -v "^$"
tail -5 gpurun_out/r03c/plugin_compare.txt
