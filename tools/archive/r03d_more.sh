mkdir -p gpurun_out/r03d
python bench.py > gpurun_out/r03d/bench_64spp.json 2>gpurun_out/r03d/e1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload synthetic-bathroom > gpurun_out/r03d/bench_bathroom.json 2>gpurun_out/r03d/e2
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload cornell-glass --width 1024 --height 1024 > gpurun_out/r03d/bench_cornell.json 2>gpurun_out/r03d/e3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --via-loader > gpurun_out/r03d/bench_loader.json 2>gpurun_out/r03d/e4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tracer-param PathSemantics=1 > gpurun_out/r03d/bench_wavefront_rules.json 2>gpurun_out/r03d/e5
python tools/plugin_compare.py > gpurun_out/r03d/plugin_compare.txt 2>&1
for f in 64spp bathroom cornell loader wavefront_rules; do echo $f; python tools/bench_brief.py < gpurun_out/r03d/bench_$f.json | cut -c1-150; tail -2 gpurun_out/r03d/e?; done 2>&1 | grep -v "^==>" | grep -v "^$"
tail -5 gpurun_out/r03d/plugin_compare.txt
