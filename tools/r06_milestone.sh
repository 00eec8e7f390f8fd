#!/bin/bash
# round 6 milestone run on the GPU box: the whole GPU suite, then the counter profiles of the three big workloads (tools/profile_round.sh); tools/summarize_profile.py follows here
TAG=${1:-r06a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2; fi
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1
bash tools/profile_round.sh ${TAG}bath --steps 20 --warmup 5 --workload synthetic-bathroom > $OUT/profile_bath.log 2>&1
bash tools/profile_round.sh ${TAG}hard --steps 20 --warmup 5 --workload synthetic-sm-hard > $OUT/profile_hard.log 2>&1
for t in $TAG ${TAG}bath ${TAG}hard; do python tools/bench_brief.py < gpurun_out/$t/bench.json; done
