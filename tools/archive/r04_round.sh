#!/bin/bash
# round 4: the whole GPU test suite, then the profile of the shipped build (tools/profile_round.sh <tag>)
TAG=${1:-r04h}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log
bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log | cut -c1-600
