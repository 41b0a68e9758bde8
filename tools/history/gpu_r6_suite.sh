#!/bin/bash
# round 6: the whole GPU suite as the driver runs it (wall clock, slowest tests, binaries' sha256), then the default bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6_suite
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6_suite/suite.log 2>&1
tail -n 30 gpurun_out/r6_suite/suite.log | cut -c1-200
( time python bench.py ) > gpurun_out/r6_suite/bench.json 2> gpurun_out/r6_suite/bench.err
tail -n 4 gpurun_out/r6_suite/bench.err
head -c 3000 gpurun_out/r6_suite/bench.json
