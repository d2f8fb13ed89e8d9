#!/bin/bash
# round 4, call 5d: functional.py split into a package: full GPU suite, the two fuzzers, headline bench (loss must be unchanged)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5d
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python tools/fuzz_mixed.py > $O/fuzz_mixed.txt 2>&1 || { tail -20 $O/fuzz_mixed.txt; exit 1; }
tail -2 $O/fuzz_mixed.txt
python tools/fuzz_paths.py > $O/fuzz_paths.txt 2>&1 || { tail -20 $O/fuzz_paths.txt; exit 1; }
tail -2 $O/fuzz_paths.txt
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5d/bench.json"))
print(d["ms_per_step"], d["value"], d["loss"], d["roofline"]["frac"], d["caller_reference"]["ms_per_step"], d["exact_f32"]["ms_per_step"])
PY
