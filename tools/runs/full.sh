# full GPU test suite + default bench (what the driver runs at round end); usage: bash tools/runs/full.sh <tag>
TAG=${1:-full}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
python -m pytest tests -m gpu -q -x > gpurun_out/$TAG/tests.log 2>&1
tail -8 gpurun_out/$TAG/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; tail -2 gpurun_out/$TAG/smoke.log
timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cat gpurun_out/$TAG/bench.json; tail -3 gpurun_out/$TAG/bench.err
