cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" > gpurun_out/pytest_r5q.txt
python bench.py > gpurun_out/bench_full_r5.json 2> gpurun_out/bench_full_r5.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_r5.txt 2>&1
