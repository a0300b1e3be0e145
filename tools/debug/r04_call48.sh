cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --timeout 1200 -p no:cacheprovider > $O/tests_final.log 2>&1; tail -3 $O/tests_final.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > $O/bench_final.json 2> $O/bench_final.err; cut -c1-200 $O/bench_final.json
