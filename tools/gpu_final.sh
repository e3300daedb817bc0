# end-of-round check on the GPU box: the whole GPU suite, smoke(), the 2-rank rehearsal of bench.py, then the evidence scripts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/evidence_r04
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -6 | tee gpurun_out/evidence_r04/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/evidence_r04/smoke.txt
MZ_BENCH_SINGLE_GPU=1 python bench.py --gpus 2 --steps 200 --warmup 20 > gpurun_out/evidence_r04/bench_2rank_rehearsal.json 2> gpurun_out/evidence_r04/bench_2rank_rehearsal.err; tail -c 600 gpurun_out/evidence_r04/bench_2rank_rehearsal.json
bash tools/evidence_r04a.sh r04
bash tools/evidence_r04b.sh r04
