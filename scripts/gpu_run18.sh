cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out && rm -rf gpurun_out/diag_*.jsonl gpurun_out/summary.log gpurun_out/bench_*
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py -m gpu -q --timeout 600 -x -k "alignment or word_timestamps" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/summary.log
tail -30 gpurun_out/pytest_gpu.log; cat gpurun_out/summary.log; grep alignment gpurun_out/diag_pipeline.jsonl
timeout 600 python bench.py --workload cfg3 --batch 128 --steps 2 --warmup 1 --word-timestamps 1 > gpurun_out/bench_cfg3_words.json 2> gpurun_out/bench_cfg3_words.err; echo "cfg3w rc=$?" >> gpurun_out/summary.log
cut -c1-400 gpurun_out/bench_cfg3_words.json; python -c "
import json; d=json.load(open('gpurun_out/bench_cfg3_words.json')); c=d['config']; print(d['value'], c['t_asr'], sum(x['step_s']+x['score_s']+x['host_s'] for x in c['beam_timing']), len(c['beam_timing']))"
tail -2 gpurun_out/bench_cfg3_words.err
