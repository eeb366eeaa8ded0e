set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(time python bench.py --steps 20 --warmup 5) > "$O/r06_bench_default.log" 2>&1
cp "$O/bench_detail.json" "$O/r06_bench_detail_default.json" 2>/dev/null
python bench.py --workload mixed_fleet --steps 20 --warmup 5 > "$O/r06_bench_mixed_fleet.log" 2>&1
for w in allegro_vector leap_position mixed_fleet; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 1 --workload $w --steps 20 --warmup 5 --headline-only --no-cpu-baseline > "$O/r06_bench_1rank_native_rccl_$w.log" 2>&1
done
for f in r06_bench_default r06_bench_mixed_fleet r06_bench_1rank_native_rccl_allegro_vector r06_bench_1rank_native_rccl_leap_position r06_bench_1rank_native_rccl_mixed_fleet; do
  grep -v "^DETAIL\|^real\|^user\|^sys\|^$" "$O/$f.log" | tail -1 > "$O/$f.json"
  echo "$f: $(head -c 200 "$O/$f.json")"
done
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed"
