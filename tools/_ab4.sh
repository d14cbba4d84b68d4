set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -k "intra" > gpurun_out/ab/p.log 2>&1 || { tail -40 gpurun_out/ab/p.log; exit 1; }
tail -1 gpurun_out/ab/p.log
for wl in 1080p_intra; do
timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-gst-latency > gpurun_out/ab/bench_$wl.log 2>&1
grep '^{' gpurun_out/ab/bench_$wl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_picture']; print('$wl', d['value'], 'dbI',s['deblock_wavefront_idr'],'intra',s['intra_wavefront'])"
done
