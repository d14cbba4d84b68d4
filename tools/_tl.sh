set -e
R=$PWD
mkdir -p gpurun_out/tl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/tl -o tl -- python3 $R/bench.py --steps 200 --warmup 40 --no-cpu-baseline --no-gst-latency --sample 0 > $R/gpurun_out/tl/tl.log 2>&1
cd $R
python tools/rocpd_timeline.py gpurun_out/tl/tl_results.db 100 36 > gpurun_out/tl/timeline.txt || true
find gpurun_out/tl -name "*.db" -size +30M -delete || true
cat gpurun_out/tl/timeline.txt
