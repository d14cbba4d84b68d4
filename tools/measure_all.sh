# Refreshes everything under profiles/ on a GPU box: bench lines of the four workloads and the multi-stream runs, the rocprofv3
# kernel trace and the two PMC passes; afterwards (here): python tools/pmc_summary.py 1 1080p_ippp gpurun_out/final/ks/ks_results.db gpurun_out/final/pmc_f gpurun_out/final/pmc_w
#   gpurun --timeout 1200 -- bash tools/measure_all.sh
set -e
R=$PWD
mkdir -p gpurun_out/final
for wl in 1080p_ippp 1080p_intra 2160p_ippp 720p_ippp; do
  extra=""; [ $wl != 1080p_ippp ] && extra="--no-gst-latency"
  timeout -k 10 400 python bench.py --workload $wl $extra > gpurun_out/final/bench_$wl.log 2>&1
  echo "bench $wl done"
done
for s in 4 8; do
  timeout -k 10 300 python bench.py --streams-per-gpu $s --no-gst-latency --no-cpu-baseline > gpurun_out/final/bench_1080p_ippp_streams$s.log 2>&1
  echo "streams $s done"
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/ks -o ks -- python3 $R/bench.py --no-cpu-baseline --no-gst-latency > $R/gpurun_out/final/ks.log 2>&1
echo "kernel stats done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_f -o f -- python3 $R/bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-gst-latency > $R/gpurun_out/final/pmc_f.log 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_w -o w -- python3 $R/bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-gst-latency > $R/gpurun_out/final/pmc_w.log 2>&1
echo "pmc write done"
cd $R
find gpurun_out/final -name "*.db" -size +30M -delete || true
ls -la gpurun_out/final gpurun_out/final/ks gpurun_out/final/pmc_f | head -40
