# Refreshes everything under profiles/ on a GPU box: the rocprofv3 kernel trace and the two PMC passes first (so that the bench lines can
# attach the HBM traffic of THIS build; MI355ENC_SERIAL=1 there: counter collection serialises kernel dispatches, and the band deblocker
# cannot follow an intra_p_kernel that is not allowed to run beside it), then the bench lines of the four workloads and the multi-stream runs.
#   gpurun --timeout 1200 -- bash tools/measure_all.sh [ROUND]      afterwards: cp gpurun_out/final/profiles/* profiles/
set -e
R=$PWD
RND=${1:-2}
mkdir -p gpurun_out/final/profiles
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/ks -o ks -- python3 $R/bench.py --no-cpu-baseline --no-gst-latency > $R/gpurun_out/final/ks.log 2>&1
echo "kernel stats done"
MI355ENC_SERIAL=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_f -o f -- python3 $R/bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-gst-latency > $R/gpurun_out/final/pmc_f.log 2>&1
echo "pmc fetch done"
MI355ENC_SERIAL=1 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_w -o w -- python3 $R/bench.py --steps 120 --warmup 20 --no-cpu-baseline --no-gst-latency > $R/gpurun_out/final/pmc_w.log 2>&1
echo "pmc write done"
cd $R
python tools/pmc_summary.py $RND 1080p_ippp gpurun_out/final/ks/ks_results.db gpurun_out/final/pmc_f gpurun_out/final/pmc_w > gpurun_out/final/pmc_summary.log 2>&1
for wl in 1080p_ippp 1080p_intra 2160p_ippp 720p_ippp; do
  extra=""; [ $wl != 1080p_ippp ] && extra="--no-gst-latency"
  [ $wl = 2160p_ippp ] && extra="$extra --depth 1" # at 20 Mbit/s the 4K clip sits where P pictures overrun their targets; with three pictures in flight rate control ends one GOP in five with runs of P_Skip pictures, which cost no device time
  timeout -k 10 400 python bench.py --workload $wl $extra > gpurun_out/final/bench_$wl.log 2>&1
  grep '^{' gpurun_out/final/bench_$wl.log | tail -1 > profiles/r0${RND}_bench_$wl.json
  echo "bench $wl done"
done
for s in 2 4; do
  timeout -k 10 300 python bench.py --streams-per-gpu $s --no-gst-latency --no-cpu-baseline > gpurun_out/final/bench_1080p_ippp_streams$s.log 2>&1
  grep '^{' gpurun_out/final/bench_1080p_ippp_streams$s.log | tail -1 > profiles/r0${RND}_bench_1080p_ippp_streams$s.json
  echo "streams $s done"
done
cp profiles/r0${RND}_* gpurun_out/final/profiles/
find gpurun_out/final -name "*.db" -size +30M -delete || true
ls -la gpurun_out/final/profiles
