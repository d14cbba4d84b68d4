# Refreshes everything under profiles/ on a GPU box: per workload the rocprofv3 kernel trace and the two PMC passes (MI355ENC_SERIAL=1 there:
# counter collection serialises kernel dispatches, and a kernel that follows another kernel's progress cannot wait for one that is not
# allowed to run beside it), then the bench lines of the workloads and the multi-stream runs.
#   gpurun --timeout 1200 -- bash tools/measure_all.sh ROUND part      part: prof1 (1080p_ippp) | prof2 (2160p_ippp, 1080p_intra) | bench (the four workloads, then bench2) | bench2 (preset 2, QP-31 all-intra, several streams, two ranks) | price
#   afterwards: cp gpurun_out/final/profiles/* profiles/
set -e
R=$PWD
RND=${1:-4}
PART=${2:-bench}
mkdir -p gpurun_out/final/profiles
prof() { # workload, extra bench args
  wl=$1; shift
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/ks_$wl -o ks -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-gst-latency --no-extras "$@" > $R/gpurun_out/final/ks_$wl.log 2>&1
  echo "kernel stats $wl done"
  MI355ENC_SERIAL=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_f_$wl -o f -- python3 $R/bench.py --workload $wl --steps 120 --warmup 20 --no-cpu-baseline --no-extras --no-gst-latency "$@" > $R/gpurun_out/final/pmc_f_$wl.log 2>&1
  echo "pmc fetch $wl done"
  MI355ENC_SERIAL=1 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/final/pmc_w_$wl -o w -- python3 $R/bench.py --workload $wl --steps 120 --warmup 20 --no-cpu-baseline --no-extras --no-gst-latency "$@" > $R/gpurun_out/final/pmc_w_$wl.log 2>&1
  echo "pmc write $wl done"
  cd $R
  python tools/pmc_summary.py $RND $wl gpurun_out/final/ks_$wl/ks_results.db gpurun_out/final/pmc_f_$wl gpurun_out/final/pmc_w_$wl > gpurun_out/final/pmc_summary_$wl.log 2>&1
  cp profiles/r0${RND}_kernel_stats_$wl.csv profiles/r0${RND}_pmc_hbm_traffic_$wl.json gpurun_out/final/profiles/
  find gpurun_out/final -name "*.db" -size +30M -delete || true
}
case $PART in
prof1) prof 1080p_ippp ;;
prof2) prof 2160p_ippp; prof 1080p_intra ;;
bench)
  for wl in 1080p_ippp 1080p_intra 2160p_ippp 720p_ippp; do
    extra=""; [ $wl != 1080p_ippp ] && extra="--no-gst-latency"
    timeout -k 10 400 python bench.py --workload $wl $extra > gpurun_out/final/bench_$wl.log 2>&1
    grep '^{' gpurun_out/final/bench_$wl.log | tail -1 > profiles/r0${RND}_bench_$wl.json
    echo "bench $wl done"
  done
  bash tools/measure_all.sh $RND bench2 ;;
bench2)
  # the toolset the reference's own files select (speed-preset=2: dct8x8 + i8x8 + aq-mode 1), and the all-intra workload at the headline's quantiser instead of the ladder's floor
  for wl in 1080p_ippp 2160p_ippp; do
    timeout -k 10 300 python bench.py --workload $wl --dct8x8 1 --i8x8 1 --aq 1 --no-gst-latency --no-cpu-baseline > gpurun_out/final/bench_${wl}_preset2.log 2>&1
    grep '^{' gpurun_out/final/bench_${wl}_preset2.log | tail -1 > profiles/r0${RND}_bench_${wl}_preset2.json
    echo "bench $wl preset2 done"
  done
  timeout -k 10 300 python bench.py --workload 1080p_intra --fixed-qp 31 --no-gst-latency --no-cpu-baseline > gpurun_out/final/bench_1080p_intra_qp31.log 2>&1
  grep '^{' gpurun_out/final/bench_1080p_intra_qp31.log | tail -1 > profiles/r0${RND}_bench_1080p_intra_qp31.json
  echo "bench 1080p_intra qp31 done"
  for s in 2 4 8; do
    timeout -k 10 300 python bench.py --streams-per-gpu $s --no-gst-latency --no-cpu-baseline > gpurun_out/final/bench_1080p_ippp_streams$s.log 2>&1
    grep '^{' gpurun_out/final/bench_1080p_ippp_streams$s.log | tail -1 > profiles/r0${RND}_bench_1080p_ippp_streams$s.json
    echo "streams $s done"
  done
  timeout -k 10 300 python bench.py --gpus 2 --no-gst-latency --no-cpu-baseline > gpurun_out/final/bench_1080p_ippp_ranks2_shared.log 2>&1
  grep '^{' gpurun_out/final/bench_1080p_ippp_ranks2_shared.log | tail -1 > profiles/r0${RND}_bench_1080p_ippp_ranks2_shared_gpu.json
  cp profiles/r0${RND}_bench_* gpurun_out/final/profiles/ ;;
price)  # what the toolsets and the slices cost and buy (tools/toolset_price.py)
  timeout -k 10 900 python tools/toolset_price.py profiles/r0${RND}_toolset_price.md > gpurun_out/final/toolset_price.log 2>&1
  cp profiles/r0${RND}_toolset_price.md gpurun_out/final/profiles/ ;;
esac
ls -la gpurun_out/final/profiles
